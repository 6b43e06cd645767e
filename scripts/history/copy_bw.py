"""Achievable read+write HBM rate of a plain device copy (reference point for the streaming kernels)."""
import torch
for n in (1 << 28, 1 << 30):
    x = torch.empty(n, device='cuda', dtype=torch.float32).normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        y.copy_(x)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print('copy %5.1f GB: %.3f ms  %.0f GB/s (read+write)' % (n * 4 / 1e9, ms, 2 * n * 4 / ms / 1e6))
