"""Is blur_noise_act limited by the odd row pitch of the (2H+1) x (2W+1) map?  Same kernel on maps whose input rows
are / are not 16-byte aligned."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rewriting_amd import hip
dev = 'cuda'
k4 = torch.tensor([1., 3., 3., 1.]); k4 = (k4[None] * k4[:, None]); k4 = (k4 / k4.sum() * 4).to(dev)
for b, c, oh, ow in [(64, 32, 1024, 1024), (64, 32, 1024, 1027), (64, 32, 1024, 1023), (64, 64, 512, 512), (64, 64, 512, 515)]:
    x = torch.randn(b, c, oh + 1, ow + 1, device=dev)
    noise = torch.randn(b, oh * ow, device=dev); nw = torch.tensor([0.1], device=dev); bias = torch.randn(c, device=dev)
    hip.blur_noise_act(x, k4, noise, nw, bias); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): y = hip.blur_noise_act(x, k4, noise, nw, bias)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    gb = 4.0 * b * c * ((oh + 1) * (ow + 1) + oh * ow) / 1e9
    print('out %dx%d (in pitch %d, %s): %.3f ms  %.2f TB/s' % (oh, ow, ow + 1, 'aligned' if (ow + 1) % 4 == 0 else 'odd', ms, gb / ms))
    del x, y, noise
# plain copy of the same volume for reference
x = torch.randn(64, 32, 1024, 1024, device=dev); y = torch.empty_like(x)
y.copy_(x); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): y.copy_(x)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
print('torch copy 8.6 GB -> 8.6 GB: %.3f ms %.2f TB/s' % (ms, 2 * x.numel() * 4 / 1e9 / ms))
