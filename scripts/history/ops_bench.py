"""Throughput of the HBM-bound generator ops (blur+noise+act, ToRGB, skip upsample) on the
1024^2 generator's layer shapes (GPU only).  Prints algorithmic GB/s per call."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rewriting_amd import hip          # noqa: E402


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main(batch=int(os.environ.get('RW_BATCH', '32'))):
    dev = 'cuda'
    k1 = torch.tensor([1., 3., 3., 1.])
    k4 = (k1[:, None] * k1[None, :])
    k4 = (k4 / k4.sum() * 4).to(dev)
    for ch, res in [(512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)]:
        x = torch.randn(batch, ch, res + 1, res + 1, device=dev)
        noise = torch.randn(batch, res * res, device=dev)
        nw = torch.full((1,), 0.3, device=dev)
        bias = torch.randn(ch, device=dev)
        ms = timed(lambda: hip.blur_noise_act(x, k4, noise, nw, bias))
        gb = 4.0 * batch * ch * ((res + 1) ** 2 + res * res) / 1e9
        print('blur_noise_act ch=%4d res=%4d  %.3f ms  %.0f GB/s' % (ch, res, ms, gb / ms * 1e3))
        del x
        f = torch.randn(batch, ch, res, res, device=dev)
        w = torch.randn(3, ch, device=dev)
        style = torch.randn(batch, ch, device=dev)
        skip = torch.randn(batch, 3, res, res, device=dev)
        b3 = torch.zeros(3, device=dev)
        ms = timed(lambda: hip.to_rgb(f, w, style, b3, skip, 0.1))
        gb = 4.0 * batch * (ch + 6) * res * res / 1e9
        print('to_rgb         ch=%4d res=%4d  %.3f ms  %.0f GB/s' % (ch, res, ms, gb / ms * 1e3))
        del f
        if res < 1024:
            img = torch.randn(batch * 3, res, res, 1, device=dev)
            ku = (k4 * 1.0).contiguous()
            ms = timed(lambda: hip.upfirdn2d_major(img, ku, 2, 2, 1, 1, 2, 1, 2, 1))
            gb = 4.0 * batch * 3 * 5 * res * res / 1e9
            print('skip upsample         res=%4d  %.3f ms  %.0f GB/s' % (res, ms, gb / ms * 1e3))


if __name__ == '__main__':
    main()
