"""Times the direct sums on the 16-bit matrix pipe (rw_dconv.hip) beside the F(4x4,3x3) split kernels they would replace,
on the layers of the 1024 generator (batch RW_BATCH, default 64), with the full epilogue.  One JSON line per layer."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rewriting_amd import hip  # noqa: E402

DEV = 'cuda:0'


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    batch = int(os.environ.get('RW_BATCH', '64'))
    layers = [('layer12', 256, 256, 128, 'conv'), ('layer14', 128, 128, 256, 'conv'), ('layer16', 64, 64, 512, 'conv'),
              ('layer18', 32, 32, 1024, 'rgb'), ('layer17', 64, 32, 512, 'up'), ('layer15', 128, 64, 256, 'up')]
    only = os.environ.get('RW_LAYERS')
    g = torch.Generator(device='cpu').manual_seed(0)
    for name, cin, cout, res, kind in layers:
        if only and name not in only.split(','):
            continue
        x = torch.randn(batch, cin, res, res, device=DEV)
        wt = torch.randn(1, cout, cin, 3, 3, generator=g).to(DEV)
        style = (1 + 0.3 * torch.randn(batch, cin, generator=g)).to(DEV)
        s = 1 / math.sqrt(cin * 9)
        dm = hip.demod(hip.weight_sqsum(wt, s), style)
        bias = torch.randn(cout, generator=g).to(DEV)
        nw = torch.tensor([0.1], device=DEV)
        amax = hip.absmax(x)
        row = dict(layer=name, cin=cin, cout=cout, res=res, kind=kind, batch=batch)
        if kind == 'up':
            noise = torch.randn(batch, 1, 2 * res, 2 * res, device=DEV)
            k1 = torch.tensor([1., 3., 3., 1.])
            k4 = (k1[:, None] * k1[None, :])
            k4 = (k4 / k4.sum() * 4).to(DEV)
            args = dict(style=style, demod=dm, noise=noise, noise_w=nw, bias=bias, act=True, x_amax=amax)
            pk = hip.pack_conv_transpose_blur_weight_direct16(wt, k4)
            uf = hip.pack_conv_transpose_blur_weight_wino4(wt, k4, split=True)
            a = hip.conv_transpose3x3s2_blur_direct16(x, pk, cout, s, **args)
            b = hip.conv_transpose3x3s2_blur_wino4(x, uf, cout, s, **args)
            row['rel_vs_f4'] = ((a - b).norm() / b.norm()).item()
            del a, b
            row['direct16_ms'] = timed(lambda: hip.conv_transpose3x3s2_blur_direct16(x, pk, cout, s, **args))
            row['f4_split_ms'] = timed(lambda: hip.conv_transpose3x3s2_blur_wino4(x, uf, cout, s, **args))
        else:
            noise = torch.randn(batch, res * res, device=DEV)
            args = dict(style=style, demod=dm, noise=noise, noise_w=nw, bias=bias, act=True, x_amax=amax)
            pk = hip.pack_conv_weight_direct16(wt)
            uf = hip.pack_conv_weight_wino4(wt, split=True)
            if kind == 'rgb':
                wrgb = torch.randn(3, cout, device=DEV)
                srgb = 1 + 0.3 * torch.randn(batch, cout, device=DEV)
                brgb = torch.randn(3, device=DEV)
                skip = torch.randn(batch, 3, res, res, device=DEV)
                ra = (wrgb, srgb, brgb, skip, 1 / math.sqrt(cout))
                a = hip.conv3x3_direct16_to_rgb(x, pk, cout, s, *ra, **args)[1]
                b = hip.conv3x3_wino4_to_rgb(x, uf, cout, s, *ra, **args)[1]
                row['rel_vs_f4'] = ((a - b).norm() / b.norm()).item()
                del a, b
                row['direct16_ms'] = timed(lambda: hip.conv3x3_direct16_to_rgb(x, pk, cout, s, *ra, **args))
                row['f4_split_ms'] = timed(lambda: hip.conv3x3_wino4_to_rgb(x, uf, cout, s, *ra, **args))
            else:
                a = hip.conv3x3_direct16(x, pk, cout, s, **args)
                b = hip.conv3x3_wino4(x, uf, cout, s, **args)
                row['rel_vs_f4'] = ((a - b).norm() / b.norm()).item()
                if os.environ.get('RW_CHECK'):
                    cargs = {k: v for k, v in args.items() if k != 'x_amax'}
                    ref = hip.conv3x3(x, hip.pack_conv_weight(wt, 0), cout, s, impl=0, **cargs)
                    row['rel_direct16_vs_fp32'] = ((a - ref).norm() / ref.norm()).item()
                    row['rel_f4_vs_fp32'] = ((b - ref).norm() / ref.norm()).item()
                    per = ((b - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1))
                    row['rel_f4_vs_fp32_worst_image'] = (int(per.argmax().item()), per.max().item())
                    del ref
                del a, b
                row['direct16_ms'] = timed(lambda: hip.conv3x3_direct16(x, pk, cout, s, **args))
                row['f4_split_ms'] = timed(lambda: hip.conv3x3_wino4(x, uf, cout, s, **args))
                for wm in (1, 2, 4):
                    if cout % (32 * wm) == 0:
                        os.environ['RW_DCONV_WM'] = str(wm)
                        row['direct16_wm%d_ms' % wm] = timed(lambda: hip.conv3x3_direct16(x, pk, cout, s, **args))
                os.environ.pop('RW_DCONV_WM', None)
        flops = 2 * 9 * cin * cout * res * res * batch * (4 if kind == 'up' else 1)
        row['direct16_pipe_frac'] = round(flops * 4 / (row['direct16_ms'] * 1e-3) / 2.5e15, 3)
        print(json.dumps({k: (float('%.4g' % v) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
