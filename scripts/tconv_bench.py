"""Times the fused transposed conv + blur kernel (rw_tconv.hip) beside the routes it replaces on the upsampling layers of
the 1024 generator (batch RW_BATCH, default 64), full epilogue.  One JSON line per layer."""
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
    layers = [('layer9', 512, 512, 32), ('layer11', 512, 256, 64), ('layer13', 256, 128, 128), ('layer15', 128, 64, 256),
              ('layer17', 64, 32, 512)]
    only = os.environ.get('RW_LAYERS')
    g = torch.Generator(device='cpu').manual_seed(0)
    for name, cin, cout, res in layers:
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
        noise = torch.randn(batch, 1, 2 * res, 2 * res, device=DEV)
        k1 = torch.tensor([1., 3., 3., 1.])
        k4 = (k1[:, None] * k1[None, :])
        k4 = (k4 / k4.sum() * 4).to(DEV)
        post = (1 + 0.3 * torch.randn(batch, cout, generator=g)).to(DEV)
        args = dict(style=style, demod=dm, noise=noise, noise_w=nw, bias=bias, act=True, post_scale=post, x_amax=amax)
        row = dict(layer=name, cin=cin, cout=cout, res=res, batch=batch)
        pk = hip.pack_conv_weight_direct16(wt)
        ymax = hip.new_bound(batch * cout * 4 * res * res, DEV)
        a = hip.conv_transpose3x3s2_blur_fused(x, pk, k4, cout, s, y_amax=ymax, **args)
        if os.environ.get('RW_TCONV_ONLY'):           # ablation builds: the fused kernel's time alone
            row['fused_ms'] = timed(lambda: hip.conv_transpose3x3s2_blur_fused(x, pk, k4, cout, s, y_amax=ymax, **args))
            print(json.dumps({k: (float('%.4g' % v) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
            del a, x, noise
            torch.cuda.empty_cache()
            continue
        # the two-pass route of the forward: F(2,2) split quads + strips + blur pass
        uf = hip.pack_conv_transpose_weight_wino(wt, split=True)
        wp = hip.pack_conv_weight(wt, 1)

        def two_pass():
            out = torch.empty(batch, cout, 2 * res + 1, 2 * res + 1, device=DEV)
            hip.conv_transpose3x3s2(x, wp, cout, s, style=style, demod=dm, impl=8, out=out)
            hip.conv_transpose3x3s2_wino(x, uf, cout, s, style=style, demod=dm, out=out, x_amax=amax)
            return hip.blur_noise_act(out, k4, noise, nw, bias, post_scale=post, y_amax=ymax)
        b = two_pass()
        row['rel_vs_two_pass'] = ((a - b).norm() / b.norm()).item()
        row['linf_vs_two_pass_over_range'] = ((a - b).abs().max() / b.abs().max()).item()
        del a, b
        row['fused_ms'] = timed(lambda: hip.conv_transpose3x3s2_blur_fused(x, pk, k4, cout, s, y_amax=ymax, **args))
        row['two_pass_ms'] = timed(two_pass)
        if hip.dconv_transpose_blur_supported(cout, cin, res, res) and cin <= 128:
            pk1 = hip.pack_conv_transpose_blur_weight_direct16(wt, k4)
            row['one_pass_phase_kernels_ms'] = timed(
                lambda: hip.conv_transpose3x3s2_blur_direct16(x, pk1, cout, s, y_amax=ymax, **args))
        alg = batch * (cin * res * res + cout * 4 * res * res) * 4
        row['fused_hbm_gbs'] = round(alg / row['fused_ms'] / 1e6, 1)
        flops = 2 * 9 * cin * cout * res * res * batch
        row['fused_issued_pipe_frac'] = round(flops * 4 * (612 / 512) / (row['fused_ms'] * 1e-3) / 2.5e15, 3)
        print(json.dumps({k: (float('%.4g' % v) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
        del x, noise
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
