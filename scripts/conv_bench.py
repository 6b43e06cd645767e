"""Per-layer throughput of the implicit-GEMM conv kernels on the layer shapes of the 256^2 and
1024^2 generators (GPU only).  Prints a table and writes gpurun_out/conv_bench.json."""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rewriting_amd import hip          # noqa: E402

LAYERS = [  # (name, cin, cout, input res, upsample)
    ('layer2', 512, 512, 4, False), ('layer3', 512, 512, 4, True), ('layer4', 512, 512, 8, False),
    ('layer5', 512, 512, 8, True), ('layer6', 512, 512, 16, False), ('layer7', 512, 512, 16, True),
    ('layer8', 512, 512, 32, False), ('layer9', 512, 512, 32, True), ('layer10', 512, 512, 64, False),
    ('layer11', 512, 256, 64, True), ('layer12', 256, 256, 128, False), ('layer13', 256, 128, 128, True),
    ('layer14', 128, 128, 256, False), ('layer15', 128, 64, 256, True), ('layer16', 64, 64, 512, False),
    ('layer17', 64, 32, 512, True), ('layer18', 32, 32, 1024, False),
]


def main(batch=int(os.environ.get('RW_BATCH', '8')), iters=5, impl=int(os.environ.get('RW_IMPL', '0'))):
    dev = 'cuda'
    rows = []
    only = os.environ.get('RW_LAYERS')
    for name, cin, cout, res, up in LAYERS:
        if only and name not in only.split(','):
            continue
        x = torch.randn(batch, cin, res, res, device=dev)
        w = torch.randn(1, cout, cin, 3, 3, device=dev)
        style = 1 + 0.3 * torch.randn(batch, cin, device=dev)
        wp = hip.pack_conv_weight(w, 1 if up else 0)
        dm = hip.demod(hip.weight_sqsum(w, 1.0), style)
        split = os.environ.get('RW_PRECISION') == 'bf16x6' and not up and hip.bf16x6_supported(cout, cin, res)
        wino = os.environ.get('RW_ALGO') == 'winograd' and not up and hip.wino_supported(cout, cin, res, res)
        wino4 = os.environ.get('RW_ALGO') == 'winograd4' and not up and hip.wino4_supported(cout, cin, res, res)
        if split:
            wb = hip.pack_conv_weight_bf16x3(w)
        ep = {}
        if not up:          # the fused epilogue of a styled conv block, as the generator runs it
            ep = dict(noise=torch.randn(batch, res * res, device=dev), noise_w=torch.tensor([0.1], device=dev),
                      bias=torch.randn(cout, device=dev), act=True)
        if wino:
            uf = hip.pack_conv_weight_wino(w)
        w4split = os.environ.get('RW_W4_MM', 'f32') != 'f32'        # F(4x4,3x3) on the 16-bit pipe (f16 operand pairs)
        if wino4:
            uf = hip.pack_conv_weight_wino4(w, split=w4split)
        amax = hip.absmax(x) if w4split else None                   # as a producer leaves it behind: not timed
        w4kw = dict(x_amax=amax) if w4split else {}
        upw = os.environ.get('RW_UP_ALGO') == 'wino' and up and hip.conv_transpose_wino_supported(cout, cin, res, res)
        upsplit = os.environ.get('RW_UPW_MM', 'f32') != 'f32' and up and hip.conv_transpose_wino_split_supported(cout, cin, res, res)
        amax_up = hip.absmax(x) if upsplit else None
        upkw = dict(x_amax=amax_up) if upsplit else {}
        if upw:
            ufu = hip.pack_conv_transpose_weight_wino(w, split=upsplit)
            yout = torch.empty(batch, cout, 2 * res + 1, 2 * res + 1, device=dev)
        upmode = os.environ.get('RW_UP_ALGO') if up else None
        if upmode in ('fused', 'wino+blur'):
            k1 = torch.tensor([1., 3., 3., 1.], device=dev)
            k4 = k1[:, None] * k1[None, :]
            k4 = k4 / k4.sum() * 4
            epu = dict(noise=None if os.environ.get('RW_NO_NOISE') else torch.randn(batch, 4 * res * res, device=dev),
                       noise_w=torch.tensor([0.1], device=dev), bias=torch.randn(cout, device=dev))
        upf = upmode == 'fused' and hip.conv_transpose_blur_wino4_supported(cout, cin, res, res)
        upb = upmode == 'wino+blur' and hip.conv_transpose_wino_supported(cout, cin, res, res)
        if upf:
            uf4 = hip.pack_conv_transpose_blur_weight_wino4(w, k4, split=w4split)
        if upb:
            ufu = hip.pack_conv_transpose_weight_wino(w, split=upsplit)
            yout = torch.empty(batch, cout, 2 * res + 1, 2 * res + 1, device=dev)

            def two_pass():
                hip.conv_transpose3x3s2(x, wp, cout, 1.0, style=style, demod=dm, impl=8, out=yout)
                hip.conv_transpose3x3s2_wino(x, ufu, cout, 1.0, style=style, demod=dm, out=yout, **upkw)
                return hip.blur_noise_act(yout, k4, epu['noise'], epu['noise_w'], epu['bias'])
        fn = (lambda: hip.conv_transpose3x3s2_blur_wino4(x, uf4, cout, 1.0, style=style, demod=dm, act=True, **epu, **w4kw)) if upf else \
             two_pass if upb else \
             (lambda: hip.conv_transpose3x3s2_wino(x, ufu, cout, 1.0, style=style, demod=dm, out=yout, **upkw)) if upw else \
             (lambda: hip.conv3x3_wino4(x, uf, cout, 1.0, style=style, demod=dm, **ep, **w4kw)) if wino4 else \
             (lambda: hip.conv3x3_wino(x, uf, cout, 1.0, style=style, demod=dm, **ep)) if wino else \
             (lambda: hip.conv3x3_bf16x6(x, wb, cout, 1.0, style=style, demod=dm, **ep)) if split else \
             (lambda: hip.conv_transpose3x3s2(x, wp, cout, 1.0, style=style, demod=dm, impl=impl)) if up else \
             (lambda: hip.conv3x3(x, wp, cout, 1.0, style=style, demod=dm, impl=impl, **ep))
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        flops = 2.0 * 9 * cin * cout * res * res * batch
        out_res = 2 * res + 1 if up else res
        bytes_io = 4.0 * batch * (cin * res * res + cout * out_res * out_res)
        rows.append(dict(layer=name, cin=cin, cout=cout, res=res, up=up, upmode=upmode, wino=bool(wino) or ('f4' if wino4 else False), mm=os.environ.get('RW_W4_MM', 'f32') if (wino4 or upf) else ('split' if up and upsplit else None), ms=round(ms, 4),
                         tflops=round(flops / ms / 1e9, 2), io_gbs=round(bytes_io / ms / 1e6, 1)))
        print(rows[-1])
        del x, w
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', os.environ.get('RW_OUT', 'conv_bench.json')), 'w') as f:
        json.dump(dict(batch=batch, rows=rows), f, indent=1)


if __name__ == '__main__':
    main()
