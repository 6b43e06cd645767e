"""What may run BESIDE the fused upsampling kernel (csrc/rw_tconv.hip)?  Round 5 found one kernel whose result changed
when it overlapped that kernel on another stream: the then build of to_rgb_kernel, whose inner product the compiler had
written with packed fp32 FMAs (profiles/r05i, r05l; the stand-alone reproducer is scripts/probe/interference_probe.hip,
its table profiles/r06b_interference_probe.jsonl).  The streaming kernels are compiled without packed fp32 math since
(tests/test_build_checks.py checks the generated code).  This file is the guard on the GPU: every kernel this library
launches on a stream other than the trunk's -- ToRGB, the RGB image's upsampling, the modulation prefetch -- and the
kernels a caller is most likely to overlap with a forward -- the one-launch solver (packed fp32 FMAs inside), torch's own
element-wise and reduction kernels -- run on a second stream WHILE the fused kernel runs on the first, in both of its
forms, and must return bit for bit what they return alone."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
REPS = 6


def _aggressors(batch):
    from rewriting_amd import hip
    g = torch.Generator().manual_seed(0)
    cin, cout, res = 64, 32, 512                         # layer 17 of the 1024 generator
    x = torch.randn(batch, cin, res, res, device=DEV)
    wt = torch.randn(1, cout, cin, 3, 3, generator=g).to(DEV)
    style = (1 + 0.3 * torch.randn(batch, cin, generator=g)).to(DEV)
    s = 1 / math.sqrt(cin * 9)
    dm = hip.demod(hip.weight_sqsum(wt, s), style)
    bias = torch.randn(cout, generator=g).to(DEV)
    nw = torch.tensor([0.1], device=DEV)
    noise = torch.randn(batch, 1, 2 * res, 2 * res, device=DEV)
    k1 = torch.tensor([1., 3., 3., 1.])
    k4 = k1[:, None] * k1[None, :]
    k4 = (k4 / k4.sum() * 4).to(DEV)
    ep = dict(style=style, demod=dm, noise=noise, noise_w=nw, bias=bias, act=True, x_amax=hip.absmax(x))
    pk = hip.pack_conv_weight_direct16(wt)

    def launch(form):
        os.environ['RW_TCONV_TY'] = form
        try:
            return hip.conv_transpose3x3s2_blur_fused(x, pk, k4, cout, s, **ep)
        finally:
            del os.environ['RW_TCONV_TY']
    return {'persistent form': lambda: launch('0'), 'one workgroup per CU': lambda: launch('16')}


def _victims(batch):
    from rewriting_amd import hip
    from rewriting_amd.rewrite import hipsolve
    from rewriting_amd.utils.stylegan2 import op
    g = torch.Generator().manual_seed(1)
    xr = torch.randn(batch, 64, 512, 512, device=DEV)
    wr = torch.randn(3, 64, device=DEV)
    sr = 1 + 0.3 * torch.randn(batch, 64, device=DEV)
    br = torch.randn(3, device=DEV)
    skip = torch.randn(batch, 3, 512, 512, device=DEV)
    small = torch.randn(batch, 3, 256, 256, device=DEV)
    k1 = torch.tensor([1., 3., 3., 1.])
    ku = (k1[:, None] * k1[None, :] / 16 * 4).to(DEV)
    lat = torch.randn(batch, 512, device=DEV)
    lw, lb = torch.randn(64, 512, device=DEV), torch.ones(64, device=DEV)
    # the solve of the full-size edit's shape (512 -> 512 channels, 5 x 8 key crop): rw_solve_run_f32, one launch
    O = I = 512
    W0 = torch.randn(1, O, I, 3, 3, generator=g).to(DEV)
    key, sty = torch.randn(1, I, 5, 8, generator=g).to(DEV), (1 + 0.3 * torch.randn(1, I, generator=g)).to(DEV)
    val, bias = torch.randn(1, O, 5, 8, generator=g).to(DEV), torch.randn(O, generator=g).to(DEV)
    ctx = torch.nn.functional.normalize(torch.randn(1, I, generator=g), dim=1).to(DEV)
    nw = torch.tensor([0.1], device=DEV)

    def solve():
        W = W0.clone()
        s = hipsolve.run(W, key, sty, val, bias, nw, ctx, niter=60, piter=10, lr=0.05)
        assert s.one_launch
        return torch.cat([W.reshape(-1), s.losses])
    return {
        'to_rgb_kernel (ToRGB on the RGB stream)': lambda: hip.to_rgb(xr, wr, sr, br, skip, 0.125),
        'upfirdn2d up 2 (the RGB image on the RGB stream)': lambda: op.upfirdn2d(small, ku, up=2, down=1, pad=(2, 1)),
        'equal_linear (modulation prefetch)': lambda: hip.equal_linear(lat, lw, lb, 1 / math.sqrt(512), 1.0, False),
        'rw_solve_run_f32 (one-launch solver, packed fp32 FMAs)': solve,
        'torch.addcmul': lambda: torch.addcmul(xr, xr, xr, value=0.5),
        'torch.sum over channels': lambda: xr.sum(1),
    }


def test_kernels_on_a_second_stream_are_bit_identical_beside_the_fused_upsampling_kernel():
    batch = 8
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    aggressors, victims = _aggressors(batch), _victims(batch)
    report = {}
    for vn, vic in victims.items():
        ref = vic()
        torch.cuda.synchronize()
        assert torch.equal(vic(), ref), vn + ': not deterministic on its own'
        for an, agg in aggressors.items():
            agg()
            torch.cuda.synchronize()
            outs = []
            for _ in range(REPS):
                side.wait_stream(main)
                a = agg()
                with torch.cuda.stream(side):
                    outs.append(vic())
                b = agg()
                del a, b
            torch.cuda.synchronize()
            bad = sum(0 if torch.equal(o, ref) else 1 for o in outs)
            report['%s | %s' % (vn, an)] = bad
            del outs
    assert not any(report.values()), report
