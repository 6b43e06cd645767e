"""GPU parity tests proper: every kernel, called through the C ABI, against the oracle
(oracle/restatement.py, oracle/native_ops.c) and the reference-generated golden fixtures.
Tolerances: fp32 throughout; streaming ops are compared at 1e-6 (bit-level where the
arithmetic order is identical), contractions at 1e-5 relative (reduction order differs from
MKL-DNN), as written at each assert."""
import math
import os

import numpy
import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def cuda(a):
    return torch.as_tensor(a).to(DEV)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_library_loaded_is_the_in_tree_so():
    from rewriting_amd import _lib
    assert _lib.load().rw_abi_version() == _lib.ABI_VERSION
    maps = open('/proc/self/maps').read()
    assert 'librewriting_hip.so' in maps


def test_fused_bias_act_and_upfirdn2d_match_golden_and_c_oracle():
    from rewriting_amd import hip
    from rewriting_amd.utils.stylegan2 import op
    from oracle import native
    g = load_golden('ops')
    ncase = len([k for k in g.files if k.startswith('upfirdn/') and k.endswith('/x')])
    for ci in range(ncase):
        x, k, y = (cuda(g['upfirdn/%d/%s' % (ci, n)]) for n in 'xky')
        up, down, p0, p1 = [int(v) for v in g['upfirdn/%d/cfg' % ci]]
        got = op.upfirdn2d(x, k, up=up, down=down, pad=(p0, p1))
        assert got.shape == y.shape, ci
        assert (got - y).abs().max().item() < 1e-5, ci
    x, b, y = (cuda(g['lrelu/' + n]) for n in ('x', 'b', 'y'))
    xx, bb = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out = op.fused_leaky_relu(xx, bb)
    assert torch.equal(out.detach(), y)                      # identical arithmetic order: bit-exact
    out.backward(cuda(g['lrelu/go']))
    assert (xx.grad - cuda(g['lrelu/gx'])).abs().max().item() < 1e-6
    assert (bb.grad - cuda(g['lrelu/gb'])).abs().max().item() < 1e-5
    want = native.fused_bias_act(g['lrelu/go'], None, g['lrelu/y'], 3, 1, 0.2, 2 ** 0.5)
    assert numpy.array_equal(xx.grad.cpu().numpy(), want)     # bit-exact vs the C statement of the .cu
    assert torch.equal(op.fused_leaky_relu(cuda(g['lrelu2/x']), cuda(g['lrelu2/b'])), cuda(g['lrelu2/y']))
    # unaligned / odd sizes take the scalar path
    rs = numpy.random.RandomState(0)
    for shape in [(1, 3, 5, 7), (2, 5, 1, 1), (3, 7), (1, 4, 16, 16)]:
        xs = rs.randn(*shape).astype('float32')
        bs = rs.randn(shape[1]).astype('float32')
        got = hip.fused_bias_act(cuda(xs), cuda(bs), None, 3, 0, 0.2, 2 ** 0.5).cpu().numpy()
        assert numpy.array_equal(got, native.fused_bias_act(xs, bs, None, 3, 0, 0.2, 2 ** 0.5)), shape
    assert hip.fused_bias_act(torch.empty(0, 3, device=DEV), cuda(bs[:3]), None, 3, 0, 0.2, 1.0).numel() == 0


def test_upfirdn2d_random_configs_against_c_oracle_and_autograd():
    from rewriting_amd import hip
    from rewriting_amd.utils.stylegan2 import op
    from oracle import native, restatement as R
    rs = numpy.random.RandomState(11)
    for trial in range(12):
        up, down = int(rs.randint(1, 3)), int(rs.randint(1, 3))
        kh, kw = int(rs.randint(1, 5)), int(rs.randint(1, 5))
        pads = [int(v) for v in rs.randint(-1, 4, size=4)]
        major, h, w, minor = int(rs.randint(1, 4)), int(rs.randint(3, 12)), int(rs.randint(3, 12)), int(rs.randint(1, 3))
        x = rs.randn(major, h, w, minor).astype('float32')
        k = rs.randn(kh, kw).astype('float32')
        want = native.upfirdn2d(x, k, up, up, down, down, *pads)
        if want.size == 0:
            continue
        got = hip.upfirdn2d_major(cuda(x), cuda(k), up, up, down, down, *pads).cpu().numpy()
        assert got.shape == want.shape
        assert numpy.abs(got - want).max() < 1e-5, (trial, up, down, kh, kw, pads)
    k = R.make_kernel([1, 3, 3, 1]) * 4
    x = torch.randn(2, 3, 8, 8, device=DEV, requires_grad=True)
    y = op.upfirdn2d(x, k.to(DEV), up=2, pad=(2, 1))
    gy = torch.randn_like(y)
    y.backward(gy)
    want = R.upfirdn2d_backward(gy.cpu(), k, 2, 1, (2, 1), x.shape)
    assert (x.grad.cpu() - want).abs().max().item() < 1e-5


def test_upsampling_kernel_of_the_rgb_skip_is_the_generic_walk():
    """up 2 with a 4 x 4 kernel and minor 1 has its own kernel (upfirdn2d_up2k4_kernel): against the C oracle over
    pads, ragged widths (scalar stores) and maps wider than one 256-column tile, and BIT FOR BIT against the generic
    kernel (which the same call takes with minor 2: two interleaved copies of the planes)."""
    from rewriting_amd import hip
    from oracle import native
    rs = numpy.random.RandomState(5)
    for (major, h, w, pads) in [(3, 8, 8, (2, 1, 2, 1)), (2, 5, 7, (2, 1, 2, 1)), (1, 130, 150, (2, 1, 2, 1)),
                                (2, 9, 6, (1, 2, 0, 3)), (1, 4, 4, (3, 3, -1, 2)), (6, 32, 32, (2, 1, 2, 1))]:
        x = rs.randn(major, h, w, 1).astype('float32')
        k = rs.randn(4, 4).astype('float32')
        want = native.upfirdn2d(x, k, 2, 2, 1, 1, *pads)
        got = hip.upfirdn2d_major(cuda(x), cuda(k), 2, 2, 1, 1, *pads).cpu().numpy()
        assert got.shape == want.shape
        assert numpy.abs(got - want).max() < 1e-5, (major, h, w, pads)
        two = numpy.concatenate([x, x], axis=3)
        generic = hip.upfirdn2d_major(cuda(two), cuda(k), 2, 2, 1, 1, *pads).cpu().numpy()
        assert numpy.array_equal(generic[..., :1], got), (major, h, w, pads)


def test_mapping_network_pieces():
    from rewriting_amd import hip
    from oracle import restatement as R
    torch.manual_seed(0)
    z = torch.randn(5, 512)
    assert (hip.pixel_norm(z.to(DEV)).cpu() - z * torch.rsqrt((z ** 2).mean(1, keepdim=True) + 1e-8)).abs().max() < 1e-6
    w, b = torch.randn(512, 512) * 100, torch.randn(512) * 10
    got = hip.equal_linear(z.to(DEV), w.to(DEV), b.to(DEV), 0.01 / math.sqrt(512), 0.01, act=True).cpu()
    want = R.equal_linear(z, w, b, lr_mul=0.01, activation=True)
    assert rel(got, want) < 1e-6
    lat = torch.randn(3, 14, 512)
    got = hip.equal_linear(lat.to(DEV)[:, 5], w.to(DEV), b.to(DEV), 1 / math.sqrt(512), 1.0).cpu()   # strided rows
    assert rel(got, R.equal_linear(lat[:, 5], w, b)) < 1e-6
    # the MFMA kernel (in/out multiples of 16) against the butterfly kernel and float64, ragged batches, 32..512 outputs
    for batch, out_dim in ((1, 512), (64, 512), (250, 512), (37, 32), (64, 64), (10, 256)):
        zz = torch.randn(batch, 512)
        ww, bb = torch.randn(out_dim, 512), torch.randn(out_dim)
        got = hip.equal_linear(zz.to(DEV), ww.to(DEV), bb.to(DEV), 1 / math.sqrt(512), 1.0).cpu()
        os.environ['RW_LINEAR_IMPL'] = '1'
        try:
            other = hip.equal_linear(zz.to(DEV), ww.to(DEV), bb.to(DEV), 1 / math.sqrt(512), 1.0).cpu()
        finally:
            del os.environ['RW_LINEAR_IMPL']
        exact = (zz.double() @ (ww.double() / math.sqrt(512)).t() + bb.double())
        assert got.shape == (batch, out_dim) and rel(got, other) < 1e-6
        assert rel(got.double(), exact) < 5e-7 and rel(other.double(), exact) < 5e-7
    # demodulation factors: the MFMA kernel against the butterfly kernel and float64
    for batch, out_dim, in_dim in ((1, 512, 512), (64, 32, 64), (250, 512, 512), (7, 256, 128)):
        st = (1 + 0.5 * torch.randn(batch, in_dim)).to(DEV)
        wsq = torch.rand(out_dim, in_dim).to(DEV)
        got = hip.demod(wsq, st).cpu()
        os.environ['RW_LINEAR_IMPL'] = '1'
        try:
            other = hip.demod(wsq, st).cpu()
        finally:
            del os.environ['RW_LINEAR_IMPL']
        exact = torch.rsqrt((st.cpu().double() ** 2) @ wsq.cpu().double().t() + 1e-8)
        assert got.shape == (batch, out_dim) and rel(got, other) < 1e-6 and rel(got.double(), exact) < 5e-7
    avg = torch.randn(512)
    got = hip.adjust_latent(z.to(DEV), avg.to(DEV), 6, 0.5).cpu()
    assert torch.allclose(got, (avg + 0.5 * (z - avg)).unsqueeze(1).repeat(1, 6, 1), atol=1e-6)
    assert torch.equal(hip.adjust_latent(z.to(DEV), None, 2, 1.0).cpu(), z.unsqueeze(1).repeat(1, 2, 1))


CONV_CASES = [  # (batch, cin, cout, h, w)
    (2, 512, 512, 4, 4), (3, 512, 512, 8, 8), (1, 512, 512, 32, 32), (2, 512, 256, 16, 16),
    (1, 256, 128, 8, 12), (2, 128, 64, 16, 16), (1, 64, 32, 32, 32), (1, 32, 32, 20, 36),
    (2, 128, 64, 40, 64), (5, 16, 32, 3, 5), (1, 48, 96, 7, 9), (2, 64, 128, 13, 29), (1, 32, 64, 70, 33),
    (1, 64, 64, 8, 8), (3, 128, 32, 4, 6), (16, 512, 512, 4, 4),
]


def _conv_inputs(b, i, o, h, w, seed=0):
    rs = numpy.random.RandomState(seed)
    x = torch.from_numpy(rs.randn(b, i, h, w).astype('float32'))
    wt = torch.from_numpy(rs.randn(1, o, i, 3, 3).astype('float32'))
    style = torch.from_numpy((1 + 0.5 * rs.randn(b, i)).astype('float32'))
    return x, wt, style


def _halo_ok(case):
    """halo_applicable() of rw_conv.hip: 32-pixel column tiles for maps >= 24 wide, 2 x 16 for 9..16,
    4 x 8 for 5..8 (128-channel tiles only)."""
    b, i, o, h, w = case
    return i % 16 == 0 and o % 32 == 0 and (w >= 24 or 9 <= w <= 16 or (5 <= w <= 8 and o % 128 == 0))


@pytest.mark.parametrize('impl', [0, 1, 2, 3, 5])
@pytest.mark.parametrize('case', CONV_CASES)
def test_demodulated_conv_matches_oracle(case, impl):
    """conv2d(x, s*W, pad 1) * demod vs the oracle (DemodulatedConv2dF, models.py:313-329);
    asymmetric random weights, so transposed fragments or swapped taps cannot pass."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    if impl == 3 and not _halo_ok(case):
        pytest.skip('no halo-tile variant for this shape')
    x, wt, style = _conv_inputs(*case)
    s = 1 / math.sqrt(i * 9)
    key = style[:, :, None, None] * x
    want = R.demod_conv(key, style, wt, upsample=False)
    wp = hip.pack_conv_weight(wt.to(DEV), 0)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    want_dm = torch.rsqrt(((s * wt * style.view(b, 1, i, 1, 1)) ** 2).sum([2, 3, 4]) + 1e-8)
    assert rel(dm, want_dm) < 1e-6
    got = hip.conv3x3(hip.style_mul(x.to(DEV), style.to(DEV)), wp, o, s, demod=dm, impl=impl)
    assert rel(got, want) < 1e-5, rel(got, want)
    assert (got.cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    fused = hip.conv3x3(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, impl=impl)   # style on load
    assert rel(fused, want) < 1e-5


# The shapes the benchmark actually runs (BASELINE.json configs[1] and the headline 1024^2 forward; the
# kernels that carry the bench line: conv_halo_kernel<1,4,1,4,8> 32->32 @1024^2, <2,2,1,4,16> 64->64 @512^2,
# <2,2,2,2,16> 128->128 @256^2 and conv_up_halo_kernel<1,4,16,32> 64->32 @512->1024, <2,2,16,32> 128->64 @256->512),
# batch 1, against the oracle's conv on the host (utils/stylegan2/models.py:313-329).
BENCH_CONV_CASES = [(1, 32, 32, 1024, 1024), (1, 64, 64, 512, 512), (1, 128, 128, 256, 256), (1, 256, 256, 128, 128)]
BENCH_UP_CASES = [(1, 64, 32, 512, 512), (1, 128, 64, 256, 256), (1, 256, 128, 128, 128)]


@pytest.mark.parametrize('case', BENCH_CONV_CASES)
def test_bench_shape_convs_match_oracle(case):
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    x, wt, style = _conv_inputs(*case, seed=21)
    rs = numpy.random.RandomState(22)
    bias = torch.from_numpy(rs.randn(o).astype('float32'))
    nw = torch.tensor([0.1])
    noise = R.noise_rows(b, h * w)
    s = 1 / math.sqrt(i * 9)
    conv = R.demod_conv(style[:, :, None, None] * x, style, wt, upsample=False)
    want = R.fused_leaky_relu(conv + nw * noise.view(b, 1, h, w), bias)
    wp = hip.pack_conv_weight(wt.to(DEV), 0)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    plain = hip.conv3x3(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm)
    assert rel(plain, conv) < 1e-5
    assert (plain.cpu() - conv).abs().max().item() < 1e-4 * max(1.0, conv.abs().max().item())
    got = hip.conv3x3(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, noise=noise.to(DEV),
                      noise_w=nw.to(DEV), bias=bias.to(DEV), act=True)
    assert rel(got, want) < 1e-5
    assert (got.cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    if hip.to_rgb_fusable(o, i, w):             # the last layer's call in the bench: ToRGB in the epilogue
        wrgb = torch.from_numpy(rs.randn(3, o).astype('float32'))
        srgb = torch.from_numpy((1 + 0.3 * rs.randn(b, o)).astype('float32'))
        brgb = torch.from_numpy(rs.randn(3).astype('float32'))
        skip = torch.from_numpy(rs.randn(b, 3, h, w).astype('float32'))
        wm = (1 / math.sqrt(o)) * wrgb[None] * srgb[:, None, :]
        want_rgb = torch.einsum('bco,bohw->bchw', wm, want) + brgb.view(1, 3, 1, 1) + skip
        _, rgb = hip.conv3x3_to_rgb(x.to(DEV), wp, o, s, wrgb.to(DEV), srgb.to(DEV), brgb.to(DEV), skip.to(DEV),
                                    1 / math.sqrt(o), style=style.to(DEV), demod=dm, noise=noise.to(DEV),
                                    noise_w=nw.to(DEV), bias=bias.to(DEV), act=True)
        assert rel(rgb, want_rgb) < 1e-5


@pytest.mark.parametrize('case', BENCH_UP_CASES)
def test_bench_shape_transposed_convs_and_blur_match_oracle(case):
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    x, wt, style = _conv_inputs(*case, seed=23)
    rs = numpy.random.RandomState(24)
    bias = torch.from_numpy(rs.randn(o).astype('float32'))
    nw = torch.tensor([0.1])
    noise = R.noise_rows(b, 4 * h * w)
    s = 1 / math.sqrt(i * 9)
    wide = R.demod_conv(style[:, :, None, None] * x, style, wt, upsample=True)
    k4 = R.make_kernel([1, 3, 3, 1]) * 4
    want = R.fused_leaky_relu(R.upfirdn2d(wide, k4, pad=(1, 1)) + nw * noise.view(b, 1, 2 * h, 2 * w), bias)
    wp = hip.pack_conv_weight(wt.to(DEV), 1)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    got_wide = hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm)
    assert rel(got_wide, wide) < 1e-5
    assert (got_wide.cpu() - wide).abs().max().item() < 1e-4 * max(1.0, wide.abs().max().item())
    got = hip.blur_noise_act(got_wide, k4.to(DEV), noise.to(DEV), nw.to(DEV), bias.to(DEV))
    assert rel(got, want) < 1e-5
    assert (got.cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


# ---- Winograd F(2x2,3x3) stride-1 convolution (rw_wino.hip)
WINO_CASES = [(1, 16, 32, 8, 32), (2, 64, 64, 16, 32), (1, 128, 128, 32, 64), (2, 32, 32, 16, 64), (1, 512, 512, 32, 32),
              (3, 48, 96, 24, 96), (2, 24, 160, 8, 64), (1, 32, 32, 1024, 1024), (1, 64, 64, 512, 512),
              (1, 128, 128, 256, 256), (1, 256, 256, 128, 128), (1, 512, 512, 64, 64),
              # maps 16 wide (the NRW shapes: a wave's 16 tiles = two map tile rows of 8): the 16^2 layers, both
              # workgroup shapes, several workgroups per image
              (3, 512, 512, 16, 16), (2, 64, 96, 16, 16), (2, 32, 64, 32, 16), (1, 16, 32, 48, 16),
              # whole 8 x 8 / 4 x 4 images per wave (layers 4 and 2), both workgroup shapes, batches that fill the last
              # workgroup and batches that do not
              (4, 512, 512, 8, 8), (3, 64, 64, 8, 8), (5, 32, 96, 8, 8), (1, 16, 32, 8, 8), (16, 512, 512, 4, 4),
              (11, 64, 64, 4, 4), (37, 32, 32, 4, 4), (1, 24, 96, 4, 4), (250, 64, 128, 4, 4)]


@pytest.mark.parametrize('case', WINO_CASES)
def test_winograd_conv_matches_oracle_and_direct_kernel(case):
    """hip.conv3x3_wino against the oracle's convolution (DemodulatedConv2dF, models.py:313-329) with the full
    epilogue, at the direct kernel's bars, and against the direct MFMA kernel itself; asymmetric random weights
    and inputs (a transposed transform or swapped tile rows / columns cannot pass), input channels spread over
    two decades so that cancellation in B^T d B is exercised."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    assert hip.wino_supported(o, i, h, w)
    x, wt, style = _conv_inputs(*case, seed=31)
    rs = numpy.random.RandomState(32)
    x = x * torch.from_numpy(numpy.exp(1.5 * rs.randn(1, i, 1, 1)).astype('float32'))
    bias = torch.from_numpy(rs.randn(o).astype('float32'))
    nw = torch.tensor([0.2])
    noise = torch.from_numpy(rs.randn(b, h * w).astype('float32'))
    s = 1 / math.sqrt(i * 9)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    uf = hip.pack_conv_weight_wino(wt.to(DEV))
    wp = hip.pack_conv_weight(wt.to(DEV), 0)
    plain = hip.conv3x3_wino(x.to(DEV), uf, o, s, style=style.to(DEV), demod=dm)
    # (the halo / im2col kernels want in_ch % 16; the VALU cross-check kernel takes anything)
    direct = hip.conv3x3(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, impl=0 if i % 16 == 0 else 1)
    scale = direct.abs().max().item()
    assert (plain - direct).abs().max().item() < 2e-5 * scale, (plain - direct).abs().max().item() / scale
    assert rel(plain, direct) < 3e-6
    args = dict(style=style.to(DEV), demod=dm, noise=noise.to(DEV), noise_w=nw.to(DEV), bias=bias.to(DEV), act=True)
    got = hip.conv3x3_wino(x.to(DEV), uf, o, s, **args)
    if b * i * o * h * w <= 2 ** 32:                       # the oracle on the host cores
        key = style[:, :, None, None] * x
        conv = R.demod_conv(key, style, wt, upsample=False)
        want = R.fused_leaky_relu(conv + nw * noise.view(b, 1, h, w), bias)
        assert rel(plain, conv) < 1e-5
        assert (plain.cpu() - conv).abs().max().item() < 1e-4 * max(1.0, conv.abs().max().item())
        assert rel(got, want) < 1e-5
        # against float64: the Winograd kernel is in the direct kernel's error class
        ref = torch.nn.functional.conv2d(key.double(), wt[0].double(), padding=1) * s * dm.cpu().double()[:, :, None, None]
        e_w = ((plain.cpu().double() - ref).norm() / ref.norm()).item()
        e_d = ((direct.cpu().double() - ref).norm() / ref.norm()).item()
        assert e_w < 4 * e_d + 2e-7 and e_w < 3e-6, (e_w, e_d)
    same = hip.conv3x3(x.to(DEV), wp, o, s, impl=0 if i % 16 == 0 else 1, **args)
    assert (got - same).abs().max().item() < 2e-5 * max(1.0, same.abs().max().item())
    if o == 32 and w >= 32:                                 # ToRGB in the epilogue (the widest layer: not the 16-wide shapes)
        wrgb = torch.from_numpy(rs.randn(3, o).astype('float32')).to(DEV)
        srgb = torch.from_numpy((1 + 0.3 * rs.randn(b, o)).astype('float32')).to(DEV)
        brgb = torch.from_numpy(rs.randn(3).astype('float32')).to(DEV)
        skip = torch.from_numpy(rs.randn(b, 3, h, w).astype('float32')).to(DEV)
        want_rgb = hip.to_rgb(got, wrgb, srgb, brgb, skip, 1 / math.sqrt(o))
        y, rgb = hip.conv3x3_wino_to_rgb(x.to(DEV), uf, o, s, wrgb, srgb, brgb, skip, 1 / math.sqrt(o),
                                         store_fmap=True, **args)
        assert torch.equal(y, got)
        assert rel(rgb, want_rgb) < 2e-6
        y2, rgb2 = hip.conv3x3_wino_to_rgb(x.to(DEV), uf, o, s, wrgb, srgb, None, None, 1 / math.sqrt(o), **args)
        assert y2 is None and rel(rgb2, want_rgb - skip - brgb.view(1, 3, 1, 1)) < 1e-5


WINO4_CASES = [(1, 8, 32, 8, 64), (2, 64, 64, 16, 64), (1, 128, 128, 32, 64), (2, 32, 32, 16, 128), (1, 512, 512, 64, 64),
               (3, 48, 96, 24, 192), (2, 24, 160, 8, 64), (1, 32, 32, 1024, 1024), (1, 64, 64, 512, 512),
               (1, 128, 128, 256, 256)]


# 'f32': the products on fp32 MFMAs; 'split' / 'split-ps': every operand as an exact pair of f16 numbers on the 16-bit
# matrix pipe, without / with the 36 points split between the two out-channel waves -- the same bars for all three
@pytest.mark.parametrize('mm', ['f32', 'split', 'split-ps'])
@pytest.mark.parametrize('case', WINO4_CASES)
def test_winograd_f4_conv_matches_oracle_within_its_error_class(case, mm, monkeypatch):
    """hip.conv3x3_wino4 (F(4x4,3x3), opt-in) against the direct kernel, the oracle's convolution with the full
    epilogue, and float64: the result of the same convolution, at the accuracy the algorithm has in fp32 -- 1e-5
    relative (Frobenius) and 1e-4 of the output range, an order of magnitude looser than the default kernels."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    assert hip.wino4_supported(o, i, h, w)
    split = mm != 'f32'
    if split:
        monkeypatch.setenv('RW_W4H_PS', '1' if mm == 'split-ps' else '0')
    x, wt, style = _conv_inputs(*case, seed=41)
    rs = numpy.random.RandomState(42)
    x = x * torch.from_numpy(numpy.exp(1.5 * rs.randn(1, i, 1, 1)).astype('float32'))
    bias = torch.from_numpy(rs.randn(o).astype('float32'))
    nw = torch.tensor([0.2])
    noise = torch.from_numpy(rs.randn(b, h * w).astype('float32'))
    s = 1 / math.sqrt(i * 9)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    uf = hip.pack_conv_weight_wino4(wt.to(DEV), split=split)
    wp = hip.pack_conv_weight(wt.to(DEV), 0)
    plain = hip.conv3x3_wino4(x.to(DEV), uf, o, s, style=style.to(DEV), demod=dm)
    direct = hip.conv3x3(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, impl=0 if i % 16 == 0 else 1)
    scale = direct.abs().max().item()
    if split:
        # the bound on the input may be loose, and the kernel reports the maximum of what it wrote
        # (the buffer is poisoned first: every slot the reduction reads must have been written by the launch itself)
        amax, ymax = hip.absmax(x.to(DEV)), hip.new_bound(b * o * h * w, DEV).fill_(3e38)
        assert hip.bound_value(amax) == x.abs().max().item()
        loose = hip.conv3x3_wino4(x.to(DEV), uf, o, s, style=style.to(DEV), demod=dm, x_amax=amax * 37.0, y_amax=ymax)
        assert hip.bound_value(ymax) == loose.abs().max().item()
        assert rel(loose, plain) < 2e-6, rel(loose, plain)
    assert (plain - direct).abs().max().item() < 1e-4 * scale, (plain - direct).abs().max().item() / scale
    assert rel(plain, direct) < 2e-5, rel(plain, direct)
    args = dict(style=style.to(DEV), demod=dm, noise=noise.to(DEV), noise_w=nw.to(DEV), bias=bias.to(DEV), act=True)
    got = hip.conv3x3_wino4(x.to(DEV), uf, o, s, **args)
    same = hip.conv3x3(x.to(DEV), wp, o, s, impl=0 if i % 16 == 0 else 1, **args)
    assert (got - same).abs().max().item() < 1e-4 * max(1.0, same.abs().max().item())
    if b * i * o * h * w <= 2 ** 32:
        key = style[:, :, None, None] * x
        conv = R.demod_conv(key, style, wt, upsample=False)
        want = R.fused_leaky_relu(conv + nw * noise.view(b, 1, h, w), bias)
        assert rel(got, want) < 2e-5
        ref = torch.nn.functional.conv2d(key.double(), wt[0].double(), padding=1) * s * dm.cpu().double()[:, :, None, None]
        e_w = ((plain.cpu().double() - ref).norm() / ref.norm()).item()
        assert e_w < 2e-5, e_w
    if o == 32:                                             # ToRGB in the epilogue (the feature map is not written)
        assert hip.wino4_to_rgb_supported(o, i, h, w)
        wrgb = torch.from_numpy(rs.randn(3, o).astype('float32')).to(DEV)
        srgb = torch.from_numpy((1 + 0.3 * rs.randn(b, o)).astype('float32')).to(DEV)
        brgb = torch.from_numpy(rs.randn(3).astype('float32')).to(DEV)
        skip = torch.from_numpy(rs.randn(b, 3, h, w).astype('float32')).to(DEV)
        want_rgb = hip.to_rgb(got, wrgb, srgb, brgb, skip, 1 / math.sqrt(o))
        y, rgb = hip.conv3x3_wino4_to_rgb(x.to(DEV), uf, o, s, wrgb, srgb, brgb, skip, 1 / math.sqrt(o), **args)
        assert y is None and rel(rgb, want_rgb) < 2e-6, rel(rgb, want_rgb)
        _, rgb2 = hip.conv3x3_wino4_to_rgb(x.to(DEV), uf, o, s, wrgb, srgb, None, None, 1 / math.sqrt(o), **args)
        assert rel(rgb2, want_rgb - skip - brgb.view(1, 3, 1, 1)) < 1e-5
        plain_rgb = hip.to_rgb(plain, wrgb, srgb, None, None, 1 / math.sqrt(o))
        _, rgb3 = hip.conv3x3_wino4_to_rgb(x.to(DEV), uf, o, s, wrgb, srgb, None, None, 1 / math.sqrt(o),
                                           style=style.to(DEV), demod=dm)
        assert rel(rgb3, plain_rgb) < 1e-5


UP_WINO_CASES = [(2, 16, 32, 4, 32), (1, 64, 64, 8, 32), (1, 128, 64, 16, 64), (3, 32, 32, 12, 96), (1, 512, 512, 32, 32),
                 (1, 24, 96, 8, 64), (1, 64, 32, 512, 512), (1, 128, 64, 256, 256), (1, 512, 256, 64, 64),
                 # input maps 16 wide (a wave's 16 blocks = two block rows of 8): layer 7 of the generators
                 (2, 512, 512, 16, 16), (1, 64, 32, 8, 16), (3, 32, 64, 24, 16),
                 # ... and launches large enough for RUNS along y (both 8-row groups of an image in one workgroup)
                 (128, 32, 256, 16, 16), (64, 512, 512, 16, 16), (171, 16, 64, 24, 16),
                 # whole 8 x 8 / 4 x 4 images per wave, two / eight images per workgroup (layers 5 and 3): batches that
                 # fill the last workgroup and batches that do not
                 (4, 512, 512, 8, 8), (3, 32, 64, 8, 8), (1, 64, 32, 8, 8), (16, 512, 512, 4, 4), (11, 32, 32, 4, 4),
                 (1, 48, 96, 4, 4), (250, 64, 64, 4, 4)]


@pytest.mark.parametrize('mm', ['f32', 'split'])
@pytest.mark.parametrize('case', UP_WINO_CASES)
def test_transposed_conv_f22_matches_direct_kernel_and_oracle(case, mm):
    """hip.conv_transpose3x3s2_wino (F(2,2): 25 instead of 36 multiplies per 2x2 block of quads) + the border strips
    against the direct transposed-conv kernels and the oracle, at the direct kernels' bars (its transforms have
    coefficients 0, +-1); asymmetric random weights, channel scales spread over two decades."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    assert hip.conv_transpose_wino_supported(o, i, h, w)
    split = mm == 'split'           # the products on the 16-bit matrix pipe (exact f16 operand pairs): the same bars
    if split and not hip.conv_transpose_wino_split_supported(o, i, h, w):
        pytest.skip('the split-operand form takes the wide maps only')
    x, wt, style = _conv_inputs(*case, seed=51)
    rs = numpy.random.RandomState(52)
    x = x * torch.from_numpy(numpy.exp(1.5 * rs.randn(1, i, 1, 1)).astype('float32'))
    s = 1 / math.sqrt(i * 9)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    wp = hip.pack_conv_weight(wt.to(DEV), 1)
    uf = hip.pack_conv_transpose_weight_wino(wt.to(DEV), split=split)
    direct = hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm,
                                     impl=0 if i % 16 == 0 else 1)
    out = torch.full_like(direct, float('nan'))
    hip.conv_transpose3x3s2_wino(x.to(DEV), uf, o, s, style=style.to(DEV), demod=dm, out=out)
    if split:           # a loose bound on the input costs low bits of the smallest values only
        out2 = torch.full_like(direct, float('nan'))
        hip.conv_transpose3x3s2_wino(x.to(DEV), uf, o, s, style=style.to(DEV), demod=dm, out=out2,
                                     x_amax=hip.absmax(x.to(DEV)) * 50.0)
        assert rel(out2[:, :, :-1, :-1], out[:, :, :-1, :-1]) < 1e-6
    assert torch.isnan(out[:, :, -1, :]).all() and torch.isnan(out[:, :, :, -1]).all()      # strips untouched
    assert torch.isfinite(out[:, :, :-1, :-1]).all()
    if i % 16 == 0:
        hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, impl=8, out=out)
    else:
        out[:, :, -1, :] = direct[:, :, -1, :]
        out[:, :, :, -1] = direct[:, :, :, -1]
    scale = direct.abs().max().item()
    assert (out - direct).abs().max().item() < 2e-5 * scale, (out - direct).abs().max().item() / scale
    assert rel(out, direct) < 3e-6
    if b * i * o * h * w <= 2 ** 31:
        key = style[:, :, None, None] * x
        want = R.demod_conv(key, style, wt, upsample=True)
        assert rel(out, want) < 1e-5
        assert (out.cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('case', [(3, 32, 64, 5, 7), (2, 64, 32, 16, 16), (1, 512, 512, 4, 4), (67, 16, 32, 8, 8),
                                  (1, 48, 96, 3, 40), (2, 128, 64, 64, 32), (5, 32, 32, 1, 1), (1, 64, 32, 512, 512)])
def test_border_strips_as_gemms_equal_the_im2col_strips(case, monkeypatch):
    """rw_conv_transpose3x3s2_f32 impl 8 (output row 2H and column 2W alone): the GEMM strip kernel against the batched
    im2col launch it replaced (RW_UP_STRIPS=im2col) and against the oracle's transposed convolution, on maps of any
    shape -- ragged position counts, one-pixel maps, batches that do not fill a tile; nothing else of the map is
    touched."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    x, wt, style = _conv_inputs(*case, seed=77)
    s = 1 / math.sqrt(i * 9)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    wp = hip.pack_conv_weight(wt.to(DEV), 1)
    outs = {}
    for mode in ('gemm', 'im2col'):
        monkeypatch.setenv('RW_UP_STRIPS', mode)
        out = torch.full((b, o, 2 * h + 1, 2 * w + 1), float('nan'), device=DEV)
        hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, impl=8, out=out)
        assert torch.isnan(out[:, :, :-1, :-1]).all()
        assert torch.isfinite(out[:, :, -1, :]).all() and torch.isfinite(out[:, :, :, -1]).all()
        outs[mode] = out
    monkeypatch.delenv('RW_UP_STRIPS')
    row = (outs['gemm'][:, :, -1, :], outs['im2col'][:, :, -1, :])
    colm = (outs['gemm'][:, :, :, -1], outs['im2col'][:, :, :, -1])
    scale = max(row[1].abs().max().item(), colm[1].abs().max().item())
    assert (row[0] - row[1]).abs().max().item() < 2e-6 * scale and (colm[0] - colm[1]).abs().max().item() < 2e-6 * scale
    if b * i * o * h * w <= 2 ** 29:
        want = R.demod_conv(style[:, :, None, None] * x, style, wt, upsample=True)
        got = outs['gemm'].cpu()
        assert (got[:, :, -1, :] - want[:, :, -1, :]).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
        assert (got[:, :, :, -1] - want[:, :, :, -1]).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())


UP_BLUR_CASES = [(2, 16, 8, 8, 64), (1, 64, 32, 16, 64), (1, 128, 64, 8, 128), (3, 32, 16, 24, 64), (1, 512, 64, 8, 64),
                 (1, 24, 40, 8, 64), (1, 64, 32, 512, 512)]


@pytest.mark.parametrize('mm', ['f32', 'split', 'split-ps'])
@pytest.mark.parametrize('case', UP_BLUR_CASES)
def test_one_pass_upsampling_conv_matches_conv_then_blur(case, mm, monkeypatch):
    """hip.conv_transpose3x3s2_blur_wino4 (the four output-parity phases of conv_transpose (*) blur as virtual channels
    of the F(4x4,3x3) kernel, noise + bias + leaky ReLU in its epilogue) against the two-pass route of the same library
    (direct transposed conv -> blur_noise_act) and the oracle, at the F(4x4,3x3) bars."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    assert hip.conv_transpose_blur_wino4_supported(o, i, h, w)
    split = mm != 'f32'
    if split:
        monkeypatch.setenv('RW_W4H_PS', '1' if mm == 'split-ps' else '0')
    x, wt, style = _conv_inputs(*case, seed=61)
    rs = numpy.random.RandomState(62)
    x = x * torch.from_numpy(numpy.exp(1.0 * rs.randn(1, i, 1, 1)).astype('float32'))
    s = 1 / math.sqrt(i * 9)
    k1 = torch.tensor([1., 3., 3., 1.])
    k4 = k1[:, None] * k1[None, :]
    k4 = (k4 / k4.sum() * 4).to(DEV)
    noise = torch.from_numpy(rs.randn(b, 1, 2 * h, 2 * w).astype('float32')).to(DEV)
    nw = torch.tensor([0.37], device=DEV)
    bias = torch.from_numpy(rs.randn(o).astype('float32')).to(DEV)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    wp = hip.pack_conv_weight(wt.to(DEV), 1)
    wide = hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm,
                                   impl=0 if i % 16 == 0 and o % 32 == 0 else 1)
    uf = hip.pack_conv_transpose_blur_weight_wino4(wt.to(DEV), k4, split=split)
    results = []
    for kw in (dict(noise=noise, noise_w=nw, bias=bias, act=True), dict()):
        want = hip.blur_noise_act(wide, k4, kw.get('noise'), kw.get('noise_w'), kw.get('bias'))
        ymax = hip.new_bound(b * o * 4 * h * w, DEV).fill_(3e38)
        got = hip.conv_transpose3x3s2_blur_wino4(x.to(DEV), uf, o, s, style=style.to(DEV), demod=dm,
                                                 **(dict(kw, y_amax=ymax) if split else kw))
        assert got.shape == want.shape == (b, o, 2 * h, 2 * w)
        assert not split or hip.bound_value(ymax) == got.abs().max().item()
        scale = want.abs().max().item()
        assert (got - want).abs().max().item() < 1e-4 * scale, (got - want).abs().max().item() / scale
        assert rel(got, want) < 3e-5, rel(got, want)
        results.append((kw, got.cpu()))
    # the oracle leg, INCLUDING the bench's own shape (1, 64, 32, 512, 512) = layer 17 of the 1024 model: the host
    # convolution of one image costs under a second (9.7 GFLOP)
    assert b * i * o * h * w <= 2 ** 29
    key = style[:, :, None, None] * x
    blur = R.upfirdn2d(R.demod_conv(key, style, wt, upsample=True), k4.cpu(), pad=(1, 1))
    for kw, got in results:
        ref = R.fused_leaky_relu(blur + nw.cpu() * noise.cpu(), bias.cpu()) if kw else blur
        assert rel(got, ref) < 3e-5, rel(got, ref)
        assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_winograd_rejects_shapes_it_does_not_take():
    from rewriting_amd import hip
    assert not hip.wino_supported(32, 32, 8, 4) and not hip.wino_supported(48, 32, 32, 32)
    assert not hip.wino_supported(32, 12, 32, 32) and not hip.wino_supported(32, 32, 36, 32)
    assert hip.wino_supported(32, 32, 16, 16) and not hip.wino_supported(32, 32, 24, 16)     # 16 wide: h % 16 == 0
    assert hip.wino_supported(32, 32, 8, 8) and hip.wino_supported(32, 32, 4, 4)             # whole images per wave
    assert not hip.wino_supported(32, 32, 16, 8) and not hip.wino_supported(32, 32, 12, 12)
    x, wt, _ = _conv_inputs(1, 32, 32, 16, 8)
    with pytest.raises(RuntimeError):
        hip.conv3x3_wino(x.to(DEV), hip.pack_conv_weight_wino(wt.to(DEV)), 32, 0.1)


@pytest.mark.parametrize('case', [(2, 32, 32, 40, 64), (1, 64, 64, 32, 32), (2, 16, 32, 24, 70), (1, 128, 64, 24, 33)])
@pytest.mark.parametrize('store', [False, True])
def test_conv_with_fused_to_rgb_matches_separate_kernels(case, store):
    """rw_conv3x3_to_rgb_f32 (ToRGB in the epilogue of the styled conv, feature map optionally not stored)
    against rw_conv3x3_f32 followed by rw_to_rgb_f32."""
    from rewriting_amd import hip
    b, i, o, h, w = case
    x, wt, style = _conv_inputs(*case, seed=7)
    rs = numpy.random.RandomState(11)
    noise = torch.from_numpy(rs.randn(b, h * w).astype('float32')).to(DEV)
    nw = torch.tensor([0.2]).to(DEV)
    bias = torch.from_numpy(rs.randn(o).astype('float32')).to(DEV)
    wrgb = torch.from_numpy(rs.randn(3, o).astype('float32')).to(DEV)
    srgb = torch.from_numpy((1 + 0.3 * rs.randn(b, o)).astype('float32')).to(DEV)
    brgb = torch.from_numpy(rs.randn(3).astype('float32')).to(DEV)
    skip = torch.from_numpy(rs.randn(b, 3, h, w).astype('float32')).to(DEV)
    s = 1 / math.sqrt(i * 9)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    wp = hip.pack_conv_weight(wt.to(DEV), 0)
    args = dict(style=style.to(DEV), demod=dm, noise=noise, noise_w=nw, bias=bias, act=True)
    fmap = hip.conv3x3(x.to(DEV), wp, o, s, **args)
    want = hip.to_rgb(fmap, wrgb, srgb, brgb, skip, 1 / math.sqrt(o))
    y, rgb = hip.conv3x3_to_rgb(x.to(DEV), wp, o, s, wrgb, srgb, brgb, skip, 1 / math.sqrt(o), store_fmap=store, **args)
    assert (y is None) == (not store)
    if store:
        assert torch.equal(y, fmap)
    assert rel(rgb, want) < 2e-6
    _, rgb2 = hip.conv3x3_to_rgb(x.to(DEV), wp, o, s, wrgb, srgb, None, None, 1 / math.sqrt(o), **args)
    assert rel(rgb2, want - skip - brgb.view(1, 3, 1, 1)) < 1e-5


@pytest.mark.parametrize('case', [(2, 128, 64, 40, 64), (1, 64, 32, 32, 32), (2, 512, 256, 16, 16), (3, 512, 512, 8, 8)])
def test_transposed_conv_tiles_and_border_parts_compose(case):
    """impl 7 (quad tiles) and impl 8 (output row 2H / column 2W) write disjoint elements and together
    give exactly what one call gives (they are issued on two streams in the un-hooked forward)."""
    from rewriting_amd import hip
    b, i, o, h, w = case
    x, wt, style = _conv_inputs(*case, seed=2)
    s = 1 / math.sqrt(i * 9)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    wp = hip.pack_conv_weight(wt.to(DEV), 1)
    whole = hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm)
    out = torch.full_like(whole, float('nan'))
    hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, impl=7, out=out)
    assert torch.isnan(out[:, :, -1, :]).all() and torch.isnan(out[:, :, :, -1]).all()      # untouched border
    assert torch.equal(out[:, :, :-1, :-1], whole[:, :, :-1, :-1])
    hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, impl=8, out=out)
    assert torch.equal(out, whole)


@pytest.mark.parametrize('case', [(1, 512, 512, 32, 32), (2, 128, 64, 40, 64), (1, 64, 128, 29, 70),
                                  (2, 32, 64, 24, 33)])
def test_split_bf16x6_conv_matches_fp32(case):
    """The opt-in bf16x6 path against the float64 convolution and against the exact-fp32 MFMA kernel:
    it must be as close to the exact result as the fp32 kernel is (operands split exactly, six piece
    products, fp32 accumulation), with the same fused epilogue."""
    from rewriting_amd import hip
    b, i, o, h, w = case
    x, wt, style = _conv_inputs(*case, seed=3)
    rs = numpy.random.RandomState(5)
    x = x * torch.from_numpy(numpy.exp(3 * rs.randn(1, i, 1, 1)).astype('float32'))     # channels over 4 decades
    noise = torch.from_numpy(rs.randn(b, h * w).astype('float32'))
    nw = torch.tensor([0.3])
    bias = torch.from_numpy(rs.randn(o).astype('float32'))
    s = 1 / math.sqrt(i * 9)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    key = (style[:, :, None, None] * x).double()
    ref = torch.nn.functional.conv2d(key, wt[0].double(), padding=1) * s * dm.cpu().double()[:, :, None, None]
    ref = ref + 0.3 * noise.view(b, 1, h, w).double() + bias.double().view(1, o, 1, 1)
    ref = torch.where(ref > 0, ref, 0.2 * ref) * math.sqrt(2)
    args = dict(style=style.to(DEV), demod=dm, noise=noise.to(DEV), noise_w=nw.to(DEV), bias=bias.to(DEV), act=True)
    f32 = hip.conv3x3(x.to(DEV), hip.pack_conv_weight(wt.to(DEV), 0), o, s, **args)
    got = hip.conv3x3_bf16x6(x.to(DEV), hip.pack_conv_weight_bf16x3(wt.to(DEV)), o, s, **args)
    e32 = (f32.cpu().double() - ref).norm() / ref.norm()
    e16 = (got.cpu().double() - ref).norm() / ref.norm()
    assert e16 < 2e-6, (e16.item(), e32.item())
    assert e16 < 4 * e32 + 2e-7, (e16.item(), e32.item())
    assert (got - f32).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize('impl', [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize('case', CONV_CASES[:9] + CONV_CASES[11:])
def test_transposed_conv_matches_oracle(case, impl):
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    if impl in (3, 4) and not _halo_ok(case):
        pytest.skip('no halo-tile variant for this shape')
    x, wt, style = _conv_inputs(*case, seed=1)
    s = 1 / math.sqrt(i * 9)
    want = R.demod_conv(style[:, :, None, None] * x, style, wt, upsample=True)
    wp = hip.pack_conv_weight(wt.to(DEV), 1)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    got = hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, impl=impl)
    assert got.shape == want.shape
    assert rel(got, want) < 1e-5, rel(got, want)


def test_fused_epilogues_and_streaming_blocks():
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = 2, 128, 64, 16, 16
    x, wt, style = _conv_inputs(b, i, o, h, w, seed=2)
    s = 1 / math.sqrt(i * 9)
    rs = numpy.random.RandomState(5)
    bias = torch.from_numpy(rs.randn(o).astype('float32'))
    nw = torch.tensor([0.37])
    noise = R.noise_rows(b, h * w)
    conv = R.demod_conv(style[:, :, None, None] * x, style, wt, upsample=False)
    want = R.fused_leaky_relu(conv + nw * noise.view(b, 1, h, w), bias)
    wp = hip.pack_conv_weight(wt.to(DEV), 0)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    got = hip.conv3x3(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, noise=noise.to(DEV),
                      noise_w=nw.to(DEV), bias=bias.to(DEV), act=True)
    assert rel(got, want) < 1e-5
    # noise add alone, blur+noise+act, ToRGB
    assert rel(hip.noise_add(conv.to(DEV), noise.to(DEV), nw.to(DEV)), conv + nw * noise.view(b, 1, h, w)) < 1e-7
    k4 = R.make_kernel([1, 3, 3, 1]) * 4
    wide = torch.from_numpy(rs.randn(b, o, 2 * h + 1, 2 * w + 1).astype('float32'))
    n2 = R.noise_rows(b, 4 * h * w)
    want = R.fused_leaky_relu(R.upfirdn2d(wide, k4, pad=(1, 1)) + nw * n2.view(b, 1, 2 * h, 2 * w), bias)
    got = hip.blur_noise_act(wide.to(DEV), k4.to(DEV), n2.to(DEV), nw.to(DEV), bias.to(DEV))
    assert rel(got, want) < 1e-6
    # partial tiles, asymmetric taps; the last two take the register-window kernel (maps >= 256 rows)
    for bb, cc, hh, ww in [(1, 3, 4, 4), (2, 5, 20, 72), (1, 2, 130, 136), (1, 2, 256, 128), (2, 1, 300, 70)]:
        kk = k4.clone()
        kk[0, 1] += 0.03
        wd = torch.from_numpy(rs.randn(bb, cc, hh + 1, ww + 1).astype('float32'))
        bs = torch.from_numpy(rs.randn(cc).astype('float32'))
        nn = R.noise_rows(bb, hh * ww)
        want2 = R.fused_leaky_relu(R.upfirdn2d(wd, kk, pad=(1, 1)) + nw * nn.view(bb, 1, hh, ww), bs)
        got2 = hip.blur_noise_act(wd.to(DEV), kk.to(DEV), nn.to(DEV), nw.to(DEV), bs.to(DEV))
        assert rel(got2, want2) < 1e-6, (bb, cc, hh, ww)
    wrgb = torch.from_numpy(rs.randn(3, i).astype('float32'))
    brgb = torch.from_numpy(rs.randn(3).astype('float32'))
    skip = torch.from_numpy(rs.randn(b, 3, h, w).astype('float32'))
    wm = (1 / math.sqrt(i)) * wrgb[None] * style[:, None, :]
    want = torch.einsum('bci,bihw->bchw', wm, x) + brgb.view(1, 3, 1, 1) + skip
    got = hip.to_rgb(x.to(DEV), wrgb.to(DEV), style.to(DEV), brgb.to(DEV), skip.to(DEV), 1 / math.sqrt(i))
    assert rel(got, want) < 1e-6
    # maps whose pixel count is not a multiple of four (cropped goal maps: 5 x 7): the one-pixel-per-thread kernel
    xo, so = x[:, :, :5, :7].contiguous(), skip[:, :, :5, :7].contiguous()
    want = torch.einsum('bci,bihw->bchw', wm, xo) + brgb.view(1, 3, 1, 1) + so
    got = hip.to_rgb(xo.to(DEV), wrgb.to(DEV), style.to(DEV), brgb.to(DEV), so.to(DEV), 1 / math.sqrt(i))
    assert rel(got, want) < 1e-6
    got = hip.to_rgb(xo.to(DEV), wrgb.to(DEV), style.to(DEV), None, None, 1 / math.sqrt(i))
    assert rel(got, want - brgb.view(1, 3, 1, 1) - so) < 1e-6


@pytest.mark.parametrize('ow', [8, 16, 32])
@pytest.mark.parametrize('planes', [(1, 3), (5, 13), (3, 64), (250, 8)])
def test_blur_noise_act_on_small_maps_packs_planes(ow, planes):
    """Outputs of 8^2 / 16^2 / 32^2 (layers 3, 5, 7) take blur_noise_act_small_kernel: 64 / 16 / 4 planes per workgroup.
    Against the oracle's upfirdn2d + noise + FusedLeakyReLU, with and without noise / bias / the post factor, plane counts
    that do not fill the last workgroup; and BIT FOR BIT against the tile kernel, which the same data takes when it is
    presented as a (2 OW) x OW map whose lower half is never looked at (no: the tile kernel needs the real rows) --
    so against the tile kernel on the map zero-extended to 2 OW x OW outputs, whose first OW rows see the same taps."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, c = planes
    rs = numpy.random.RandomState(ow + b)
    k4 = R.make_kernel([1, 3, 3, 1]) * 4
    k4[1, 2] += 0.05                                                    # asymmetric: a flipped kernel cannot pass
    wide = torch.from_numpy(rs.randn(b, c, ow + 1, ow + 1).astype('float32'))
    bias = torch.from_numpy(rs.randn(c).astype('float32'))
    nw = torch.tensor([0.41])
    noise = R.noise_rows(b, ow * ow)
    post = torch.from_numpy((1 + 0.3 * rs.randn(b, c)).astype('float32'))
    want = R.fused_leaky_relu(R.upfirdn2d(wide, k4, pad=(1, 1)) + nw * noise.view(b, 1, ow, ow), bias)
    got = hip.blur_noise_act(wide.to(DEV), k4.to(DEV), noise.to(DEV), nw.to(DEV), bias.to(DEV))
    assert rel(got, want) < 1e-6
    plain = hip.blur_noise_act(wide.to(DEV), k4.to(DEV), None, None, None)
    assert rel(plain, R.upfirdn2d(wide, k4, pad=(1, 1))) < 1e-6
    scaled = hip.blur_noise_act(wide.to(DEV), k4.to(DEV), noise.to(DEV), nw.to(DEV), bias.to(DEV), post_scale=post.to(DEV))
    assert torch.equal(scaled, got * post.to(DEV)[:, :, None, None])
    # the tile kernel on the same planes extended downwards by zeros: its first OW - 2 output rows read the same inputs
    tall = torch.zeros(b, c, 2 * ow + 1, ow + 1)
    tall[:, :, :ow + 1] = wide
    ntall = torch.zeros(b, 2 * ow * ow)
    ntall[:, :ow * ow] = noise
    ref = hip.blur_noise_act(tall.to(DEV), k4.to(DEV), ntall.to(DEV), nw.to(DEV), bias.to(DEV))
    assert torch.equal(ref[:, :, :ow - 2], got[:, :, :ow - 2])


@pytest.mark.parametrize('channels,batch,h,w', [(512, 10, 32, 32), (512, 3, 4, 4), (128, 2, 64, 64),
                                                (64, 4, 16, 16), (32, 2, 32, 32), (96, 2, 8, 8)])
def test_second_moment_and_channel_sums(channels, batch, h, w):
    from rewriting_amd import hip
    from rewriting_amd.utils import runningstats
    rs = numpy.random.RandomState(channels + h)
    acts = torch.from_numpy((rs.randn(batch, channels, h, w) + 0.3).astype('float32'))
    rows = acts.permute(0, 2, 3, 1).reshape(-1, channels)
    want = (rows.double().t() @ rows.double())
    r1, r2 = runningstats.RunningSecondMoment(), runningstats.RunningSecondMoment()
    for _ in range(2):                      # accumulate twice: += semantics
        r1.add_nchw(acts.to(DEV))
        r2.add(rows.to(DEV))
    assert r1.count == r2.count == 2 * rows.shape[0]
    for r in (r1, r2):
        assert rel(r.mom2, 2 * want) < 2e-6
        assert (r.mom2 - r.mom2.t()).abs().max().item() == 0.0          # mirrored exactly
    assert rel(r1.moment(), want / rows.shape[0]) < 2e-6
    sums = hip.channel_sums(acts.to(DEV), nchw=True, square_input=True).cpu().double()
    sq = rows.double() ** 2
    assert rel(sums[0], sq.sum(0)) < 1e-6 and rel(sums[1], (sq ** 2).sum(0)) < 1e-6
    sums2 = hip.channel_sums(rows.to(DEV)).cpu().double()
    assert rel(sums2[0], rows.double().sum(0)) < 1e-5 and rel(sums2[1], sq.sum(0)) < 1e-6


def test_projection_kernel():
    from rewriting_amd import hip
    from oracle import restatement as R
    rs = numpy.random.RandomState(3)
    for o, i, r in [(512, 512, 1), (64, 128, 3), (32, 48, 5)]:
        w = torch.from_numpy(rs.randn(1, o, i, 3, 3).astype('float32'))
        d = torch.linalg.qr(torch.from_numpy(rs.randn(i, r).astype('float32')))[0].t().contiguous()
        want = R.projected_conv(w, d)
        assert rel(hip.project_weight(w.to(DEV), d.to(DEV)), want) < 1e-5
        base = torch.from_numpy(rs.randn(1, o, i, 3, 3).astype('float32'))
        wd = w.to(DEV).clone()
        hip.project_weight(wd, d.to(DEV), base=base.to(DEV), out=wd)       # in place
        assert rel(wd, base + want) < 1e-5


GRAD_CASES = [  # batch, in_ch, out_ch, h, w  (forward kernels: in_ch % 16 == 0, out_ch % 32 == 0)
    (1, 16, 32, 8, 8), (3, 48, 96, 5, 9), (2, 64, 64, 16, 16), (1, 512, 512, 6, 7), (2, 128, 64, 32, 32),
    (1, 32, 160, 4, 4),
]


@pytest.mark.parametrize('upsample', [False, True])
@pytest.mark.parametrize('case', GRAD_CASES)
def test_conv_weight_and_input_gradients_match_autograd_of_the_oracle(case, upsample):
    """rw_conv_wgrad_f32 (split-K MFMA GEMM over batch x positions, gather of the stride-1 / stride-2-transposed
    columns, per-(image, channel) factors on both operands), rw_rowdot_f32, and the backward-to-input formed from
    the forward kernels on transposed weights -- i.e. grad.DemodConv end to end -- against torch.autograd of
    oracle/restatement.py's demod_conv on the host, for the map, the weight (both terms) and the style."""
    from rewriting_amd import hip
    from rewriting_amd.utils.stylegan2 import models
    from oracle import restatement as R
    b, i, o, h, w = case
    rs = numpy.random.RandomState(7 + b + i)
    x = torch.from_numpy(rs.randn(b, i, h, w).astype('float32'))
    wt = torch.from_numpy(rs.randn(1, o, i, 3, 3).astype('float32'))
    st = torch.from_numpy((1 + 0.3 * rs.randn(b, i)).astype('float32'))
    m = models.DemodulatedConv2dF(i, o, 3, upsample=upsample).to(DEV)
    with torch.no_grad():
        m.weight.copy_(wt.to(DEV))
    xd, sd_ = x.to(DEV).requires_grad_(True), st.to(DEV).requires_grad_(True)
    y = m(models.DataBag(fmap=xd, style=sd_)).fmap
    x2, st2, w2 = (t.clone().requires_grad_(True) for t in (x, st, wt))
    y2 = R.demod_conv(x2, st2, w2, upsample=upsample)
    g = torch.from_numpy(rs.randn(*y2.shape).astype('float32'))
    assert rel(y, y2) < 1e-5
    (y * g.to(DEV)).sum().backward()
    (y2 * g).sum().backward()
    for name, got, want in (('d fmap', xd.grad, x2.grad), ('d weight', m.weight.grad, w2.grad), ('d style', sd_.grad, st2.grad)):
        assert rel(got, want) < 2e-5, (name, rel(got, want))
        assert (got.cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item()), name
    # the raw kernel with both per-channel factors, against the same gradient taken on pre-scaled operands
    gs = torch.from_numpy((1 + 0.2 * rs.randn(b, o)).astype('float32')).to(DEV)
    xs = torch.from_numpy((1 + 0.2 * rs.randn(b, i)).astype('float32')).to(DEV)
    # ... and on channel counts no other kernel of the library takes
    if case == GRAD_CASES[0]:
        gg = torch.from_numpy(rs.randn(2, 5, 7 if not upsample else 15, 9 if not upsample else 19).astype('float32'))
        xx = torch.from_numpy(rs.randn(2, 7, 7, 9).astype('float32'))
        wz = torch.zeros(5, 7, 3, 3, requires_grad=True)
        yy = torch.nn.functional.conv_transpose2d(xx, wz.transpose(0, 1), stride=2) if upsample else \
            torch.nn.functional.conv2d(xx, wz, padding=1)
        (yy * gg).sum().backward()
        assert rel(hip.conv_wgrad(gg.to(DEV), xx.to(DEV), upsample), wz.grad) < 1e-5
    a = hip.conv_wgrad(g.to(DEV), x.to(DEV), upsample, scale=0.37, gscale=gs, xscale=xs)
    c = hip.conv_wgrad(g.to(DEV) * gs[:, :, None, None], x.to(DEV) * xs[:, :, None, None], upsample, scale=0.37)
    assert rel(a, c) < 1e-6
    r = hip.rowdot(y.detach(), g.to(DEV))
    assert rel(r, (y2.detach() * g).sum((2, 3))) < 1e-5


def test_other_dtypes_are_refused_with_the_fp32_only_message():
    """The B3 boundary is fp32 only (the reference's pybind modules also dispatch half and double): the wrappers say so
    instead of converting."""
    from rewriting_amd import hip
    x = torch.randn(2, 4, 8, 8, device=DEV)
    for bad in (x.double(), x.half()):
        with pytest.raises(RuntimeError, match='fp32 only'):
            hip.fused_bias_act(bad, torch.zeros(4, device=DEV, dtype=bad.dtype), None, 3, 0, 0.2, 1.0)
        with pytest.raises(RuntimeError, match='fp32 only'):
            hip.pixel_norm(bad.reshape(2, -1))


# ---- direct sums on the 16-bit matrix pipe (rw_dconv.hip): the direct fp32 kernels' bars, not the F(4x4,3x3) ones
DIRECT16_CASES = [(1, 16, 32, 16, 32), (2, 64, 64, 16, 64), (1, 32, 32, 48, 96), (1, 128, 128, 32, 64), (3, 48, 96, 16, 32),
                  (1, 512, 128, 16, 32), (1, 64, 64, 512, 512), (1, 32, 32, 1024, 1024), (1, 128, 128, 256, 256)]


@pytest.mark.parametrize('ver', ['one-role', 'specialised'])
@pytest.mark.parametrize('wm', [0, 1, 2, 4])
@pytest.mark.parametrize('case', DIRECT16_CASES)
def test_direct16_conv_matches_direct_fp32_kernel_and_oracle(case, wm, ver, monkeypatch):
    """hip.conv3x3_direct16 (direct sum, operands split into exact f16 pairs, v_mfma_f32_16x16x32_f16) against the direct
    fp32 kernel, the oracle with the full epilogue and float64 -- every workgroup shape (32 / 64 / 128 out-channels)."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    assert hip.dconv_supported(o, i, h, w)
    monkeypatch.setenv('RW_DCONV_V', '1' if ver == 'one-role' else '2')
    if ver == 'specialised' and b == 3:
        monkeypatch.setenv('RW_DCONV_GRID', '5')           # runs of several tiles, across images, ragged
    if wm:
        if o % (32 * wm):
            pytest.skip('out_ch not a multiple of the workgroup shape')
        monkeypatch.setenv('RW_DCONV_WM', str(wm))
    x, wt, style = _conv_inputs(*case, seed=141)
    rs = numpy.random.RandomState(142)
    x = x * torch.from_numpy(numpy.exp(1.5 * rs.randn(1, i, 1, 1)).astype('float32'))
    bias = torch.from_numpy(rs.randn(o).astype('float32'))
    nw = torch.tensor([0.2])
    noise = torch.from_numpy(rs.randn(b, h * w).astype('float32'))
    s = 1 / math.sqrt(i * 9)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    pk = hip.pack_conv_weight_direct16(wt.to(DEV))
    wp = hip.pack_conv_weight(wt.to(DEV), 0)
    ymax = hip.new_bound(b * o * h * w, DEV).fill_(3e38)
    plain = hip.conv3x3_direct16(x.to(DEV), pk, o, s, style=style.to(DEV), demod=dm, y_amax=ymax)
    assert hip.bound_value(ymax) == plain.abs().max().item()
    direct = hip.conv3x3(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm, impl=0)
    scale = direct.abs().max().item()
    assert (plain - direct).abs().max().item() < 2e-5 * scale, (plain - direct).abs().max().item() / scale
    assert rel(plain, direct) < 3e-6, rel(plain, direct)
    loose = hip.conv3x3_direct16(x.to(DEV), pk, o, s, style=style.to(DEV), demod=dm, x_amax=hip.absmax(x.to(DEV)) * 37.0)
    assert rel(loose, plain) < 2e-6, rel(loose, plain)
    args = dict(style=style.to(DEV), demod=dm, noise=noise.to(DEV), noise_w=nw.to(DEV), bias=bias.to(DEV), act=True)
    got = hip.conv3x3_direct16(x.to(DEV), pk, o, s, **args)
    same = hip.conv3x3(x.to(DEV), wp, o, s, impl=0, **args)
    assert (got - same).abs().max().item() < 2e-5 * max(1.0, same.abs().max().item())
    if b * i * o * h * w <= 2 ** 32:
        key = style[:, :, None, None] * x
        conv = R.demod_conv(key, style, wt, upsample=False)
        want = R.fused_leaky_relu(conv + nw * noise.view(b, 1, h, w), bias)
        assert rel(got, want) < 3e-6, rel(got, want)
        ref = torch.nn.functional.conv2d(key.double(), wt[0].double(), padding=1) * s * dm.cpu().double()[:, :, None, None]
        e_w = ((plain.cpu().double() - ref).norm() / ref.norm()).item()
        assert e_w < 3e-6, e_w
    if o == 32 and wm in (0, 1):                            # ToRGB in the epilogue (the feature map is not written)
        assert hip.dconv_to_rgb_supported(o, i, h, w)
        wrgb = torch.from_numpy(rs.randn(3, o).astype('float32')).to(DEV)
        srgb = torch.from_numpy((1 + 0.3 * rs.randn(b, o)).astype('float32')).to(DEV)
        brgb = torch.from_numpy(rs.randn(3).astype('float32')).to(DEV)
        skip = torch.from_numpy(rs.randn(b, 3, h, w).astype('float32')).to(DEV)
        want_rgb = hip.to_rgb(got, wrgb, srgb, brgb, skip, 1 / math.sqrt(o))
        y, rgb = hip.conv3x3_direct16_to_rgb(x.to(DEV), pk, o, s, wrgb, srgb, brgb, skip, 1 / math.sqrt(o), **args)
        assert y is None and rel(rgb, want_rgb) < 2e-6, rel(rgb, want_rgb)
        _, rgb2 = hip.conv3x3_direct16_to_rgb(x.to(DEV), pk, o, s, wrgb, srgb, None, None, 1 / math.sqrt(o), **args)
        assert rel(rgb2, want_rgb - skip - brgb.view(1, 3, 1, 1)) < 1e-5


UP_DIRECT16_CASES = [(2, 16, 16, 8, 32), (1, 64, 32, 16, 64), (1, 128, 64, 8, 128), (3, 32, 16, 24, 64), (1, 512, 64, 8, 64),
                     (1, 48, 48, 8, 32), (1, 64, 32, 512, 512)]


@pytest.mark.parametrize('ver', ['one-role', 'specialised'])
@pytest.mark.parametrize('case', UP_DIRECT16_CASES)
def test_direct16_one_pass_upsampling_conv_matches_conv_then_blur(case, ver, monkeypatch):
    """hip.conv_transpose3x3s2_blur_direct16 (the four output-parity phases of conv_transpose (*) blur as direct sums on the
    16-bit matrix pipe, noise + bias + leaky ReLU + post scale in the epilogue) against the two-pass route of the same
    library (direct transposed conv -> blur_noise_act) and the oracle, at the direct kernels' bars."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    assert hip.dconv_transpose_blur_supported(o, i, h, w)
    monkeypatch.setenv('RW_DCONV_V', '1' if ver == 'one-role' else '2')
    if ver == 'specialised' and b > 1:
        monkeypatch.setenv('RW_DCONV_GRID', '7')
    x, wt, style = _conv_inputs(*case, seed=161)
    rs = numpy.random.RandomState(162)
    x = x * torch.from_numpy(numpy.exp(1.0 * rs.randn(1, i, 1, 1)).astype('float32'))
    s = 1 / math.sqrt(i * 9)
    k1 = torch.tensor([1., 3., 3., 1.])
    k4 = k1[:, None] * k1[None, :]
    k4 = (k4 / k4.sum() * 4).to(DEV)
    noise = torch.from_numpy(rs.randn(b, 1, 2 * h, 2 * w).astype('float32')).to(DEV)
    nw = torch.tensor([0.37], device=DEV)
    bias = torch.from_numpy(rs.randn(o).astype('float32')).to(DEV)
    post = torch.from_numpy((1 + 0.3 * rs.randn(b, o)).astype('float32')).to(DEV)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    wp = hip.pack_conv_weight(wt.to(DEV), 1)
    wide = hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm,
                                   impl=0 if i % 16 == 0 and o % 32 == 0 else 1)
    pk = hip.pack_conv_transpose_blur_weight_direct16(wt.to(DEV), k4)
    results = []
    for kw in (dict(noise=noise, noise_w=nw, bias=bias, act=True), dict(), dict(post_scale=post)):
        want = hip.blur_noise_act(wide, k4, kw.get('noise'), kw.get('noise_w'), kw.get('bias'), kw.get('post_scale'))
        ymax = hip.new_bound(b * o * 4 * h * w, DEV).fill_(3e38)
        got = hip.conv_transpose3x3s2_blur_direct16(x.to(DEV), pk, o, s, style=style.to(DEV), demod=dm, y_amax=ymax, **kw)
        assert got.shape == want.shape == (b, o, 2 * h, 2 * w)
        assert hip.bound_value(ymax) == got.abs().max().item()
        scale = want.abs().max().item()
        assert (got - want).abs().max().item() < 2e-5 * scale, (got - want).abs().max().item() / scale
        assert rel(got, want) < 3e-6, rel(got, want)
        results.append((kw, got.cpu()))
    assert b * i * o * h * w <= 2 ** 29
    key = style[:, :, None, None] * x
    blur = R.upfirdn2d(R.demod_conv(key, style, wt, upsample=True), k4.cpu(), pad=(1, 1))
    for kw, got in results[:2]:
        ref = R.fused_leaky_relu(blur + nw.cpu() * noise.cpu(), bias.cpu()) if kw else blur
        assert rel(got, ref) < 5e-6, rel(got, ref)
        assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


# ---------------------------------------------------------------------------------------
# Bounds and weight scales of the split-operand kernels (round 5): per-wave slots stored plainly + one reduction launch,
# per-lane vector loads in the consumers, weight scales by value.
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize('n', [1, 3, 4, 5, 1023, 4096, 65537, 2 ** 20 + 7, 3 * 2 ** 22])
def test_absmax_bound_is_exact_for_any_length(n):
    from rewriting_amd import hip
    x = torch.randn(n, generator=torch.Generator().manual_seed(n)) * 3
    if n > 2:
        x[n // 3] = -17.5
    b = hip.absmax(x.to(DEV))
    assert b.numel() == hip.bound_floats(0)
    lanes = b[:hip.BOUND_LANES].cpu()
    assert torch.isfinite(lanes).all() and (lanes >= 0).all()
    assert hip.bound_value(b) == x.abs().max().item()


@pytest.mark.parametrize('b,c,out', [(2, 32, 64), (3, 8, 128), (1, 16, 40), (2, 4, 96), (5, 64, 16), (2, 512, 8), (1, 8, 4)])
def test_blur_noise_act_reports_the_bound_of_what_it_wrote(b, c, out):
    """The generic kernel (a slot per workgroup), the small-map kernels (a measuring pass) and tiles too small for a slot
    each (the same): the bound equals max |result|, post_scale included, from a POISONED buffer."""
    from rewriting_amd import hip
    rs = numpy.random.RandomState(out + c)
    wide = torch.from_numpy(rs.randn(b, c, out + 1, out + 1).astype('float32')).to(DEV)
    k1 = torch.tensor([1., 3., 3., 1.])
    k4 = k1[:, None] * k1[None, :]
    k4 = (k4 / k4.sum() * 4).to(DEV)
    noise = torch.from_numpy(rs.randn(b, out * out).astype('float32')).to(DEV)
    nw = torch.tensor([0.3], device=DEV)
    bias = torch.from_numpy(rs.randn(c).astype('float32')).to(DEV)
    post = torch.from_numpy((1 + 0.5 * rs.randn(b, c)).astype('float32')).to(DEV)
    for ps in (None, post):
        ymax = hip.new_bound(b * c * out * out, DEV).fill_(3e38)
        got = hip.blur_noise_act(wide, k4, noise, nw, bias, post_scale=ps, y_amax=ymax)
        same = hip.blur_noise_act(wide, k4, noise, nw, bias, post_scale=ps)
        assert torch.equal(got, same)
        assert hip.bound_value(ymax) == got.abs().max().item()
    with pytest.raises(ValueError, match='new_bound'):
        hip.blur_noise_act(wide, k4, noise, nw, bias, y_amax=torch.zeros(1, device=DEV))


@pytest.mark.parametrize('o,i', [(32, 8), (64, 64), (512, 512)])
def test_split_weight_scales_travel_by_value(o, i):
    """Every split packing: u_scale = rw_split_weight_scale(max |U|) with max |U| measured on the device and read back
    ONCE; it rides on the packed tensor (rw_u_inv) and in the trailer no kernel reads; a copy of the tensor is refused."""
    from rewriting_amd import hip
    rs = numpy.random.RandomState(o + i)
    wt = torch.from_numpy(rs.randn(1, o, i, 3, 3).astype('float32') * 2.5)
    k1 = torch.tensor([1., 3., 3., 1.])
    k4 = k1[:, None] * k1[None, :]
    k4 = (k4 / k4.sum() * 4).to(DEV)
    G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
    U = torch.einsum('ab,oibc,dc->oiad', G, wt[0].double(), G)
    scale_of = lambda m: float(hip.lib().rw_split_weight_scale(float(m)))
    uf = hip.pack_conv_weight_wino4(wt.to(DEV), split=True)
    su = 1.0 / uf.rw_u_inv
    assert su in (scale_of(U.abs().max()), scale_of(U.abs().max() * (1 + 1e-6)), scale_of(U.abs().max() * (1 - 1e-6)))
    assert uf[-4:].cpu().tolist() == [1.0 / su, su, 0.0, 0.0]
    # the words are the f16 pairs of U su: none overflows, the largest uses the top binade
    words = uf[:-4].view(torch.int32).cpu()
    hi = (words & 0xffff).to(torch.int16).view(torch.float16).float()
    assert torch.isfinite(hi).all() and 2.0 ** 14 <= hi.abs().max().item() < 2.0 ** 15
    packs = [uf]
    if i % 8 == 0 and o % 16 == 0:
        packs.append(hip.pack_conv_transpose_weight_wino(wt.to(DEV), split=True))
    packs.append(hip.pack_conv_transpose_blur_weight_wino4(wt.to(DEV), k4, split=True))
    if i % 16 == 0:
        d16 = hip.pack_conv_weight_direct16(wt.to(DEV))
        assert 1.0 / d16.rw_u_inv == scale_of(wt.abs().max())
        packs += [d16, hip.pack_conv_transpose_blur_weight_direct16(wt.to(DEV), k4)]
    for pk in packs:
        s_ = 1.0 / pk.rw_u_inv
        assert s_ == 2.0 ** round(math.log2(s_))
        assert pk[-4:].cpu().tolist() == [pk.rw_u_inv, s_, 0.0, 0.0]
    x = torch.randn(1, i, 8, 64).to(DEV)
    if hip.wino4_supported(o, i, 8, 64):
        with pytest.raises(ValueError, match='rw_u_inv'):
            hip.conv3x3_wino4(x, uf.clone(), o, 1.0)


TCONV_CASES = [(1, 16, 16, 16, 32), (2, 64, 32, 16, 64), (1, 32, 16, 32, 32), (1, 128, 64, 16, 32), (3, 48, 48, 32, 64),
               (1, 512, 32, 16, 32), (2, 256, 128, 32, 32), (1, 64, 32, 512, 512)]


TCONV_FORMS = {'two-workgroups-per-cu': '8', 'one-workgroup-per-cu': '16', 'specialised-persistent': '0',
               'pipelined-persistent': '2', 'thirty-two-out-channels': '32', 'twelve-wave-persistent': '12',
               'automatic': None}


@pytest.mark.parametrize('form', sorted(TCONV_FORMS))
@pytest.mark.parametrize('case', TCONV_CASES)
def test_fused_transposed_conv_and_blur_matches_conv_then_blur(case, form, monkeypatch):
    """hip.conv_transpose3x3s2_blur_fused (rw_tconv.hip: the transposed convolution as a direct sum on the 16-bit matrix
    pipe at its own multiply count, its (2H+1)^2 result in LDS, the blur from there, noise + bias + leaky ReLU + post
    scale in the epilogue) against the two-pass route of the same library (direct fp32 transposed conv -> blur_noise_act)
    and the oracle (utils/stylegan2/models.py:313-316,275-281 through oracle/restatement.py), at the direct kernels'
    bars; image borders, tile borders (h, w beyond one 16 x 32 tile), 16 .. 512 input channels, a loose bound -- in the
    kernel's three forms (RW_TCONV_TY: two 4-wave workgroups per CU, one 8-wave workgroup, the specialised persistent one;
    that one also with a grid of five workgroups: runs of many tiles that cross images, bit-identical to the full grid)."""
    from rewriting_amd import hip
    from oracle import restatement as R
    b, i, o, h, w = case
    if TCONV_FORMS[form] is None:
        monkeypatch.delenv('RW_TCONV_TY', raising=False)
    else:
        monkeypatch.setenv('RW_TCONV_TY', TCONV_FORMS[form])
    assert hip.tconv_blur_supported(o, i, h, w)
    x, wt, style = _conv_inputs(*case, seed=171)
    rs = numpy.random.RandomState(172)
    x = x * torch.from_numpy(numpy.exp(1.0 * rs.randn(1, i, 1, 1)).astype('float32'))
    s = 1 / math.sqrt(i * 9)
    k1 = torch.tensor([1., 3., 3., 1.])
    k4 = k1[:, None] * k1[None, :]
    k4 = (k4 / k4.sum() * 4).to(DEV)
    noise = torch.from_numpy(rs.randn(b, 1, 2 * h, 2 * w).astype('float32')).to(DEV)
    nw = torch.tensor([0.37], device=DEV)
    bias = torch.from_numpy(rs.randn(o).astype('float32')).to(DEV)
    post = torch.from_numpy((1 + 0.3 * rs.randn(b, o)).astype('float32')).to(DEV)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    wp = hip.pack_conv_weight(wt.to(DEV), 1)
    wide = hip.conv_transpose3x3s2(x.to(DEV), wp, o, s, style=style.to(DEV), demod=dm,
                                   impl=0 if i % 16 == 0 and o % 32 == 0 else 1)
    pk = hip.pack_conv_weight_direct16(wt.to(DEV))
    results = []
    for kw in (dict(noise=noise, noise_w=nw, bias=bias, act=True), dict(), dict(post_scale=post)):
        want = hip.blur_noise_act(wide, k4, kw.get('noise'), kw.get('noise_w'), kw.get('bias'), kw.get('post_scale'))
        ymax = hip.new_bound(b * o * 4 * h * w, DEV).fill_(3e38)
        got = hip.conv_transpose3x3s2_blur_fused(x.to(DEV), pk, k4, o, s, style=style.to(DEV), demod=dm, y_amax=ymax, **kw)
        assert got.shape == want.shape == (b, o, 2 * h, 2 * w)
        assert torch.isfinite(got).all()
        assert hip.bound_value(ymax) == got.abs().max().item()
        scale = want.abs().max().item()
        assert (got - want).abs().max().item() < 2e-5 * scale, (got - want).abs().max().item() / scale
        assert rel(got, want) < 3e-6, rel(got, want)
        if form in ('specialised-persistent', 'pipelined-persistent', 'twelve-wave-persistent'):
            monkeypatch.setenv('RW_TCONV_GRID', '5')
            again = hip.conv_transpose3x3s2_blur_fused(x.to(DEV), pk, k4, o, s, style=style.to(DEV), demod=dm, **kw)
            monkeypatch.delenv('RW_TCONV_GRID')
            assert torch.equal(again, got)
        results.append((kw, got.cpu()))
    loose = hip.conv_transpose3x3s2_blur_fused(x.to(DEV), pk, k4, o, s, style=style.to(DEV), demod=dm,
                                               x_amax=hip.absmax(x.to(DEV)) * 37.0)
    assert rel(loose, results[1][1]) < 2e-6
    assert b * i * o * h * w <= 2 ** 29
    key = style[:, :, None, None] * x
    blur = R.upfirdn2d(R.demod_conv(key, style, wt, upsample=True), k4.cpu(), pad=(1, 1))
    for kw, got in results[:2]:
        ref = R.fused_leaky_relu(blur + nw.cpu() * noise.cpu(), bias.cpu()) if kw else blur
        assert rel(got, ref) < 5e-6, rel(got, ref)
        assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('form', sorted(TCONV_FORMS))
def test_fused_transposed_conv_and_blur_with_a_fir_that_is_no_outer_product(form, monkeypatch):
    """The reference's FIR is always make_kernel([1, 3, 3, 1]) (an outer product: the strip walk); any other 4 x 4 kernel
    takes the 16-tap form of the epilogue.  Against blur_noise_act on the direct fp32 transposed convolution."""
    from rewriting_amd import hip
    if TCONV_FORMS[form] is None:
        monkeypatch.delenv('RW_TCONV_TY', raising=False)
    else:
        monkeypatch.setenv('RW_TCONV_TY', TCONV_FORMS[form])
    b, i, o, h, w = 2, 64, 32, 16, 64
    x, wt, style = _conv_inputs(b, i, o, h, w, seed=181)
    rs = numpy.random.RandomState(182)
    s = 1 / math.sqrt(i * 9)
    k4 = torch.from_numpy((0.25 + 0.1 * rs.randn(4, 4)).astype('float32')).to(DEV)
    noise = torch.from_numpy(rs.randn(b, 1, 2 * h, 2 * w).astype('float32')).to(DEV)
    nw = torch.tensor([0.21], device=DEV)
    bias = torch.from_numpy(rs.randn(o).astype('float32')).to(DEV)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), style.to(DEV))
    wide = hip.conv_transpose3x3s2(x.to(DEV), hip.pack_conv_weight(wt.to(DEV), 1), o, s, style=style.to(DEV), demod=dm, impl=0)
    want = hip.blur_noise_act(wide, k4, noise, nw, bias)
    got = hip.conv_transpose3x3s2_blur_fused(x.to(DEV), hip.pack_conv_weight_direct16(wt.to(DEV)), k4, o, s,
                                             style=style.to(DEV), demod=dm, noise=noise, noise_w=nw, bias=bias, act=True)
    assert (got - want).abs().max().item() < 2e-5 * want.abs().max().item()
    assert rel(got, want) < 3e-6, rel(got, want)
    from oracle import restatement as R
    ref = R.upfirdn2d(R.demod_conv(style[:, :, None, None] * x, style, wt, upsample=True), k4.cpu(), pad=(1, 1))
    ref = R.fused_leaky_relu(ref + nw.cpu() * noise.cpu(), bias.cpu())
    assert rel(got.cpu(), ref) < 5e-6, rel(got.cpu(), ref)


@pytest.mark.parametrize('case', [(2, 64, 32, 16, 32), (1, 64, 64, 32, 64), (2, 128, 128, 16, 64), (1, 512, 512, 16, 32),
                                  (3, 32, 256, 32, 32), (64, 64, 64, 16, 32), (9, 64, 64, 32, 128)])
@pytest.mark.parametrize('on_load', [False, True])
def test_direct_sum_conv_that_leaves_the_to_rgb_sums_matches_separate_kernels(case, on_load):
    """rw_dconv3x3_rgb_partial_f32 (the stride-1 direct sum on the 16-bit pipe that also leaves, per 32 out-channels, the
    channel sums of the ToRGB reading its result: ToRGBF.forward, models.py:639-655) + rw_rgb_combine_f32 against
    rw_dconv3x3_f32 followed by rw_to_rgb_f32: the feature map bit for bit, the image at the direct kernels' bar; 32 / 64 /
    128-channel workgroups, one and several partials, with and without the style on load (with it, on maps of 64 | columns: the
    specialised persistent kernels, over several tiles per workgroup and images), bias / skip absent."""
    from rewriting_amd import hip
    b, i, o, h, w = case
    assert hip.dconv_supported(o, i, h, w)
    x, wt, style = _conv_inputs(*case, seed=311)
    rs = numpy.random.RandomState(312)
    noise = torch.from_numpy(rs.randn(b, h * w).astype('float32')).to(DEV)
    nw = torch.tensor([0.2]).to(DEV)
    bias = torch.from_numpy(rs.randn(o).astype('float32')).to(DEV)
    wrgb = torch.from_numpy(rs.randn(3, o).astype('float32')).to(DEV)
    srgb = torch.from_numpy((1 + 0.3 * rs.randn(b, o)).astype('float32')).to(DEV)
    brgb = torch.from_numpy(rs.randn(3).astype('float32')).to(DEV)
    skip = torch.from_numpy(rs.randn(b, 3, h, w).astype('float32')).to(DEV)
    s = 1 / math.sqrt(i * 9)
    st = style.to(DEV)
    dm = hip.demod(hip.weight_sqsum(wt.to(DEV), s), st)
    xin = x.to(DEV) if on_load else (x * style[:, :, None, None]).to(DEV)
    pk = hip.pack_conv_weight_direct16(wt.to(DEV))
    args = dict(style=st if on_load else None, demod=dm, noise=noise, noise_w=nw, bias=bias, act=True, x_amax=hip.absmax(xin))
    ymax_a, ymax_b = hip.new_bound(b * o * h * w, DEV), hip.new_bound(b * o * h * w, DEV)
    fmap = hip.conv3x3_direct16(xin, pk, o, s, y_amax=ymax_a, **args)
    want = hip.to_rgb(fmap, wrgb, srgb, brgb, skip, 1 / math.sqrt(o))
    y, part = hip.conv3x3_direct16_rgb_partial(xin, pk, o, s, wrgb, srgb, 1 / math.sqrt(o), y_amax=ymax_b, **args)
    assert part.shape == (o // 32, b, 3, h, w)
    # (with a style on load, 64 | out-channels and 64 | columns BOTH take their specialised kernels -- dconv_ws_w2(_rgbp): round 6)
    assert torch.equal(y, fmap) and hip.bound_value(ymax_a) == hip.bound_value(ymax_b)
    fmap = y
    want = hip.to_rgb(fmap, wrgb, srgb, brgb, skip, 1 / math.sqrt(o))
    rgb = hip.rgb_combine(part, brgb, skip)
    assert rgb.shape == want.shape and torch.isfinite(rgb).all()
    assert rel(rgb, want) < 2e-6, rel(rgb, want)
    assert (rgb - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
    bare = hip.rgb_combine(part, None, None)
    assert rel(bare, want - skip - brgb.view(1, 3, 1, 1)) < 1e-5
    # the einsum the reference computes (models.py:649-655), in float64
    ref = torch.einsum('co,bo,bohw->bchw', wrgb.double().cpu(), srgb.double().cpu(), fmap.double().cpu()) / math.sqrt(o)
    ref = ref + brgb.double().cpu().view(1, 3, 1, 1) + skip.double().cpu()
    assert rel(rgb.double().cpu(), ref) < 2e-6
