"""TEST INFRASTRUCTURE -- a torch-CPU stand-in for the tensor-level kernel wrappers of
``rewriting_amd.hip``, so that the HOST logic above the C ABI (module tree, fusion switch,
rewriter orchestration, solver driver) can be exercised by the CPU test-suite against the
golden fixtures.  It lives under tests/ and is installed by the ``emulated_hip`` fixture only;
the product package never imports it and has no CPU path for these ops.

Each function mirrors the *interface* of the wrapper it replaces and computes the result with
plain torch ops (delegating to oracle/restatement.py where that has the op already).
"""
import math

import torch
import torch.nn.functional as F

from oracle import restatement as R

SQRT2 = 2 ** 0.5


def fused_bias_act(x, b, ref, act, grad, alpha, scale):
    b = b if (b is not None and b.numel()) else None
    ref = ref if (ref is not None and ref.numel()) else None
    return R.fused_bias_act(x.detach(), None if b is None else b.detach(), ref, act, grad, alpha, scale)


def bias_grad(g):
    dims = [0] + list(range(2, g.ndim))
    return g.sum(dims)


def upfirdn2d_major(x, k, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    return R.upfirdn2d_major(x.detach(), k.detach(), up_x, up_y, down_x, down_y, px0, px1, py0, py1)


def pixel_norm(x, eps=1e-8):
    x = x.detach()
    return x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + eps)


def equal_linear(x, weight, bias, w_scale, b_scale, act=False, alpha=0.2, act_scale=SQRT2):
    out = F.linear(x.detach(), weight.detach() * w_scale)
    if bias is not None:
        out = out + bias.detach() * b_scale
    if act:
        out = F.leaky_relu(out, alpha) * act_scale
    return out


def adjust_latent(w, avg, n_latent, psi):
    w = w.detach()
    if avg is not None:
        w = avg + psi * (w - avg)
    return w.unsqueeze(1).repeat(1, n_latent, 1)


def style_mul(x, style):
    return style.detach()[:, :, None, None] * x.detach()


def weight_sqsum(weight, w_scale):
    w = weight.detach().reshape(weight.shape[-4], weight.shape[-3], -1)
    return ((w * w_scale) ** 2).sum(-1)


def demod(wsq, style, eps=1e-8):
    return torch.rsqrt((style.detach() ** 2) @ wsq.t() + eps)


_UP_ORDER = [0, 2, 6, 8, 1, 7, 3, 5, 4]


def pack_conv_weight(weight, mode):
    w = weight.detach().reshape(weight.shape[-4], weight.shape[-3], 9)   # [o][i][tap]
    wp = w.permute(2, 1, 0).contiguous()                                   # [tap][i][o]
    if mode == 1:
        wp = wp[_UP_ORDER].contiguous()
    return wp


def _unpack(wp, mode):
    if mode == 1:
        inv = [0] * 9
        for slab, tap in enumerate(_UP_ORDER):
            inv[tap] = slab
        wp = wp[inv]
    o, i = wp.shape[2], wp.shape[1]
    return wp.permute(2, 1, 0).reshape(o, i, 3, 3)


def conv3x3(x, wp, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None,
            act=False, impl=0):
    x = x.detach()
    if style is not None:
        x = x * style.detach()[:, :, None, None]
    y = F.conv2d(x, _unpack(wp, 0), padding=1) * w_scale
    if demod is not None:
        y = y * demod[:, :, None, None]
    b, _, h, w = y.shape
    if noise is not None:
        y = y + noise_w.detach().reshape(1) * noise.reshape(b, 1, h, w)
    if act:
        y = F.leaky_relu(y + bias.detach().view(1, -1, 1, 1), 0.2) * SQRT2
    return y


def conv_transpose3x3s2(x, wp, out_ch, w_scale, style=None, demod=None, impl=0, out=None):
    x = x.detach()
    if style is not None:
        x = x * style.detach()[:, :, None, None]
    w = _unpack(wp, 1)                                   # [o][i][3][3]
    y = F.conv_transpose2d(x, w.transpose(0, 1), stride=2) * w_scale
    if demod is not None:
        y = y * demod[:, :, None, None]
    if out is None:
        return y
    if impl == 8:                                        # the border row / column strips only
        out[:, :, -1, :] = y[:, :, -1, :]
        out[:, :, :, -1] = y[:, :, :, -1]
    elif impl == 7:                                      # the quad tiles only
        out[:, :, :-1, :-1] = y[:, :, :-1, :-1]
    else:
        out.copy_(y)
    return out


# ---- the split-operand ("H16") kernels: every operand of the matrix products as the sum of two f16 numbers of the value
# scaled by a power of two (rw_wino4.hip, rw_upwino.hip); the emulation rounds the operands the same way and multiplies
# in fp32
class _Split:
    def __init__(self, handle):
        self.handle = handle


def _pow2_above(t):
    """e with max |t| < 2^e (the kernels read it off the exponent field)"""
    m = float(t.abs().max())
    return 0 if m == 0 else math.frexp(m)[1]


def _f16_pair(t, scale_exp):
    ts = t * (2.0 ** scale_exp)
    h = ts.half().float()
    return (h + (ts - h).half().float()) * (2.0 ** -scale_exp)


BOUND_LANES = 64


def bound_floats(n_elems):
    return BOUND_LANES + 2048 + int(n_elems) // 1024 + 1     # rw_bound_floats


def new_bound(n_elems, device):
    """hip.new_bound: BOUND_LANES floats whose maximum is the bound + the producer's slots (garbage here, as on the
    device: nothing may read them)."""
    return torch.full((bound_floats(n_elems),), float('nan'))


def bound_value(bound):
    return float(bound[:BOUND_LANES].max())


def _bound_of(x_amax, x):
    """what a kernel's prologue reduces: the lanes of the bound (never its slots)"""
    if x_amax is None:                     # hip._amax_in: measured by the wrapper, through the module's own hip.absmax
        from rewriting_amd import hip
        x_amax = hip.absmax(x)
    return x_amax.detach()[:BOUND_LANES].max().reshape(1)


def _fill_bound(y_amax, y, measured=False):
    """a producer's report: lanes whose maximum is max |y| (the device spreads its slots' maxima over them)"""
    if y_amax is not None:
        assert y_amax.numel() >= bound_floats(0 if measured else y.numel())        # rw_absmax_f32: at most 2048 slots
        lanes = torch.zeros(BOUND_LANES)
        lanes[int(y.numel()) % BOUND_LANES] = y.detach().abs().max()
        y_amax[:BOUND_LANES] = lanes


def absmax(x):
    out = new_bound(0, 'cpu')
    _fill_bound(out, x, measured=True)
    return out


def conv_transpose_wino_split_supported(out_ch, in_ch, height, width):
    return width % 32 == 0 and height % 4 == 0 and in_ch % 8 == 0 and 16 <= in_ch <= 512 and out_ch % 32 == 0


def pack_conv_transpose_weight_wino(weight, split=False):
    h = pack_conv_weight(weight, 1)                      # opaque handle
    return _Split(h) if split else h


def conv_transpose3x3s2_wino(x, uf, out_ch, w_scale, style=None, demod=None, out=None, x_amax=None):
    """The arithmetic of rw_upwino.hip in torch fp32: per axis the even outputs by F(2,2) (points d0-d1, d1, d2-d1
    against w2, w2+w0, w0), the odd ones by their single tap; 25 products per 2x2 block of quads.  Writes the
    quads y < H, x < W of `out` only."""
    x = x.detach()
    split = isinstance(uf, _Split)
    if split:
        # |T| <= 4 max |x style| < 2^(e + 2): the kernel scales by 2^(12 - e); the weights by 2^(15 - eu), eu from the
        # largest of the 25 points (at most 4 max |w|: the exact bound is taken here, one binade cannot matter)
        smax = 1.0 if style is None else float(style.detach().abs().max())
        ev = 12 - _pow2_above(_bound_of(x_amax, x) * smax)
        uf = uf.handle
    if style is not None:
        x = x * style.detach()[:, :, None, None]
    w = _unpack(uf, 1)                                   # [o][i][ky][kx]
    if split:
        eu = 15 - _pow2_above(w.abs().max().reshape(1) * 4)
    b, c, h, wd = x.shape
    y = out if out is not None else torch.zeros(b, out_ch, 2 * h + 1, 2 * wd + 1)
    xp = F.pad(x, (1, 1, 1, 1))
    d = [[xp[:, :, r:r + h:2, cc:cc + wd:2] for cc in range(3)] for r in range(3)]      # windows of the 2x2 blocks

    def pts(kind, a0, a1, a2):
        return [a0 - a1, a1, a2 - a1] if kind == 'E' else [a1, a2]

    def wts(kind, g0, g1, g2):
        return [g2, g2 + g0, g0] if kind == 'E' else [g1, g1]

    def outs(kind, m):
        return [m[0] + m[1], m[1] + m[2]] if kind == 'E' else [m[0], m[1]]
    for py, kv in ((0, 'E'), (1, 'O')):
        for px, kh in ((0, 'E'), (1, 'O')):
            rows = [pts(kv, d[0][cc], d[1][cc], d[2][cc]) for cc in range(3)]            # rows[cc][a]
            V = [pts(kh, rows[0][a], rows[1][a], rows[2][a]) for a in range(len(rows[0]))]   # V[a][b]
            gv = wts(kv, w[:, :, 0], w[:, :, 1], w[:, :, 2])                              # [a] -> (o,i,kx)
            U = [wts(kh, g[:, :, 0], g[:, :, 1], g[:, :, 2]) for g in gv]                  # U[a][b] (o,i)
            if split:
                V = [[_f16_pair(v, ev) for v in row] for row in V]
                U = [[_f16_pair(u, eu) for u in row] for row in U]
            M = [[torch.einsum('oi,nihw->nohw', U[a][bb], V[a][bb]) for bb in range(len(V[0]))] for a in range(len(V))]
            cols = [outs(kh, M[a]) for a in range(len(M))]
            for bq in range(2):
                col = outs(kv, [cols[a][bq] for a in range(len(M))])
                for aq in range(2):
                    val = col[aq] * w_scale
                    if demod is not None:
                        val = val * demod[:, :, None, None]
                    y[:, :, 2 * aq + py:2 * h:4, 2 * bq + px:2 * wd:4] = val
    return y


def noise_add(x, noise, noise_w):
    b, c, h, w = x.shape
    return x.detach() + noise_w.detach().reshape(1) * noise.reshape(b, 1, h, w)


def blur_noise_act(x, k4, noise, noise_w, bias, post_scale=None, y_amax=None):
    y = R.upfirdn2d(x.detach(), k4, pad=(1, 1))
    b, c, h, w = y.shape
    if noise is not None:
        y = y + noise_w.detach().reshape(1) * noise.reshape(b, 1, h, w)
    if bias is not None:
        y = F.leaky_relu(y + bias.detach().view(1, -1, 1, 1), 0.2) * SQRT2
    if post_scale is not None:
        y = y * post_scale.detach()[:, :, None, None]
    _fill_bound(y_amax, y)
    return y


def to_rgb(x, weight, style, bias, skip, w_scale):
    wm = (w_scale * weight.detach()[None] * style.detach()[:, None, :])      # (B,3,C)
    y = torch.einsum('bci,bihw->bchw', wm, x.detach())
    if bias is not None:
        y = y + bias.detach().view(1, 3, 1, 1)
    if skip is not None:
        y = y + skip
    return y


def second_moment_accumulate(mom2, a, nchw=False):
    a = a.detach()
    if nchw:
        a = a.permute(0, 2, 3, 1).reshape(-1, a.shape[1])
    mom2 += a.t() @ a
    return mom2


def channel_sums(a, nchw=False, square_input=False):
    a = a.detach()
    if nchw:
        a = a.permute(0, 2, 3, 1).reshape(-1, a.shape[1])
    if square_input:
        a = a * a
    return torch.stack([a.sum(0), (a * a).sum(0)])


def project_weight(w, context, base=None, out=None):
    res = R.projected_conv(w.detach(), context.detach())
    if base is not None:
        res = res + base
    if out is not None:
        out.copy_(res)
        return out
    return res


def solve_ksplit(out_ch, in_ch, h, w):
    return 2


_solvers = {}


def register_solver(solver):
    import ctypes
    _solvers[ctypes.addressof(solver.problem)] = solver


def solve_step(problem, project):
    """One iteration of the solve in torch autograd, on the driver's own tensors."""
    import ctypes
    s = _solvers[ctypes.addressof(problem)]
    it = int(s.counter.item()) + 1
    s.counter.fill_(it)
    _solve_iteration(s, problem, it, project)


def solve_run(problem, it_begin, it_end, niter, piter, low_rank_insert, lpart):
    """rw_solve_run_f32: iterations [it_begin, it_end), projecting where the reference does."""
    import ctypes
    s = _solvers[ctypes.addressof(problem)]
    for it in range(it_begin, it_end):
        _solve_iteration(s, problem, it, bool(low_rank_insert) and (it % piter == 0 or it == niter - 1))


def _solve_iteration(s, problem, it, project):
    W = s._w
    O, I = W.shape[1:3]
    up = bool(problem.upsample)
    Wg = W.clone().requires_grad_(True)
    with torch.enable_grad():
        out = R.demod_conv(s.key[None], s.style[None], Wg, up)
        if s.bias is not None:              # bias None: the target is the demodulated conv alone
            if up:
                out = R.upfirdn2d(out, s.blur_k, pad=(1, 1))
            out = out + s.noise_w * s.noise.view(1, 1, *out.shape[2:])
            out = R.fused_leaky_relu(out, s.bias)
        loss = F.l1_loss(s.val[None], out)
        loss.backward()
    s.losses[it] = loss.detach()
    dW = Wg.grad
    if s.linear:
        g = torch.einsum('goiyx,di->godyx', dW, s.context)[0].reshape(O, -1, 9)
        m, v = s.exp_avg.view(-1)[:g.numel()].view_as(g), s.exp_avg_sq.view(-1)[:g.numel()].view_as(g)
        m += (g - m) * (1 - 0.9)
        v.mul_(0.999).add_((1 - 0.999) * g * g)
        s.lam += (-s.step_size[it] * m) / (v.sqrt() / s.bc2_sqrt[it] + 1e-8)
        W.copy_(s.ortho + torch.einsum('ody,di->oiy', s.lam, s.context).reshape(W.shape))
        return
    if s.low_rank_gradient:
        dW = R.projected_conv(dW, s.context)
    m, v = s.exp_avg, s.exp_avg_sq
    m += (dW - m) * (1 - 0.9)
    v.mul_(0.999).add_((1 - 0.999) * dW * dW)
    W += (-s.step_size[it] * m) / (v.sqrt() / s.bc2_sqrt[it] + 1e-8)
    if project:
        W.copy_(s.ortho + R.projected_conv(W, s.context))


def pack_conv_weight_bf16x3(weight):
    return pack_conv_weight(weight, 0)          # opaque handle; the emulated product is exact


def conv3x3_bf16x6(x, wb, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None,
                   act=False):
    return conv3x3(x, wb, out_ch, w_scale, style=style, demod=demod, noise=noise, noise_w=noise_w, bias=bias,
                   act=act)


def pack_conv_weight_wino(weight):
    return pack_conv_weight(weight, 0)          # opaque handle


_WG = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
_WBT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
_WAT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def conv3x3_wino(x, uf, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None, act=False):
    """The arithmetic of rw_wino.hip in torch fp32: U = G g G^T, V = B^T d B on 4x4 tiles of stride 2,
    M = sum_i U.V per transform point, Y = A^T M A; then the shared epilogue."""
    x = x.detach()
    if style is not None:
        x = x * style.detach()[:, :, None, None]
    wt = _unpack(uf, 0)
    b, i, h, w = x.shape
    U = torch.einsum('ab,oibc,dc->oiad', _WG, wt, _WG)
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum('ab,nithbc,dc->nithad', _WBT, d, _WBT)
    M = torch.einsum('oiad,nithad->nothad', U, V)
    Y = torch.einsum('ab,nothbc,dc->nothad', _WAT, M, _WAT)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(b, out_ch, h, w) * w_scale
    if demod is not None:
        y = y * demod[:, :, None, None]
    if noise is not None:
        y = y + noise_w.detach().reshape(1) * noise.reshape(b, 1, h, w)
    if act:
        y = F.leaky_relu(y + bias.detach().view(1, -1, 1, 1), 0.2) * SQRT2
    return y


def pack_conv_weight_wino4(weight, split=False):
    h = pack_conv_weight(weight, 0)             # opaque handle
    return _Split(h) if split else h


_W4G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                     [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
_W4BT = torch.tensor([[4., 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                      [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]])
_W4AT = torch.tensor([[1., 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]])


def conv3x3_wino4(x, uf, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None, act=False,
                  x_amax=None, y_amax=None):
    """The arithmetic of rw_wino4.hip in torch fp32: F(4x4,3x3) on 6x6 tiles of stride 4; with split weights the
    transformed operands rounded to pairs of f16 numbers as the kernels on the 16-bit matrix pipe do."""
    x = x.detach()
    split = isinstance(uf, _Split)
    if split:
        smax = 1.0 if style is None else float(style.detach().abs().max())
        ev = 8 - _pow2_above(_bound_of(x_amax, x) * smax)     # |B^T d B| <= 100 max < 2^(e + 7)
        uf = uf.handle
    if style is not None:
        x = x * style.detach()[:, :, None, None]
    wt = _unpack(uf, 0)
    b, i, h, w = x.shape
    U = torch.einsum('ab,oibc,dc->oiad', _W4G, wt, _W4G)
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 6, 4).unfold(3, 6, 4)
    V = torch.einsum('ab,nithbc,dc->nithad', _W4BT, d, _W4BT)
    if split:
        U, V = _f16_pair(U, 15 - _pow2_above(U)), _f16_pair(V, ev)
    M = torch.einsum('oiad,nithad->nothad', U, V)
    Y = torch.einsum('ab,nothbc,dc->nothad', _W4AT, M, _W4AT)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(b, out_ch, h, w) * w_scale
    if demod is not None:
        y = y * demod[:, :, None, None]
    if noise is not None:
        y = y + noise_w.detach().reshape(1) * noise.reshape(b, 1, h, w)
    if act:
        y = F.leaky_relu(y + bias.detach().view(1, -1, 1, 1), 0.2) * SQRT2
    _fill_bound(y_amax, y)
    return y


def pack_conv_transpose_blur_weight_wino4(weight, k4, split=False):
    h = (pack_conv_weight(weight, 1), k4.detach().clone())             # opaque handle
    return _Split(h) if split else h


def conv_transpose3x3s2_blur_wino4(x, uf, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None,
                                   bias=None, act=False, post_scale=None, x_amax=None, y_amax=None):
    """The arithmetic of rw_wino4.hip's conv_up_wino36_kernel in torch fp32: the four output-parity phases of
    conv_transpose(stride 2) (*) blur as 4 * out_ch virtual channels of the F(4x4,3x3) convolution, pixel-shuffled."""
    split = isinstance(uf, _Split)
    wp, k4 = uf.handle if split else uf
    w = _unpack(wp, 1)                                   # [o][i][3][3]
    o, i = w.shape[:2]
    kf = torch.flip(k4, [0, 1])
    g6 = torch.zeros(o, i, 6, 6)
    for c in range(4):
        for d in range(4):
            # g6[t + 2] += k'[c] w[t - 1 + c]  ->  w[ky] lands at t = ky + 1 - c
            g6[:, :, 3 - c:6 - c, 3 - d:6 - d] += kf[c, d] * w
    wv = torch.zeros(4 * o, i, 3, 3)
    for py in range(2):
        for px in range(2):
            for a in range(3):
                for b in range(3):
                    wv[2 * py + px::4, :, a, b] = g6[:, :, 4 - 2 * a + py, 4 - 2 * b + px]
    z = conv3x3_wino4(x, pack_conv_weight_wino4(wv.reshape(1, 4 * o, i, 3, 3), split), 4 * o, w_scale, style=style,
                      demod=None if demod is None else demod.repeat_interleave(4, dim=1), x_amax=x_amax)
    n, _, h, wd = z.shape
    y = z.reshape(n, o, 2, 2, h, wd).permute(0, 1, 4, 2, 5, 3).reshape(n, o, 2 * h, 2 * wd)
    if noise is not None:
        y = y + noise_w.detach().reshape(1) * noise.reshape(n, 1, 2 * h, 2 * wd)
    if act:
        y = F.leaky_relu(y + bias.detach().view(1, -1, 1, 1), 0.2) * SQRT2
    if post_scale is not None:
        y = y * post_scale.detach()[:, :, None, None]
    _fill_bound(y_amax, y)
    return y


# ---- the direct sums on the 16-bit matrix pipe (rw_dconv.hip): the operands rounded to f16 pairs (|x style| scaled below
# 2^14, the weights below 2^15), the sum itself in fp32
class _Direct16:
    def __init__(self, handle):
        self.handle = handle


def pack_conv_weight_direct16(weight):
    return _Direct16(pack_conv_weight(weight, 0))


def conv3x3_direct16(x, wp, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None, act=False,
                     x_amax=None, y_amax=None):
    x = x.detach()
    smax = 1.0 if style is None else float(style.detach().abs().max())
    ev = 14 - _pow2_above(_bound_of(x_amax, x) * smax)
    if style is not None:
        x = x * style.detach()[:, :, None, None]
    wt = _unpack(wp.handle, 0)
    b, i, h, w = x.shape
    y = F.conv2d(_f16_pair(x, ev), _f16_pair(wt, 15 - _pow2_above(wt)), padding=1) * w_scale
    if demod is not None:
        y = y * demod[:, :, None, None]
    if noise is not None:
        y = y + noise_w.detach().reshape(1) * noise.reshape(b, 1, h, w)
    if act:
        y = F.leaky_relu(y + bias.detach().view(1, -1, 1, 1), 0.2) * SQRT2
    _fill_bound(y_amax, y)
    return y


def conv3x3_direct16_to_rgb(x, wp, out_ch, w_scale, rgb_weight, rgb_style, rgb_bias, rgb_skip, rgb_scale, style=None,
                            demod=None, noise=None, noise_w=None, bias=None, act=False, x_amax=None):
    y = conv3x3_direct16(x, wp, out_ch, w_scale, style=style, demod=demod, noise=noise, noise_w=noise_w, bias=bias, act=act,
                         x_amax=x_amax)
    return None, to_rgb(y, rgb_weight, rgb_style, rgb_bias, rgb_skip, rgb_scale)


def pack_conv_transpose_blur_weight_direct16(weight, k4):
    return _Direct16((pack_conv_weight(weight, 1), k4.detach().clone()))


def conv_transpose3x3s2_blur_direct16(x, wp, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None,
                                      bias=None, act=False, post_scale=None, x_amax=None, y_amax=None):
    """rw_dconv.hip's one-pass upsampling layer: the four output-parity phases of conv_transpose(stride 2) (*) blur as
    4 * out_ch virtual channels of a direct 3x3 convolution, pixel-shuffled."""
    packed, k4 = wp.handle
    w = _unpack(packed, 1)                               # [o][i][3][3]
    o, i = w.shape[:2]
    kf = torch.flip(k4, [0, 1])
    g6 = torch.zeros(o, i, 6, 6)
    for c in range(4):
        for d in range(4):
            g6[:, :, 3 - c:6 - c, 3 - d:6 - d] += kf[c, d] * w
    wv = torch.zeros(4 * o, i, 3, 3)
    for py in range(2):
        for px in range(2):
            for a in range(3):
                for b in range(3):
                    wv[2 * py + px::4, :, a, b] = g6[:, :, 4 - 2 * a + py, 4 - 2 * b + px]
    z = conv3x3_direct16(x, pack_conv_weight_direct16(wv.reshape(1, 4 * o, i, 3, 3)), 4 * o, w_scale, style=style,
                         demod=None if demod is None else demod.repeat_interleave(4, dim=1), x_amax=x_amax)
    n, _, h, wd = z.shape
    y = z.reshape(n, o, 2, 2, h, wd).permute(0, 1, 4, 2, 5, 3).reshape(n, o, 2 * h, 2 * wd)
    if noise is not None:
        y = y + noise_w.detach().reshape(1) * noise.reshape(n, 1, 2 * h, 2 * wd)
    if act:
        y = F.leaky_relu(y + bias.detach().view(1, -1, 1, 1), 0.2) * SQRT2
    if post_scale is not None:
        y = y * post_scale.detach()[:, :, None, None]
    _fill_bound(y_amax, y)
    return y


def tconv_blur_supported(out_ch, in_ch, height, width):
    return out_ch % 16 == 0 and in_ch % 16 == 0 and 16 <= in_ch <= 512 and width % 32 == 0 and height % 16 == 0


def conv_transpose3x3s2_blur_fused(x, wp, k4, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None,
                                   bias=None, act=False, post_scale=None, x_amax=None, y_amax=None):
    """rw_tconv.hip: the transposed convolution itself as a direct sum with the operands rounded to f16 pairs, then the blur,
    noise, bias + leaky ReLU and the post scale."""
    x = x.detach()
    smax = 1.0 if style is None else float(style.detach().abs().max())
    ev = 14 - _pow2_above(_bound_of(x_amax, x) * smax)
    if style is not None:
        x = x * style.detach()[:, :, None, None]
    wt = _unpack(wp.handle, 0)                           # [o][i][3][3]
    z = F.conv_transpose2d(_f16_pair(x, ev), _f16_pair(wt, 15 - _pow2_above(wt)).transpose(0, 1), stride=2) * w_scale
    if demod is not None:
        z = z * demod[:, :, None, None]
    y = R.upfirdn2d(z, k4.detach(), pad=(1, 1))
    n, _, h2, w2 = y.shape
    if noise is not None:
        y = y + noise_w.detach().reshape(1) * noise.reshape(n, 1, h2, w2)
    if act:
        y = F.leaky_relu(y + bias.detach().view(1, -1, 1, 1), 0.2) * SQRT2
    if post_scale is not None:
        y = y * post_scale.detach()[:, :, None, None]
    _fill_bound(y_amax, y)
    return y


def conv_wgrad(g, x, upsample, scale=1.0, gscale=None, xscale=None):
    """d W of conv2d(x, W, pad 1) / conv_transpose2d(x, W^T, stride 2) given g = d L / d y, through torch's own
    autograd of the same op (rw_conv_wgrad_f32's definition)."""
    g, x = g.detach(), x.detach()
    if gscale is not None:
        g = g * gscale.detach()[:, :, None, None]
    if xscale is not None:
        x = x * xscale.detach()[:, :, None, None]
    o, i = g.shape[1], x.shape[1]
    w = torch.zeros(o, i, 3, 3, requires_grad=True)
    with torch.enable_grad():
        y = F.conv_transpose2d(x, w.transpose(0, 1), stride=2) if upsample else F.conv2d(x, w, padding=1)
        (y * g).sum().backward()
    return w.grad.detach() * scale


def rowdot(a, b):
    return (a.detach() * b.detach()).sum((-2, -1))


def install(monkeypatch):
    """Replaces the kernel wrappers of rewriting_amd.hip and makes host code take the
    'tensors live on the device' branches."""
    from rewriting_amd import hip
    from rewriting_amd.rewrite import hipsolve
    names = ['fused_bias_act', 'bias_grad', 'upfirdn2d_major', 'pixel_norm', 'equal_linear',
             'adjust_latent', 'style_mul', 'weight_sqsum', 'demod', 'pack_conv_weight', 'conv3x3',
             'conv_transpose3x3s2', 'noise_add', 'blur_noise_act', 'to_rgb', 'pack_conv_weight_bf16x3',
             'conv3x3_bf16x6', 'pack_conv_weight_wino', 'conv3x3_wino', 'pack_conv_weight_wino4', 'conv3x3_wino4', 'pack_conv_transpose_weight_wino', 'conv_transpose3x3s2_wino',
             'pack_conv_transpose_blur_weight_wino4', 'conv_transpose3x3s2_blur_wino4', 'absmax', 'new_bound',
             'bound_value', 'bound_floats',
             'pack_conv_weight_direct16', 'conv3x3_direct16', 'conv3x3_direct16_to_rgb',
             'pack_conv_transpose_blur_weight_direct16', 'conv_transpose3x3s2_blur_direct16',
             'conv_transpose_wino_split_supported', 'tconv_blur_supported', 'conv_transpose3x3s2_blur_fused',
             'second_moment_accumulate', 'channel_sums', 'project_weight', 'solve_ksplit',
             'solve_step', 'solve_run', 'conv_wgrad', 'rowdot']
    for n in names:
        monkeypatch.setattr(hip, n, globals()[n])
    monkeypatch.setattr(hip, 'on_device', lambda t: True)
    orig_init = hipsolve.Solver.__init__

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        register_solver(self)
    monkeypatch.setattr(hipsolve.Solver, '__init__', init)
    monkeypatch.setenv('RW_SOLVE_GRAPH', '0')
