"""Tensor-level wrappers over the C ABI (include/rewriting_hip.h).

PyTorch is used here for device memory and streams only: every wrapper allocates its output
with torch, passes ``data_ptr()`` and ``torch.cuda.current_stream().cuda_stream`` to the
library and returns the torch tensor -- the ownership rule of SURVEY.md section 8b.  Inputs
must be float32 tensors on a HIP device; anything else raises (no CPU path).
"""
import ctypes
import os
import math

import torch

from . import _lib
from ._lib import ConvEpilogue, SolveProblem, check

SQRT2 = 2 ** 0.5


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name='tensor'):
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('rewriting_amd: %s is on %s; the HIP kernels need a GPU tensor '
                           '(there is no CPU fallback)' % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError('rewriting_amd: %s is %s; the C ABI is fp32 only (RW_ERR_UNSUPPORTED, include/rewriting_hip.h: '
                           'the reference\'s pybind modules also dispatch half and double, this library does not) -- '
                           'cast the model / tensor with .float()' % (name, t.dtype))
    return t.detach().contiguous()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _opt(t, name):
    return None if t is None else _dev(t, name)


def lib():
    return _lib.load()


def on_device(t):
    """True when ``t`` lives on a HIP device, i.e. when the kernels (and only the kernels) must
    be used for it.  Host code branches on this, never on try/except around a kernel call."""
    return bool(t.is_cuda)


# ------------------------------------------------------------------ L1 native ops
def fused_bias_act(x, b, ref, act, grad, alpha, scale):
    """fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
    (utils/stylegan2/op/fused_bias_act.cpp:11-20); empty tensor = absent."""
    x = _dev(x, 'input')
    b = _dev(b, 'bias') if b is not None and b.numel() else None
    ref = _dev(ref, 'refer') if ref is not None and ref.numel() else None
    y = torch.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    size_b = b.numel() if b is not None else 1
    check(lib().rw_fused_bias_act_f32(_p(x), _p(b), _p(ref), _p(y), x.numel(), step_b, size_b,
                                      int(act), int(grad), float(alpha), float(scale), _stream()))
    return y


def bias_grad(g):
    g = _dev(g, 'grad')
    outer = g.shape[0]
    channels = g.shape[1] if g.ndim > 1 else 1
    inner = 1
    for d in g.shape[2:]:
        inner *= d
    gb = torch.empty(channels, device=g.device, dtype=g.dtype)
    check(lib().rw_bias_grad_f32(_p(g), _p(gb), outer, channels, inner, _stream()))
    return gb


def upfirdn2d_major(x, k, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    """upfirdn2d_op.upfirdn2d on (major, H, W, minor) (utils/stylegan2/op/upfirdn2d.cpp:12-22)."""
    x = _dev(x, 'input')
    k = _dev(k, 'kernel')
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h = (in_h * up_y + py0 + py1 - kh + down_y) // down_y
    out_w = (in_w * up_x + px0 + px1 - kw + down_x) // down_x
    y = torch.empty(major, max(out_h, 0), max(out_w, 0), minor, device=x.device, dtype=x.dtype)
    if y.numel():
        check(lib().rw_upfirdn2d_f32(_p(x), _p(k), _p(y), major, in_h, in_w, minor, kh, kw,
                                     up_x, up_y, down_x, down_y, px0, px1, py0, py1, _stream()))
    return y


# ------------------------------------------------------------------ generator pieces
def pixel_norm(x, eps=1e-8):
    x = _dev(x, 'latent')
    y = torch.empty_like(x)
    check(lib().rw_pixel_norm_f32(_p(x), _p(y), x.shape[0], x.shape[1], eps, _stream()))
    return y


def equal_linear(x, weight, bias, w_scale, b_scale, act=False, alpha=0.2, act_scale=SQRT2):
    """x may be a strided row view (latent[:, index]): last dim contiguous."""
    if not x.is_cuda:
        _dev(x, 'input')
    x = x.detach()
    if x.dtype != torch.float32 or x.ndim != 2 or x.stride(1) != 1:
        x = _dev(x.reshape(x.shape[0], -1), 'input')
    weight = _dev(weight, 'weight')
    bias = _opt(bias, 'bias')
    batch, in_dim = x.shape
    out_dim = weight.shape[0]
    y = torch.empty(batch, out_dim, device=x.device, dtype=torch.float32)
    x_stride = x.stride(0) if batch > 1 else in_dim
    check(lib().rw_equal_linear_f32(_p(x), _p(weight), _p(bias), _p(y), batch, in_dim, out_dim,
                                    max(x_stride, in_dim), float(w_scale), float(b_scale), int(bool(act)),
                                    float(alpha), float(act_scale), _stream()))
    return y


def adjust_latent(w, avg, n_latent, psi):
    w = _dev(w, 'latent')
    avg = _opt(avg, 'latent_avg')
    out = torch.empty(w.shape[0], n_latent, w.shape[1], device=w.device, dtype=w.dtype)
    check(lib().rw_adjust_latent_f32(_p(w), _p(avg), _p(out), w.shape[0], n_latent, w.shape[1],
                                     float(psi), _stream()))
    return out


def style_mul(x, style):
    x = _dev(x, 'fmap')
    style = _dev(style, 'style')
    b, c, h, w = x.shape
    y = torch.empty_like(x)
    check(lib().rw_style_mul_f32(_p(x), _p(style), _p(y), b, c, h * w, _stream()))
    return y


def weight_sqsum(weight, w_scale):
    weight = _dev(weight, 'weight')
    o, i = weight.shape[-4], weight.shape[-3]
    taps = weight.shape[-1] * weight.shape[-2]
    wsq = torch.empty(o, i, device=weight.device, dtype=weight.dtype)
    check(lib().rw_weight_sqsum_f32(_p(weight), _p(wsq), o, i, taps, float(w_scale), _stream()))
    return wsq


def demod(wsq, style, eps=1e-8):
    wsq = _dev(wsq, 'wsq')
    style = _dev(style, 'style')
    b = style.shape[0]
    o, i = wsq.shape
    out = torch.empty(b, o, device=wsq.device, dtype=wsq.dtype)
    check(lib().rw_demod_f32(_p(wsq), _p(style), _p(out), b, o, i, float(eps), _stream()))
    return out


def pack_conv_weight(weight, mode):
    weight = _dev(weight, 'weight')
    o, i = weight.shape[-4], weight.shape[-3]
    n = lib().rw_packed_conv_weight_elems(o, i, int(mode))
    if n <= 0:
        raise ValueError('bad weight shape / mode')
    wp = torch.empty(n, device=weight.device, dtype=weight.dtype)   # slab order, then fragment order
    check(lib().rw_pack_conv_weight_f32(_p(weight), _p(wp), o, i, int(mode), _stream()))
    return wp


def _check_packed(wp, out_ch, in_ch, mode):
    if wp.numel() != lib().rw_packed_conv_weight_elems(out_ch, in_ch, mode):
        raise ValueError('packed weight does not come from pack_conv_weight(%d x %d, mode %d)'
                         % (out_ch, in_ch, mode))


def _epilogue(style=None, demod=None, noise=None, noise_w=None, bias=None, act=False):
    keep = [_opt(style, 'style'), _opt(demod, 'demod'), _opt(noise, 'noise'),
            _opt(noise_w, 'noise weight'), _opt(bias, 'bias')]
    ep = ConvEpilogue(_p(keep[0]).value, _p(keep[1]).value, _p(keep[2]).value,
                      _p(keep[3]).value, _p(keep[4]).value, int(bool(act)))
    return ep, keep


def conv3x3(x, wp, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None,
            act=False, impl=0):
    x = _dev(x, 'fmap')
    packed, wp = wp, _dev(wp, 'packed weight')
    b, i, h, w = x.shape
    y = torch.empty(b, out_ch, h, w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    _check_packed(wp, out_ch, i, 0)
    check(lib().rw_conv3x3_f32(_p(x), _p(wp), _p(y), b, i, out_ch, h, w, float(w_scale),
                               ctypes.byref(ep), int(impl), _stream()))
    return y


def to_rgb_fusable(out_ch, in_ch, width):
    """Shapes rw_conv3x3_to_rgb_f32 takes: one wave holds every out-channel of its pixels."""
    return out_ch in (32, 64) and width >= 24 and in_ch % 16 == 0 and in_ch <= 1024


def conv3x3_to_rgb(x, wp, out_ch, w_scale, rgb_weight, rgb_style, rgb_bias, rgb_skip, rgb_scale, style=None,
                   demod=None, noise=None, noise_w=None, bias=None, act=False, store_fmap=False):
    """The styled convolution with ToRGB fused into its epilogue: returns (fmap or None, rgb image)."""
    x = _dev(x, 'fmap')
    packed, wp = wp, _dev(wp, 'packed weight')
    rgb_weight = _dev(rgb_weight, 'rgb weight').contiguous()
    rgb_style = _dev(rgb_style, 'rgb style').contiguous()
    rgb_bias = _opt(rgb_bias, 'rgb bias')
    rgb_skip = _opt(rgb_skip, 'rgb skip')
    b, i, h, w = x.shape
    if tuple(rgb_weight.shape) != (3, out_ch) or tuple(rgb_style.shape) != (b, out_ch):
        raise ValueError('rgb weight / style shapes')
    if rgb_skip is not None and tuple(rgb_skip.shape) != (b, 3, h, w):
        raise ValueError('rgb skip shape')
    _check_packed(wp, out_ch, i, 0)
    y = torch.empty(b, out_ch, h, w, device=x.device, dtype=x.dtype) if store_fmap else None
    rgb = torch.empty(b, 3, h, w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    from ._lib import RgbEpilogue
    re = RgbEpilogue(_p(rgb_weight).value, _p(rgb_style).value, _p(rgb_bias).value, _p(rgb_skip).value,
                     _p(rgb).value, float(rgb_scale))
    check(lib().rw_conv3x3_to_rgb_f32(_p(x), _p(wp), _p(y), b, i, out_ch, h, w, float(w_scale),
                                      ctypes.byref(ep), ctypes.byref(re), _stream()))
    return y, rgb


def wino_supported(out_ch, in_ch, height, width):
    """Shapes the Winograd F(2x2,3x3) stride-1 convolution takes (rw_conv3x3_wino_supported)."""
    return bool(lib().rw_conv3x3_wino_supported(int(out_ch), int(in_ch), int(height), int(width)))


def pack_conv_weight_wino(weight):
    weight = _dev(weight, 'weight')
    o, i = weight.shape[-4], weight.shape[-3]
    n = lib().rw_packed_conv_weight_wino_elems(o, i)
    if n <= 0:
        raise ValueError('no Winograd packing for a %d x %d weight' % (o, i))
    uf = torch.empty(n, device=weight.device, dtype=torch.float32)
    check(lib().rw_pack_conv_weight_wino_f32(_p(weight), _p(uf), o, i, _stream()))
    return uf


def conv3x3_wino(x, uf, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None, act=False):
    """Stride-1 3x3 convolution by Winograd F(2x2,3x3) in fp32; same arguments and epilogue as conv3x3."""
    x = _dev(x, 'fmap')
    packed, uf = uf, _dev(uf, 'packed weight')      # (the scale of a split packing rides on the caller's tensor)
    b, i, h, w = x.shape
    if uf.numel() != lib().rw_packed_conv_weight_wino_elems(out_ch, i):
        raise ValueError('packed weight does not come from pack_conv_weight_wino(%d x %d)' % (out_ch, i))
    y = torch.empty(b, out_ch, h, w, device=x.device, dtype=x.dtype)
    if style is not None and h == w and w in (4, 8):
        # whole 8^2 / 4^2 images per wave (several images per workgroup): the library takes exactly these maps already
        # multiplied by their style (style == NULL, see the header) -- the same product, rounded the same way, one tiny
        # launch earlier
        x, style = style_mul(x, _dev(style, 'style')), None
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    check(lib().rw_conv3x3_wino_f32(_p(x), _p(uf), _p(y), b, i, out_ch, h, w, float(w_scale), ctypes.byref(ep),
                                    _stream()))
    return y


def wino4_supported(out_ch, in_ch, height, width):
    """Shapes the Winograd F(4x4,3x3) stride-1 convolution takes (rw_conv3x3_wino4_supported)."""
    return bool(lib().rw_conv3x3_wino4_supported(int(out_ch), int(in_ch), int(height), int(width)))


BOUND_LANES = 64                # RW_BOUND_LANES of include/rewriting_hip.h


def bound_floats(n_elems):
    """Floats of the y_amax buffer of a producer whose result has n_elems floats (rw_bound_floats)."""
    return int(lib().rw_bound_floats(int(n_elems)))


def new_bound(n_elems, device):
    """An (uninitialised) y_amax buffer for a result of n_elems floats: BOUND_LANES floats that will hold the bound --
    their maximum is >= max |result| -- followed by the producer's per-wave slots ("a BOUND on a map" in
    include/rewriting_hip.h).  The same tensor is the x_amax of the convolution that reads the result.  Nothing in it is
    zeroed, raised with atomics or read back by a later launch through a device scalar: a producer's waves store their
    own slots plainly, one small launch reduces them, the launch boundary publishes the floats like any feature map."""
    return torch.empty(bound_floats(n_elems), device=device, dtype=torch.float32)


def bound_value(bound):
    """The number a bound stands for (host sync; tests and the packing of split weights, never the forward)."""
    return float(bound[:BOUND_LANES].max())


def absmax(x):
    """The bound of x (rw_absmax_f32): the x_amax of the split-operand kernels where the producer of x did not leave one
    behind."""
    x = _dev(x, 'tensor')
    out = new_bound(0, x.device)            # at most 2048 slots
    check(lib().rw_absmax_f32(_p(x), x.numel(), _p(out), _stream()))
    return out


def _amax_in(x, x_amax):
    """The bound a split-operand kernel gets: the caller's, or measured here."""
    if x_amax is None:
        return absmax(x)
    x_amax = _dev(x_amax, 'x_amax')
    if x_amax.numel() < BOUND_LANES:
        raise ValueError('x_amax must be a bound: at least %d floats (hip.new_bound / hip.absmax)' % BOUND_LANES)
    return x_amax


def _amax_out(y_amax, n_elems):
    if y_amax is None:
        return None
    y_amax = _dev(y_amax, 'y_amax')
    if y_amax.numel() < bound_floats(n_elems):
        raise ValueError('y_amax must come from hip.new_bound(%d, device): %d floats' % (n_elems, bound_floats(n_elems)))
    return y_amax


def _split_scale(measure, *args):
    """(u_scale, u_inv) of a split-operand packing: `measure` (a rw_*_absmax_f32 entry) leaves max |U| as a bound, the
    host reads it -- ONE sync per weight version, never inside a forward whose weights are packed -- and
    rw_split_weight_scale turns it into the power of two that the pack kernel and every convolution launch then get BY
    VALUE."""
    device = args[0].device
    bound = new_bound(0, device)
    check(measure(*[_p(a) if torch.is_tensor(a) else a for a in args], _p(bound), _stream()))
    u_scale = float(lib().rw_split_weight_scale(bound_value(bound)))
    return u_scale, 1.0 / u_scale


def _u_inv(packed):
    """1 / u_scale of a split-operand packing: it travels with the tensor the pack function returned."""
    try:
        return float(packed.rw_u_inv)
    except AttributeError:
        raise ValueError('split-operand packed weights must be the tensor a hip.pack_*(split=True) / pack_*_direct16 call '
                         'returned (its scale travels as the attribute rw_u_inv; a copy does not carry it)') from None


def pack_conv_weight_wino4(weight, split=False):
    """G g G^T of every filter in the fragment order of the F(4x4,3x3) kernels; split=True: every value as the pair
    of f16 numbers the kernels on the 16-bit matrix pipe multiply (rw_pack_conv_weight_wino4h_f32) -- the convolution
    functions tell the two formats apart by their size."""
    weight = _dev(weight, 'weight')
    o, i = weight.shape[-4], weight.shape[-3]
    n = (lib().rw_packed_conv_weight_wino4h_elems if split else lib().rw_packed_conv_weight_wino4_elems)(o, i)
    if n <= 0:
        raise ValueError('no F(4x4,3x3) packing for a %d x %d weight' % (o, i))
    uf = torch.empty(n, device=weight.device, dtype=torch.float32)
    if split:
        u_scale, uf.rw_u_inv = _split_scale(lib().rw_conv_weight_wino4h_absmax_f32, weight, o, i)
        check(lib().rw_pack_conv_weight_wino4h_f32(_p(weight), _p(uf), o, i, u_scale, _stream()))
    else:
        check(lib().rw_pack_conv_weight_wino4_f32(_p(weight), _p(uf), o, i, _stream()))
    return uf


def _wino4_split(uf, out_ch, in_ch, what='pack_conv_weight_wino4'):
    """Is uf the split (f16 pair) packing?  Raises when it is neither packing of this shape."""
    if uf.numel() == lib().rw_packed_conv_weight_wino4_elems(out_ch, in_ch):
        return False
    if uf.numel() == lib().rw_packed_conv_weight_wino4h_elems(out_ch, in_ch):
        return True
    raise ValueError('packed weight does not come from %s(%d x %d)' % (what, out_ch, in_ch))


def conv3x3_wino4(x, uf, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None, act=False,
                  x_amax=None, y_amax=None):
    """Stride-1 3x3 convolution by Winograd F(4x4,3x3) (~1e-5 relative error per layer: the default of the
    un-hooked whole-generator forward, never used by a hooked or sliced model -- models.conv_algo); same arguments
    and epilogue as conv3x3.  With weights from pack_conv_weight_wino4(split=True) the products run on the 16-bit
    matrix pipe (exact f16 operand split, fp32 accumulation): x_amax = the bound of x (hip.absmax(x), or the y_amax its
    producer filled; measured here when None), y_amax = hip.new_bound(y.numel(), device), which receives the bound of y."""
    x = _dev(x, 'fmap')
    packed, uf = uf, _dev(uf, 'packed weight')      # (the scale of a split packing rides on the caller's tensor)
    b, i, h, w = x.shape
    split = _wino4_split(uf, out_ch, i)
    y = torch.empty(b, out_ch, h, w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    if split:
        x_amax, y_amax = _amax_in(x, x_amax), _amax_out(y_amax, y.numel())
        check(lib().rw_conv3x3_wino4h_f32(_p(x), _p(uf), _p(y), b, i, out_ch, h, w, float(w_scale), ctypes.byref(ep),
                                          _u_inv(packed), _p(x_amax), _p(y_amax), _stream()))
        return y
    check(lib().rw_conv3x3_wino4_f32(_p(x), _p(uf), _p(y), b, i, out_ch, h, w, float(w_scale), ctypes.byref(ep),
                                     _stream()))
    return y


def conv3x3_wino_to_rgb(x, uf, out_ch, w_scale, rgb_weight, rgb_style, rgb_bias, rgb_skip, rgb_scale, style=None,
                        demod=None, noise=None, noise_w=None, bias=None, act=False, store_fmap=False):
    """conv3x3_wino with ToRGB in the epilogue (out_ch == 32): returns (fmap or None, rgb image)."""
    x = _dev(x, 'fmap')
    packed, uf = uf, _dev(uf, 'packed weight')      # (the scale of a split packing rides on the caller's tensor)
    rgb_weight = _dev(rgb_weight, 'rgb weight').contiguous()
    rgb_style = _dev(rgb_style, 'rgb style').contiguous()
    rgb_bias = _opt(rgb_bias, 'rgb bias')
    rgb_skip = _opt(rgb_skip, 'rgb skip')
    b, i, h, w = x.shape
    if tuple(rgb_weight.shape) != (3, out_ch) or tuple(rgb_style.shape) != (b, out_ch):
        raise ValueError('rgb weight / style shapes')
    if rgb_skip is not None and tuple(rgb_skip.shape) != (b, 3, h, w):
        raise ValueError('rgb skip shape')
    if uf.numel() != lib().rw_packed_conv_weight_wino_elems(out_ch, i):
        raise ValueError('packed weight does not come from pack_conv_weight_wino(%d x %d)' % (out_ch, i))
    y = torch.empty(b, out_ch, h, w, device=x.device, dtype=x.dtype) if store_fmap else None
    rgb = torch.empty(b, 3, h, w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    from ._lib import RgbEpilogue
    re = RgbEpilogue(_p(rgb_weight).value, _p(rgb_style).value, _p(rgb_bias).value, _p(rgb_skip).value,
                     _p(rgb).value, float(rgb_scale))
    check(lib().rw_conv3x3_wino_to_rgb_f32(_p(x), _p(uf), _p(y), b, i, out_ch, h, w, float(w_scale),
                                           ctypes.byref(ep), ctypes.byref(re), _stream()))
    return y, rgb


def wino4_to_rgb_supported(out_ch, in_ch, height, width):
    """Shapes conv3x3_wino4_to_rgb takes (rw_conv3x3_wino4_to_rgb_supported: out_ch == 32)."""
    return bool(lib().rw_conv3x3_wino4_to_rgb_supported(int(out_ch), int(in_ch), int(height), int(width)))


def conv3x3_wino4_to_rgb(x, uf, out_ch, w_scale, rgb_weight, rgb_style, rgb_bias, rgb_skip, rgb_scale, style=None,
                         demod=None, noise=None, noise_w=None, bias=None, act=False, x_amax=None):
    """conv3x3_wino4 with ToRGB in the epilogue (out_ch == 32): returns (None, rgb image); the feature map is not
    written.  Split weights and x_amax as in conv3x3_wino4."""
    x = _dev(x, 'fmap')
    packed, uf = uf, _dev(uf, 'packed weight')      # (the scale of a split packing rides on the caller's tensor)
    rgb_weight = _dev(rgb_weight, 'rgb weight').contiguous()
    rgb_style = _dev(rgb_style, 'rgb style').contiguous()
    rgb_bias = _opt(rgb_bias, 'rgb bias')
    rgb_skip = _opt(rgb_skip, 'rgb skip')
    b, i, h, w = x.shape
    if tuple(rgb_weight.shape) != (3, out_ch) or tuple(rgb_style.shape) != (b, out_ch):
        raise ValueError('rgb weight / style shapes')
    if rgb_skip is not None and tuple(rgb_skip.shape) != (b, 3, h, w):
        raise ValueError('rgb skip shape')
    split = _wino4_split(uf, out_ch, i)
    rgb = torch.empty(b, 3, h, w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    from ._lib import RgbEpilogue
    re = RgbEpilogue(_p(rgb_weight).value, _p(rgb_style).value, _p(rgb_bias).value, _p(rgb_skip).value,
                     _p(rgb).value, float(rgb_scale))
    if split:
        x_amax = _amax_in(x, x_amax)
        check(lib().rw_conv3x3_wino4h_to_rgb_f32(_p(x), _p(uf), b, i, out_ch, h, w, float(w_scale), ctypes.byref(ep),
                                                 ctypes.byref(re), _u_inv(packed), _p(x_amax), _stream()))
        return None, rgb
    check(lib().rw_conv3x3_wino4_to_rgb_f32(_p(x), _p(uf), b, i, out_ch, h, w, float(w_scale), ctypes.byref(ep),
                                            ctypes.byref(re), _stream()))
    return None, rgb


def bf16x6_supported(out_ch, in_ch, width):
    """Shapes the opt-in split-precision convolution takes (rw_conv3x3_bf16x6_f32)."""
    return width >= 24 and in_ch % 16 == 0 and in_ch <= 1024 and out_ch % 64 == 0


def pack_conv_weight_bf16x3(weight):
    weight = _dev(weight, 'weight')
    o, i = weight.shape[-4], weight.shape[-3]
    n = lib().rw_packed_conv_weight_bf16x3_bytes(o, i)
    if n <= 0:
        raise ValueError('no bf16x3 packing for a %d x %d weight' % (o, i))
    wb = torch.empty(n // 4, device=weight.device, dtype=torch.float32)     # opaque: bf16 fragments
    check(lib().rw_pack_conv_weight_bf16x3(_p(weight), _p(wb), o, i, _stream()))
    return wb


def conv3x3_bf16x6(x, wb, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None,
                   act=False):
    """Opt-in: the stride-1 convolution on the bf16 matrix pipe with exact three-way operand splits
    (six piece products, fp32 accumulation); same arguments and epilogue as conv3x3."""
    x = _dev(x, 'fmap')
    wb = _dev(wb, 'packed weight')
    b, i, h, w = x.shape
    if wb.numel() * 4 != lib().rw_packed_conv_weight_bf16x3_bytes(out_ch, i):
        raise ValueError('packed weight does not come from pack_conv_weight_bf16x3(%d x %d)' % (out_ch, i))
    y = torch.empty(b, out_ch, h, w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    check(lib().rw_conv3x3_bf16x6_f32(_p(x), _p(wb), _p(y), b, i, out_ch, h, w, float(w_scale),
                                      ctypes.byref(ep), _stream()))
    return y


def up_halo_applicable(out_ch, in_ch, width):
    """Shapes conv_transpose3x3s2 runs on the halo-tile kernel (halo_applicable() in rw_conv.hip)."""
    return (in_ch % 16 == 0 and in_ch <= 1024 and out_ch % 32 == 0
            and (width >= 24 or 9 <= width <= 16 or (5 <= width <= 8 and out_ch % 128 == 0)))


def up_strips_applicable(out_ch, in_ch):
    """Channel counts conv_transpose3x3s2(impl=8) -- the border row and column of the (2H+1) x (2W+1) map alone -- takes."""
    return in_ch % 16 == 0 and in_ch <= 1024 and out_ch % 32 == 0


def conv_transpose3x3s2(x, wp, out_ch, w_scale, style=None, demod=None, impl=0, out=None):
    x = _dev(x, 'fmap')
    packed, wp = wp, _dev(wp, 'packed weight')
    b, i, h, w = x.shape
    y = out if out is not None else torch.empty(b, out_ch, 2 * h + 1, 2 * w + 1, device=x.device, dtype=x.dtype)
    if tuple(y.shape) != (b, out_ch, 2 * h + 1, 2 * w + 1) or not y.is_contiguous():
        raise ValueError('out has the wrong shape')
    ep, keep = _epilogue(style, demod)
    _check_packed(wp, out_ch, i, 1)
    check(lib().rw_conv_transpose3x3s2_f32(_p(x), _p(wp), _p(y), b, i, out_ch, h, w, float(w_scale),
                                           ctypes.byref(ep), int(impl), _stream()))
    return y


def conv_transpose_wino_supported(out_ch, in_ch, height, width):
    """Shapes the F(2,2) transposed convolution takes (rw_conv_transpose3x3s2_wino_supported)."""
    return bool(lib().rw_conv_transpose3x3s2_wino_supported(int(out_ch), int(in_ch), int(height), int(width)))


def conv_transpose_wino_split_supported(out_ch, in_ch, height, width):
    """Shapes the split-operand (16-bit matrix pipe) form of conv_transpose3x3s2_wino takes."""
    return bool(lib().rw_conv_transpose3x3s2_winoh_supported(int(out_ch), int(in_ch), int(height), int(width)))


def pack_conv_transpose_weight_wino(weight, split=False):
    """The 25 F(2,2) points of every filter in fragment order; split=True: as pairs of f16 numbers for the kernel on
    the 16-bit matrix pipe (rw_pack_conv_transpose_winoh_f32) -- told apart by size."""
    weight = _dev(weight, 'weight')
    o, i = weight.shape[-4], weight.shape[-3]
    n = (lib().rw_packed_conv_transpose_winoh_elems if split else lib().rw_packed_conv_transpose_wino_elems)(o, i)
    if n <= 0:
        raise ValueError('no F(2,2) packing for a %d x %d transposed-conv weight' % (o, i))
    uf = torch.empty(n, device=weight.device, dtype=torch.float32)
    if split:
        u_scale, uf.rw_u_inv = _split_scale(lib().rw_conv_transpose_weight_winoh_absmax_f32, weight, o, i)
        check(lib().rw_pack_conv_transpose_winoh_f32(_p(weight), _p(uf), o, i, u_scale, _stream()))
    else:
        check(lib().rw_pack_conv_transpose_wino_f32(_p(weight), _p(uf), o, i, _stream()))
    return uf


def conv_transpose3x3s2_wino(x, uf, out_ch, w_scale, style=None, demod=None, out=None, x_amax=None):
    """The quads y < H, x < W of conv_transpose3x3s2 by F(2,2) (25 instead of 36 multiplies per 2x2 block of quads);
    output row 2H and column 2W are left to conv_transpose3x3s2(..., impl=8, out=...).  With weights from
    pack_conv_transpose_weight_wino(split=True) the products run on the 16-bit matrix pipe (exact f16 operand split,
    fp32 accumulation; x_amax = the bound of x as in conv3x3_wino4, measured here when None)."""
    x = _dev(x, 'fmap')
    packed, uf = uf, _dev(uf, 'packed weight')      # (the scale of a split packing rides on the caller's tensor)
    b, i, h, w = x.shape
    if uf.numel() == lib().rw_packed_conv_transpose_wino_elems(out_ch, i):
        split = False
    elif uf.numel() == lib().rw_packed_conv_transpose_winoh_elems(out_ch, i):
        split = True
    else:
        raise ValueError('packed weight does not come from pack_conv_transpose_weight_wino(%d x %d)' % (out_ch, i))
    y = out if out is not None else torch.empty(b, out_ch, 2 * h + 1, 2 * w + 1, device=x.device, dtype=x.dtype)
    if tuple(y.shape) != (b, out_ch, 2 * h + 1, 2 * w + 1) or not y.is_contiguous():
        raise ValueError('out has the wrong shape')
    style = _opt(style, 'style')
    demod = _opt(demod, 'demod')
    if split:
        x_amax = _amax_in(x, x_amax)
        check(lib().rw_conv_transpose3x3s2_winoh_f32(_p(x), _p(uf), _p(y), b, i, out_ch, h, w, float(w_scale),
                                                     _p(style), _p(demod), _u_inv(packed), _p(x_amax), _stream()))
        return y
    if style is not None and h == w and w in (4, 8):
        # whole 8^2 / 4^2 images per wave (several images per workgroup): the library takes exactly these maps already
        # multiplied by their style -- the same product, rounded the same way, one tiny launch earlier
        x, style = style_mul(x, style), None
    check(lib().rw_conv_transpose3x3s2_wino_f32(_p(x), _p(uf), _p(y), b, i, out_ch, h, w, float(w_scale), _p(style),
                                                _p(demod), _stream()))
    return y


def conv_transpose_blur_wino4_supported(out_ch, in_ch, height, width):
    """Shapes the one-pass upsampling StyledConv takes (rw_conv_transpose_blur_wino4_supported)."""
    return bool(lib().rw_conv_transpose_blur_wino4_supported(int(out_ch), int(in_ch), int(height), int(width)))


def pack_conv_transpose_blur_weight_wino4(weight, k4, split=False):
    """F(4x4,3x3) weights of the four output-parity phases of conv_transpose(stride 2) followed by the 4x4 FIR k4;
    split=True: as f16 pairs (pack_conv_weight_wino4)."""
    weight = _dev(weight, 'weight')
    k4 = _dev(k4, 'blur kernel').contiguous()
    if tuple(k4.shape) != (4, 4):
        raise ValueError('the blur kernel must be 4 x 4')
    o, i = weight.shape[-4], weight.shape[-3]
    n = (lib().rw_packed_conv_transpose_blur_wino4h_elems if split
         else lib().rw_packed_conv_transpose_blur_wino4_elems)(o, i)
    if n <= 0:
        raise ValueError('no F(4x4,3x3) phase packing for a %d x %d transposed-conv weight' % (o, i))
    uf = torch.empty(n, device=weight.device, dtype=torch.float32)
    if split:
        u_scale, uf.rw_u_inv = _split_scale(lib().rw_conv_transpose_blur_weight_wino4h_absmax_f32, weight, k4, o, i)
        check(lib().rw_pack_conv_transpose_blur_weight_wino4h_f32(_p(weight), _p(k4), _p(uf), o, i, u_scale, _stream()))
    else:
        check(lib().rw_pack_conv_transpose_blur_weight_wino4_f32(_p(weight), _p(k4), _p(uf), o, i, _stream()))
    return uf


def conv_transpose3x3s2_blur_wino4(x, uf, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None,
                                   bias=None, act=False, post_scale=None, x_amax=None, y_amax=None):
    """conv_transpose3x3s2 -> blur(pad 1,1) -> noise -> bias + leaky ReLU in one pass: (B,Cin,H,W) -> (B,Cout,2H,2W),
    the four output-parity phases as virtual channels of the F(4x4,3x3) kernel (its error class: image generation).
    Split weights, x_amax and y_amax (max |y|, post_scale included) as in conv3x3_wino4."""
    x = _dev(x, 'fmap')
    packed, uf = uf, _dev(uf, 'packed weight')      # (the scale of a split packing rides on the caller's tensor)
    b, i, h, w = x.shape
    if uf.numel() == lib().rw_packed_conv_transpose_blur_wino4_elems(out_ch, i):
        split = False
    elif uf.numel() == lib().rw_packed_conv_transpose_blur_wino4h_elems(out_ch, i):
        split = True
    else:
        raise ValueError('packed weight does not come from pack_conv_transpose_blur_weight_wino4(%d x %d)'
                         % (out_ch, i))
    y = torch.empty(b, out_ch, 2 * h, 2 * w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    post_scale = _opt(post_scale, 'post scale')
    if post_scale is not None and tuple(post_scale.shape) != (b, out_ch):
        raise ValueError('post_scale must be batch x out_ch')
    if split:
        x_amax, y_amax = _amax_in(x, x_amax), _amax_out(y_amax, y.numel())
        check(lib().rw_conv_transpose3x3s2_blur_wino4h_f32(_p(x), _p(uf), _p(y), b, i, out_ch, h, w, float(w_scale),
                                                           ctypes.byref(ep), _p(post_scale), _u_inv(packed), _p(x_amax),
                                                           _p(y_amax), _stream()))
        return y
    check(lib().rw_conv_transpose3x3s2_blur_wino4_f32(_p(x), _p(uf), _p(y), b, i, out_ch, h, w, float(w_scale),
                                                      ctypes.byref(ep), _p(post_scale), _stream()))
    return y


def dconv_supported(out_ch, in_ch, height, width):
    """Shapes conv3x3_direct16 takes (rw_dconv3x3_supported)."""
    return bool(lib().rw_dconv3x3_supported(int(out_ch), int(in_ch), int(height), int(width)))


def dconv_to_rgb_supported(out_ch, in_ch, height, width):
    return bool(lib().rw_dconv3x3_to_rgb_supported(int(out_ch), int(in_ch), int(height), int(width)))


def dconv_transpose_blur_supported(out_ch, in_ch, height, width):
    return bool(lib().rw_dconv_transpose_blur_supported(int(out_ch), int(in_ch), int(height), int(width)))


def pack_conv_weight_direct16(weight):
    """The weights as f16 pairs in the operand order of the direct kernels on the 16-bit matrix pipe
    (rw_pack_dconv_weight_f32)."""
    weight = _dev(weight, 'weight')
    o, i = weight.shape[-4], weight.shape[-3]
    n = lib().rw_packed_dconv_weight_elems(o, i)
    if n <= 0:
        raise ValueError('no direct-16 packing for a %d x %d weight' % (o, i))
    wp = torch.empty(n, device=weight.device, dtype=torch.float32)
    u_scale, wp.rw_u_inv = _split_scale(lib().rw_dconv_weight_absmax_f32, weight, o, i)
    check(lib().rw_pack_dconv_weight_f32(_p(weight), _p(wp), o, i, u_scale, _stream()))
    return wp


def _direct16_check(wp, n, what):
    if wp.numel() != n:
        raise ValueError('packed weight does not come from %s' % what)


def conv3x3_direct16(x, wp, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None, bias=None, act=False,
                     x_amax=None, y_amax=None):
    """Stride-1 3x3 convolution as a direct sum on the 16-bit matrix pipe (exact f16 operand split, fp32 accumulation:
    rw_dconv3x3_f32); arguments, epilogue, x_amax and y_amax as conv3x3_wino4 with split weights."""
    x = _dev(x, 'fmap')
    packed, wp = wp, _dev(wp, 'packed weight')
    b, i, h, w = x.shape
    _direct16_check(wp, lib().rw_packed_dconv_weight_elems(out_ch, i), 'pack_conv_weight_direct16(%d x %d)' % (out_ch, i))
    y = torch.empty(b, out_ch, h, w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    x_amax, y_amax = _amax_in(x, x_amax), _amax_out(y_amax, y.numel())
    check(lib().rw_dconv3x3_f32(_p(x), _p(wp), _p(y), b, i, out_ch, h, w, float(w_scale), ctypes.byref(ep),
                                _u_inv(packed), _p(x_amax), _p(y_amax), _stream()))
    return y


def conv3x3_direct16_rgb_partial(x, wp, out_ch, w_scale, rgb_weight, rgb_style, rgb_scale, style=None, demod=None, noise=None,
                                 noise_w=None, bias=None, act=False, x_amax=None, y_amax=None):
    """conv3x3_direct16 that also leaves the channel sums of the ToRGB which reads its result (rw_dconv3x3_rgb_partial_f32):
    returns (feature map, partial images (out_ch / 32, B, 3, H, W)); rgb_combine() turns the partials into the image."""
    x = _dev(x, 'fmap')
    packed, wp = wp, _dev(wp, 'packed weight')
    rgb_weight = _dev(rgb_weight, 'rgb weight').contiguous()
    rgb_style = _dev(rgb_style, 'rgb style').contiguous()
    b, i, h, w = x.shape
    if tuple(rgb_weight.shape) != (3, out_ch) or tuple(rgb_style.shape) != (b, out_ch):
        raise ValueError('rgb weight / style shapes')
    _direct16_check(wp, lib().rw_packed_dconv_weight_elems(out_ch, i), 'pack_conv_weight_direct16(%d x %d)' % (out_ch, i))
    n_part = lib().rw_dconv3x3_rgb_partials(out_ch)
    if n_part <= 0:
        raise ValueError('no ToRGB partial sums for %d out-channels' % out_ch)
    y = torch.empty(b, out_ch, h, w, device=x.device, dtype=x.dtype)
    part = torch.empty(n_part, b, 3, h, w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    from ._lib import RgbEpilogue
    re = RgbEpilogue(_p(rgb_weight).value, _p(rgb_style).value, None, None, _p(part).value, float(rgb_scale))
    x_amax, y_amax = _amax_in(x, x_amax), _amax_out(y_amax, y.numel())
    check(lib().rw_dconv3x3_rgb_partial_f32(_p(x), _p(wp), _p(y), b, i, out_ch, h, w, float(w_scale), ctypes.byref(ep),
                                            ctypes.byref(re), _u_inv(packed), _p(x_amax), _p(y_amax), _stream()))
    return y, part


def rgb_combine(partials, bias, skip):
    """ToRGB from the partial sums a convolution left (conv3x3_direct16_rgb_partial): sum over partials + bias + skip."""
    partials = _dev(partials, 'rgb partial sums')
    bias = _opt(bias, 'rgb bias')
    skip = _opt(skip, 'rgb skip')
    n, b, c, h, w = partials.shape
    if c != 3 or (skip is not None and tuple(skip.shape) != (b, 3, h, w)):
        raise ValueError('rgb partial / skip shapes')
    y = torch.empty(b, 3, h, w, device=partials.device, dtype=partials.dtype)
    check(lib().rw_rgb_combine_f32(_p(partials), n, _p(bias), _p(skip), _p(y), b, h * w, _stream()))
    return y


def conv3x3_direct16_to_rgb(x, wp, out_ch, w_scale, rgb_weight, rgb_style, rgb_bias, rgb_skip, rgb_scale, style=None,
                            demod=None, noise=None, noise_w=None, bias=None, act=False, x_amax=None):
    """conv3x3_direct16 with ToRGB in the epilogue (out_ch == 32): returns (None, rgb image)."""
    x = _dev(x, 'fmap')
    packed, wp = wp, _dev(wp, 'packed weight')
    rgb_weight = _dev(rgb_weight, 'rgb weight').contiguous()
    rgb_style = _dev(rgb_style, 'rgb style').contiguous()
    rgb_bias = _opt(rgb_bias, 'rgb bias')
    rgb_skip = _opt(rgb_skip, 'rgb skip')
    b, i, h, w = x.shape
    if tuple(rgb_weight.shape) != (3, out_ch) or tuple(rgb_style.shape) != (b, out_ch):
        raise ValueError('rgb weight / style shapes')
    if rgb_skip is not None and tuple(rgb_skip.shape) != (b, 3, h, w):
        raise ValueError('rgb skip shape')
    _direct16_check(wp, lib().rw_packed_dconv_weight_elems(out_ch, i), 'pack_conv_weight_direct16(%d x %d)' % (out_ch, i))
    rgb = torch.empty(b, 3, h, w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    from ._lib import RgbEpilogue
    re = RgbEpilogue(_p(rgb_weight).value, _p(rgb_style).value, _p(rgb_bias).value, _p(rgb_skip).value,
                     _p(rgb).value, float(rgb_scale))
    x_amax = _amax_in(x, x_amax)
    check(lib().rw_dconv3x3_to_rgb_f32(_p(x), _p(wp), b, i, out_ch, h, w, float(w_scale), ctypes.byref(ep),
                                       ctypes.byref(re), _u_inv(packed), _p(x_amax), _stream()))
    return None, rgb


def pack_conv_transpose_blur_weight_direct16(weight, k4):
    """The four output-parity phases of conv_transpose(stride 2) followed by the 4x4 FIR k4, as f16 pairs in the operand
    order of the direct kernel (rw_pack_dconv_transpose_blur_weight_f32)."""
    weight = _dev(weight, 'weight')
    k4 = _dev(k4, 'blur kernel').contiguous()
    if tuple(k4.shape) != (4, 4):
        raise ValueError('the blur kernel must be 4 x 4')
    o, i = weight.shape[-4], weight.shape[-3]
    n = lib().rw_packed_dconv_transpose_blur_weight_elems(o, i)
    if n <= 0:
        raise ValueError('no direct-16 phase packing for a %d x %d transposed-conv weight' % (o, i))
    wp = torch.empty(n, device=weight.device, dtype=torch.float32)
    u_scale, wp.rw_u_inv = _split_scale(lib().rw_dconv_transpose_blur_weight_absmax_f32, weight, k4, o, i)
    check(lib().rw_pack_dconv_transpose_blur_weight_f32(_p(weight), _p(k4), _p(wp), o, i, u_scale, _stream()))
    return wp


def conv_transpose3x3s2_blur_direct16(x, wp, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None,
                                      bias=None, act=False, post_scale=None, x_amax=None, y_amax=None):
    """conv_transpose3x3s2 -> blur(pad 1,1) -> noise -> bias + leaky ReLU in one pass as a direct sum on the 16-bit matrix
    pipe: (B,Cin,H,W) -> (B,Cout,2H,2W); arguments as conv_transpose3x3s2_blur_wino4 with split weights."""
    x = _dev(x, 'fmap')
    packed, wp = wp, _dev(wp, 'packed weight')
    b, i, h, w = x.shape
    _direct16_check(wp, lib().rw_packed_dconv_transpose_blur_weight_elems(out_ch, i),
                    'pack_conv_transpose_blur_weight_direct16(%d x %d)' % (out_ch, i))
    y = torch.empty(b, out_ch, 2 * h, 2 * w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    post_scale = _opt(post_scale, 'post scale')
    if post_scale is not None and tuple(post_scale.shape) != (b, out_ch):
        raise ValueError('post_scale must be batch x out_ch')
    x_amax, y_amax = _amax_in(x, x_amax), _amax_out(y_amax, y.numel())
    check(lib().rw_dconv_transpose3x3s2_blur_f32(_p(x), _p(wp), _p(y), b, i, out_ch, h, w, float(w_scale),
                                                 ctypes.byref(ep), _p(post_scale), _u_inv(packed), _p(x_amax), _p(y_amax),
                                                 _stream()))
    return y


def tconv_blur_supported(out_ch, in_ch, height, width):
    """Shapes conv_transpose3x3s2_blur_fused takes (rw_tconv_blur_supported)."""
    return bool(lib().rw_tconv_blur_supported(int(out_ch), int(in_ch), int(height), int(width)))


def conv_transpose3x3s2_blur_fused(x, wp, k4, out_ch, w_scale, style=None, demod=None, noise=None, noise_w=None,
                                   bias=None, act=False, post_scale=None, x_amax=None, y_amax=None):
    """conv_transpose3x3s2 -> blur(pad 1,1) -> noise -> bias + leaky ReLU in one pass at the transposed convolution's OWN
    multiply count (rw_tconv_blur_f32: a direct sum on the 16-bit matrix pipe with the exact f16 operand split, the
    (2H+1)^2 map kept in LDS, the FIR read from there): (B,Cin,H,W) -> (B,Cout,2H,2W).  wp = pack_conv_weight_direct16 of
    the layer's weight (the plain packing: nothing is composed with the blur), k4 the 4x4 FIR; the other arguments as
    conv_transpose3x3s2_blur_direct16."""
    x = _dev(x, 'fmap')
    packed, wp = wp, _dev(wp, 'packed weight')
    k4 = _dev(k4, 'blur kernel').contiguous()
    if tuple(k4.shape) != (4, 4):
        raise ValueError('the blur kernel must be 4 x 4')
    b, i, h, w = x.shape
    _direct16_check(wp, lib().rw_packed_dconv_weight_elems(out_ch, i), 'pack_conv_weight_direct16(%d x %d)' % (out_ch, i))
    y = torch.empty(b, out_ch, 2 * h, 2 * w, device=x.device, dtype=x.dtype)
    ep, keep = _epilogue(style, demod, noise, noise_w, bias, act)
    post_scale = _opt(post_scale, 'post scale')
    if post_scale is not None and tuple(post_scale.shape) != (b, out_ch):
        raise ValueError('post_scale must be batch x out_ch')
    x_amax, y_amax = _amax_in(x, x_amax), _amax_out(y_amax, y.numel())
    check(lib().rw_tconv_blur_f32(_p(x), _p(wp), _p(k4), _p(y), b, i, out_ch, h, w, float(w_scale), ctypes.byref(ep),
                                  _p(post_scale), _u_inv(packed), _p(x_amax), _p(y_amax), _stream()))
    return y


def noise_add(x, noise, noise_w):
    x = _dev(x, 'fmap')
    noise = _dev(noise, 'noise')
    noise_w = _dev(noise_w, 'noise weight')
    b, c, h, w = x.shape
    y = torch.empty_like(x)
    check(lib().rw_noise_add_f32(_p(x), _p(noise), _p(noise_w), _p(y), b, c, h * w, _stream()))
    return y


def blur_noise_act(x, k4, noise, noise_w, bias, post_scale=None, y_amax=None):
    """Blur(pad 1,1) + noise + bias + leaky ReLU of an upsampling layer in one pass; post_scale (B x C, optional):
    a factor on the result -- the style of the convolution that consumes it; y_amax (hip.new_bound(result elements),
    optional) receives the bound of the result."""
    x = _dev(x, 'fmap')
    k4 = _dev(k4, 'blur kernel')
    noise = _opt(noise, 'noise')
    noise_w = _opt(noise_w, 'noise weight')
    bias = _opt(bias, 'bias')
    b, c, ih, iw = x.shape
    y = torch.empty(b, c, ih - 1, iw - 1, device=x.device, dtype=x.dtype)
    post_scale = _opt(post_scale, 'post scale')
    if post_scale is not None and tuple(post_scale.shape) != (b, c):
        raise ValueError('post_scale must be batch x channels')
    y_amax = _amax_out(y_amax, y.numel())
    check(lib().rw_blur_noise_act_amax_f32(_p(x), _p(k4), _p(noise), _p(noise_w), _p(bias), _p(post_scale), _p(y),
                                           b, c, ih - 1, iw - 1, _p(y_amax), _stream()))
    return y


def to_rgb(x, weight, style, bias, skip, w_scale):
    x = _dev(x, 'fmap')
    weight = _dev(weight, 'rgb weight')
    style = _dev(style, 'style')
    bias = _opt(bias, 'bias')
    skip = _opt(skip, 'skip')
    b, c, h, w = x.shape
    y = torch.empty(b, 3, h, w, device=x.device, dtype=x.dtype)
    check(lib().rw_to_rgb_f32(_p(x), _p(weight), _p(style), _p(bias), _p(skip), _p(y), b, c, h * w,
                              float(w_scale), _stream()))
    return y


# ------------------------------------------------------------------ statistics
_workspaces = {}


def _workspace(nbytes, device):
    key = (device.index, 'ws')
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def second_moment_accumulate(mom2, a, nchw=False):
    """mom2 (C,C) += a^T a.  a: (rows, C) row-major, or (B, C, H, W) when nchw."""
    a = _dev(a, 'sample')
    if not (mom2.is_cuda and mom2.dtype == torch.float32 and mom2.is_contiguous()):
        raise RuntimeError('mom2 must be a contiguous float32 GPU tensor')
    if nchw:
        b, c, h, w = a.shape
        rows, hw, layout = b * h * w, h * w, 1
        if hw % 16:
            a = a.permute(0, 2, 3, 1).reshape(-1, c).contiguous()
            rows, hw, layout = a.shape[0], 0, 0
    else:
        rows, c = a.shape
        hw, layout = 0, 0
    nbytes = lib().rw_second_moment_workspace_bytes(c, rows)
    ws = _workspace(nbytes, a.device)
    check(lib().rw_second_moment_f32(_p(a), _p(mom2), rows, c, hw, layout, _p(ws), _stream()))
    return mom2


def channel_sums(a, nchw=False, square_input=False):
    a = _dev(a, 'sample')
    if nchw:
        b, c, h, w = a.shape
        rows, hw, layout = b * h * w, h * w, 1
    else:
        rows, c = a.shape
        hw, layout = 0, 0
    sums = torch.empty(2, c, device=a.device, dtype=a.dtype)
    check(lib().rw_channel_sums_f32(_p(a), _p(sums), rows, c, hw, layout, int(bool(square_input)),
                                    _stream()))
    return sums


# ------------------------------------------------------------------ solve
def project_weight(w, context, base=None, out=None):
    """out = base + P(w), P = projected_conv (rewrite/ganrewrite.py:806-813).  w (..., O, I, kh, kw)."""
    wc = _dev(w, 'weight')
    context = _dev(context, 'context')
    base = _opt(base, 'base')
    o, i = wc.shape[-4], wc.shape[-3]
    taps = wc.shape[-1] * wc.shape[-2]
    if out is None:
        out = torch.empty_like(wc)
    check(lib().rw_project_weight_f32(_p(wc), _p(context), _p(base), _p(out), o, i, taps,
                                      context.shape[0], 1.0, _stream()))
    return out


def solve_ksplit(out_ch, in_ch, h, w):
    return lib().rw_solve_ksplit(out_ch, in_ch, h, w)


def solve_supported(out_ch, in_ch, h, w, upsample, plain, constrained):
    """True when rw_solve_step_f32 takes this target (h, w: the key crop); never raises."""
    return lib().rw_solve_supported(out_ch, in_ch, h, w, int(bool(upsample)), int(bool(plain)),
                                    int(bool(constrained))) == 1


def solve_scratch_elems(out_ch, in_ch, h, w, upsample):
    """Element counts of the solver's scratch buffers and the split-K factor they are sized for -- both from the
    library, so that rw_solve_problem.ksplit and the buffers cannot disagree."""
    sizes = (ctypes.c_longlong * 6)()
    check(lib().rw_solve_scratch_elems(out_ch, in_ch, h, w, int(bool(upsample)), sizes))
    return dict(zip(('conv', 'wsq', 'gd', 'c2', 'grad', 'ksplit'), (int(v) for v in sizes)))


def solve_run_supported(out_ch, in_ch, h, w, rank, upsample, linear):
    """True when the one-launch solver (rw_solve_run_f32) takes this target."""
    return lib().rw_solve_run_supported(out_ch, in_ch, h, w, int(rank), int(bool(upsample)), int(bool(linear))) == 1


def solve_run_scratch_elems(out_ch, in_ch, h, w, niter):
    """Floats of the scratch rw_solve_run_f32 needs (per-channel loss parts + the streamed crop's copy)."""
    n = lib().rw_solve_run_scratch_elems(int(out_ch), int(in_ch), int(h), int(w), int(niter))
    if n < 0:
        raise ValueError('rw_solve_run_scratch_elems(%d, %d, %d, %d, %d)' % (out_ch, in_ch, h, w, niter))
    return n


def solve_run(problem, it_begin, it_end, niter, piter, low_rank_insert, lpart):
    check(lib().rw_solve_run_f32(ctypes.byref(problem), int(it_begin), int(it_end), int(niter), int(piter),
                                 int(bool(low_rank_insert)), _p(lpart), _stream()))


def solve_step(problem, project):
    check(lib().rw_solve_step_f32(ctypes.byref(problem), int(bool(project)), _stream()))


# ------------------------------------------------------------------ gradients of the styled convolution (autograd path)
def conv_wgrad(g, x, upsample, scale=1.0, gscale=None, xscale=None):
    """dW (out_ch, in_ch, 3, 3) of y = conv(x, W) [stride 1, pad 1] or conv_transpose(x, W) [stride 2] given
    g = dL/dy: scale * sum_{b,p} (g * gscale[b,o]) * (xcol * xscale[b,i]) -- rw_conv_wgrad_f32."""
    g = _dev(g, 'output gradient')
    x = _dev(x, 'fmap')
    b, i, h, w = x.shape
    o = g.shape[1]
    want = (b, o, 2 * h + 1, 2 * w + 1) if upsample else (b, o, h, w)
    if tuple(g.shape) != want:
        raise ValueError('output gradient %s does not belong to an input map %s' % (tuple(g.shape), tuple(x.shape)))
    gscale, xscale = _opt(gscale, 'gscale'), _opt(xscale, 'xscale')
    ks = lib().rw_conv_wgrad_ksplit(b, i, o, h, w, int(bool(upsample)))
    scratch = torch.empty(ks * o * i * 9, device=x.device, dtype=torch.float32)
    dw = torch.empty(o, i, 3, 3, device=x.device, dtype=torch.float32)
    check(lib().rw_conv_wgrad_f32(_p(g), _p(x), _p(gscale), _p(xscale), _p(scratch), _p(dw), b, i, o, h, w,
                                  int(bool(upsample)), float(scale), _stream()))
    return dw


def rowdot(a, b):
    """(rows,) sums of a * b over everything but the leading `a.dim() - 2` ... dimensions: a, b (..., H, W) ->
    (...,) = sum over the last two dimensions (rw_rowdot_f32)."""
    a = _dev(a, 'a')
    b = _dev(b, 'b')
    if a.shape != b.shape or a.dim() < 3:
        raise ValueError('rowdot takes two maps of the same (..., H, W) shape')
    lead = tuple(a.shape[:-2])
    n = a.shape[-1] * a.shape[-2]
    rows = a.numel() // n
    out = torch.empty(rows, device=a.device, dtype=torch.float32)
    check(lib().rw_rowdot_f32(_p(a), _p(b), _p(out), rows, n, _stream()))
    return out.view(lead)


__all__ = [n for n in dir() if not n.startswith('_')] + ['SolveProblem', 'ConvEpilogue']
