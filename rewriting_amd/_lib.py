"""ctypes binding of librewriting_hip.so (C ABI declared in include/rewriting_hip.h).

The library is prebuilt in-tree by ``__graft_entry__.build()`` /
``rewriting_amd/csrc/build.sh`` (hipcc, gfx950); nothing is JIT-compiled at import, unlike
the reference's ``torch.utils.cpp_extension.load`` (utils/stylegan2/op/fused_act.py:10-16).
There is NO fallback: if the shared object is missing or a tensor is not on a HIP device the
wrappers raise.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RW_HIP_LIB') or os.path.join(_HERE, 'librewriting_hip.so')    # RW_HIP_LIB: tuning builds

ABI_VERSION = 9        # 9: rw_dconv3x3_rgb_partial_f32 / rw_rgb_combine_f32 (ToRGB sums left by the producing convolution); 8: rw_tconv_blur_* (fused transposed conv + blur); 7: bounds as RW_BOUND_LANES-float vectors written by plain stores, weight scales by value (rw_split_weight_scale, rw_*_absmax_f32), no rw_publish_scalar_f32; 6: rw_publish_scalar_f32; 5: rw_dconv* (direct sums on the 16-bit pipe); 4: rw_*_wino4h_* (operands split into f16 pairs), rw_absmax_f32; 3: rw_solve_run_*, the 8x8 / 4x4 shapes (style == NULL), packed F(4x4,3x3) point order w4_nat; 2: rw_solve_supported 1/0, sizes[6]


class ConvEpilogue(Structure):
    _fields_ = [('style', c_void_p), ('demod', c_void_p), ('noise', c_void_p),
                ('noise_w', c_void_p), ('bias', c_void_p), ('act', c_int)]


class RgbEpilogue(Structure):
    _fields_ = [('weight', c_void_p), ('style', c_void_p), ('bias', c_void_p), ('skip', c_void_p),
                ('out', c_void_p), ('scale', c_float)]


class SolveProblem(Structure):
    _fields_ = [
        ('out_ch', c_int), ('in_ch', c_int), ('h', c_int), ('w', c_int), ('rank', c_int),
        ('key', c_void_p), ('style', c_void_p), ('val', c_void_p), ('bias', c_void_p),
        ('noise', c_void_p), ('noise_w', c_void_p), ('context', c_void_p), ('ortho', c_void_p),
        ('weight', c_void_p), ('exp_avg', c_void_p), ('exp_avg_sq', c_void_p),
        ('step_size', c_void_p), ('bc2_sqrt', c_void_p), ('step_counter', c_void_p),
        ('losses', c_void_p),
        ('conv', c_void_p), ('wsq', c_void_p), ('gd', c_void_p), ('c2', c_void_p),
        ('grad', c_void_p),
        ('ksplit', c_int), ('beta1', c_float), ('beta2', c_float), ('eps', c_float),
        ('w_scale', c_float), ('low_rank_gradient', c_int),
        ('one_minus_beta1', c_float), ('one_minus_beta2', c_float),
        ('upsample', c_int), ('blur_k', c_void_p), ('linear_insert', c_int), ('lambda_', c_void_p),
    ]


# name -> (restype, argtypes); must list EVERY symbol include/rewriting_hip.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it).
SIGNATURES = {
    'rw_abi_version': (c_int, []),
    'rw_error_string': (c_char_p, [c_int]),
    'rw_fused_bias_act_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                      c_int64, c_int, c_int, c_float, c_float, c_void_p]),
    'rw_bias_grad_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    'rw_upfirdn2d_f32': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 14 + [c_void_p]),
    'rw_pixel_norm_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    'rw_equal_linear_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_int64, c_float, c_float, c_int, c_float, c_float, c_void_p]),
    'rw_adjust_latent_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                     c_void_p]),
    'rw_style_mul_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    'rw_weight_sqsum_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    'rw_demod_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    'rw_packed_conv_weight_elems': (ctypes.c_longlong, [c_int, c_int, c_int]),
    'rw_pack_conv_weight_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rw_conv3x3_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_float, POINTER(ConvEpilogue), c_int, c_void_p]),
    'rw_packed_conv_weight_bf16x3_bytes': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_conv_weight_bf16x3': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'rw_conv3x3_bf16x6_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_float, POINTER(ConvEpilogue), c_void_p]),
    'rw_conv3x3_to_rgb_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_float, POINTER(ConvEpilogue), POINTER(RgbEpilogue), c_void_p]),
    'rw_conv_transpose3x3s2_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_int, c_float, POINTER(ConvEpilogue), c_int, c_void_p]),
    'rw_noise_add_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                 c_void_p]),
    'rw_blur_noise_act_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_int, c_int, c_int, c_void_p]),
    'rw_to_rgb_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                              c_int, c_int64, c_float, c_void_p]),
    'rw_second_moment_workspace_bytes': (c_int64, [c_int, c_int64]),
    'rw_second_moment_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p,
                                     c_void_p]),
    'rw_channel_sums_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int,
                                    c_void_p]),
    'rw_conv3x3_wino_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_packed_conv_weight_wino_elems': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_conv_weight_wino_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'rw_conv3x3_wino_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                    POINTER(ConvEpilogue), c_void_p]),
    'rw_conv3x3_wino_to_rgb_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                           POINTER(ConvEpilogue), POINTER(RgbEpilogue), c_void_p]),
    'rw_conv3x3_wino4_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_packed_conv_weight_wino4_elems': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_conv_weight_wino4_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'rw_conv3x3_wino4_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                     POINTER(ConvEpilogue), c_void_p]),
    'rw_conv_transpose3x3s2_wino_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_packed_conv_transpose_wino_elems': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_conv_transpose_wino_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'rw_conv_transpose3x3s2_wino_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                c_float, c_void_p, c_void_p, c_void_p]),
    'rw_conv_transpose3x3s2_winoh_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_packed_conv_transpose_winoh_elems': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_conv_transpose_winoh_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    'rw_conv_transpose3x3s2_winoh_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                 c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    'rw_conv3x3_wino4_to_rgb_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_conv3x3_wino4_to_rgb_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                            POINTER(ConvEpilogue), POINTER(RgbEpilogue), c_void_p]),
    'rw_absmax_f32': (c_int, [c_void_p, ctypes.c_longlong, c_void_p, c_void_p]),
    'rw_bound_floats': (ctypes.c_longlong, [ctypes.c_longlong]),
    'rw_split_weight_scale': (c_float, [c_float]),
    'rw_conv_weight_wino4h_absmax_f32': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'rw_conv_transpose_blur_weight_wino4h_absmax_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'rw_conv_transpose_weight_winoh_absmax_f32': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'rw_dconv_weight_absmax_f32': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'rw_dconv_transpose_blur_weight_absmax_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'rw_packed_conv_weight_wino4h_elems': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_conv_weight_wino4h_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    'rw_conv3x3_wino4h_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                      POINTER(ConvEpilogue), c_float, c_void_p, c_void_p, c_void_p]),
    'rw_conv3x3_wino4h_to_rgb_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                             POINTER(ConvEpilogue), POINTER(RgbEpilogue), c_float, c_void_p, c_void_p]),
    'rw_packed_conv_transpose_blur_wino4h_elems': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_conv_transpose_blur_weight_wino4h_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                                              c_void_p]),
    'rw_conv_transpose3x3s2_blur_wino4h_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                       c_int, c_float, POINTER(ConvEpilogue), c_void_p, c_float,
                                                       c_void_p, c_void_p, c_void_p]),
    'rw_dconv3x3_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_packed_dconv_weight_elems': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_dconv_weight_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    'rw_dconv3x3_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                POINTER(ConvEpilogue), c_float, c_void_p, c_void_p, c_void_p]),
    'rw_dconv3x3_rgb_partials': (c_int, [c_int]),
    'rw_dconv3x3_rgb_partial_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                            POINTER(ConvEpilogue), POINTER(RgbEpilogue), c_float, c_void_p, c_void_p,
                                            c_void_p]),
    'rw_rgb_combine_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    'rw_dconv3x3_to_rgb_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_dconv3x3_to_rgb_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                       POINTER(ConvEpilogue), POINTER(RgbEpilogue), c_float, c_void_p, c_void_p]),
    'rw_dconv_transpose_blur_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_packed_dconv_transpose_blur_weight_elems': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_dconv_transpose_blur_weight_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    'rw_dconv_transpose3x3s2_blur_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                 c_float, POINTER(ConvEpilogue), c_void_p, c_float, c_void_p, c_void_p,
                                                 c_void_p]),
    'rw_tconv_blur_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_tconv_blur_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                  POINTER(ConvEpilogue), c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    'rw_conv_transpose_blur_wino4_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_packed_conv_transpose_blur_wino4_elems': (ctypes.c_longlong, [c_int, c_int]),
    'rw_pack_conv_transpose_blur_weight_wino4_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'rw_conv_transpose3x3s2_blur_wino4_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                      c_int, c_float, POINTER(ConvEpilogue), c_void_p, c_void_p]),
    'rw_blur_noise_act_scaled_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int, c_int, c_int, c_int, c_void_p]),
    'rw_blur_noise_act_amax_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'rw_solve_ksplit': (c_int, [c_int, c_int, c_int, c_int]),
    'rw_solve_supported': (c_int, [c_int] * 7),
    'rw_solve_scratch_elems': (c_int, [c_int] * 5 + [POINTER(ctypes.c_longlong)]),
    'rw_solve_step_f32': (c_int, [POINTER(SolveProblem), c_int, c_void_p]),
    'rw_solve_run_supported': (c_int, [c_int] * 7),
    'rw_solve_run_scratch_elems': (ctypes.c_longlong, [c_int] * 5),
    'rw_solve_run_f32': (c_int, [POINTER(SolveProblem), c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'rw_project_weight_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_int, c_float, c_void_p]),
    'rw_conv_wgrad_ksplit': (c_int, [c_int] * 6),
    'rw_conv_wgrad_f32': (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_float, c_void_p]),
    'rw_rowdot_f32': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, ctypes.c_longlong, c_void_p]),
}

_lib = None


def load():
    """Returns the loaded CDLL; raises RuntimeError (never falls back) if it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            'rewriting_amd: %s is missing -- build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` or rewriting_amd/csrc/build.sh (hipcc --offload-arch=gfx950). There is no '
            'CPU or PyTorch fallback for the HIP kernels.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    # version first: a stale .so (it is git-ignored; RW_HIP_LIB may point at an old tuning build) must say
    # "rebuild" instead of failing on the first symbol it lacks
    stale = 'rewriting_amd: %s is stale (%s) -- rebuild it with rewriting_amd/csrc/build.sh or ' \
            '`python -c "import __graft_entry__ as g; g.build()"`'
    try:
        lib.rw_abi_version.restype = c_int
        lib.rw_abi_version.argtypes = []
        have = lib.rw_abi_version()
    except AttributeError:
        raise RuntimeError(stale % (LIB_PATH, 'no rw_abi_version symbol')) from None
    if have != ABI_VERSION:
        raise RuntimeError(stale % (LIB_PATH, 'ABI %d, this package binds ABI %d' % (have, ABI_VERSION)))
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise RuntimeError(stale % (LIB_PATH, 'ABI %d but no symbol %s' % (have, name))) from None
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().rw_error_string(code)
        raise RuntimeError('librewriting_hip: %s (code %d)' % (msg.decode() if msg else '?', code))
