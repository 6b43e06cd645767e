"""TEST INFRASTRUCTURE ONLY -- ctypes access to oracle/native_ops.c (built by oracle/Makefile)."""
import ctypes
import os
import subprocess

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_build', 'liboracle_native.so')


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def _lib():
    if not os.path.isfile(SO):
        build()
    return ctypes.CDLL(SO)


def _f(a):
    return numpy.ascontiguousarray(a, dtype=numpy.float32)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else ctypes.c_void_p(0)


def fused_bias_act(x, b, ref, act, grad, alpha, scale):
    x = _f(x)
    b = _f(b) if b is not None else None
    ref = _f(ref) if ref is not None else None
    y = numpy.empty_like(x)
    step_b = int(numpy.prod(x.shape[2:])) if x.ndim > 2 else 1
    size_b = b.size if b is not None else 1
    fn = _lib().oracle_fused_bias_act
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_int] * 2 + [ctypes.c_float] * 2
    fn(_ptr(x), _ptr(b), _ptr(ref), _ptr(y), x.size, step_b, size_b, act, grad, alpha, scale)
    return y


def upfirdn2d(x, k, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    x, k = _f(x), _f(k)
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h = (in_h * up_y + py0 + py1 - kh + down_y) // down_y
    out_w = (in_w * up_x + px0 + px1 - kw + down_x) // down_x
    y = numpy.empty((major, out_h, out_w, minor), dtype=numpy.float32)
    fn = _lib().oracle_upfirdn2d
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 14
    fn(_ptr(x), _ptr(k), _ptr(y), major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y,
       px0, px1, py0, py1)
    return y
