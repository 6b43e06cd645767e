"""upfirdn2d (upsample, FIR filter, downsample) with autograd.

Same call signature as utils/stylegan2/op/upfirdn2d.py:144-149.  The adjoint is the same
operator with up and down exchanged, the flipped kernel and the ``g_pad`` algebra of
:100-115 (restated in ``_adjoint_pads``); because the operator is linear the double backward
is the forward again (:52-84).  Kernels: ``rw_upfirdn2d_f32``.
"""
import torch
from torch.autograd import Function

from .... import hip


def _out_size(n, up, down, pad0, pad1, k):
    return (n * up + pad0 + pad1 - k) // down + 1


def _adjoint_pads(in_size, out_size, k, up, down, pad0):
    g0 = k - pad0 - 1
    g1 = in_size * up - out_size * down + pad0 - up + 1
    return g0, g1


def _run(x4, kernel, up, down, pads):
    b, c, h, w = x4.shape
    out = hip.upfirdn2d_major(x4.reshape(-1, h, w, 1), kernel, up[0], up[1], down[0], down[1], *pads)
    return out.view(b, c, out.shape[1], out.shape[2])


class _UpFirDn2dAdjoint(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, cfg):
        up, down, pads, in_shape = cfg
        ctx.save_for_backward(kernel)
        ctx.cfg = cfg
        kh, kw = kernel.shape
        oh, ow = grad_output.shape[2:]
        gx0, gx1 = _adjoint_pads(in_shape[3], ow, kw, up[0], down[0], pads[0])
        gy0, gy1 = _adjoint_pads(in_shape[2], oh, kh, up[1], down[1], pads[2])
        g = _run(grad_output, torch.flip(kernel, [0, 1]), down, up, (gx0, gx1, gy0, gy1))
        return g.reshape(in_shape)

    @staticmethod
    def backward(ctx, gg_input):
        kernel, = ctx.saved_tensors
        up, down, pads, _ = ctx.cfg
        return _run(gg_input, kernel, up, down, pads), None, None


class _UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pads):
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pads, tuple(input.shape))
        return _run(input, kernel, up, down, pads)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, = ctx.saved_tensors
        return _UpFirDn2dAdjoint.apply(grad_output, kernel, ctx.cfg), None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return _UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
