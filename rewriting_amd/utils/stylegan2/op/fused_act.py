"""fused bias + leaky ReLU with first- and second-order autograd.

Mirrors the public surface of utils/stylegan2/op/fused_act.py (``FusedLeakyReLU`` :73-82,
``fused_leaky_relu`` :85-86) and its autograd structure (forward: kernel case act=3/grad=0,
:51-60; backward: case grad=1 against the saved OUTPUT, then ``grad_bias`` as a reduction,
:19-39; double backward :41-48), with every kernel call going to the HIP library through the
C ABI (``rw_fused_bias_act_f32`` / ``rw_bias_grad_f32``).
"""
import torch
from torch import nn
from torch.autograd import Function

from .... import hip


class _LeakyReLUGrad(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.slope_scale = (negative_slope, scale)
        grad_input = hip.fused_bias_act(grad_output, None, out, 3, 1, negative_slope, scale)
        return grad_input, hip.bias_grad(grad_input)

    @staticmethod
    def backward(ctx, gg_input, gg_bias):
        out, = ctx.saved_tensors
        slope, scale = ctx.slope_scale
        return hip.fused_bias_act(gg_input, gg_bias, out, 3, 1, slope, scale), None, None, None


class _LeakyReLU(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = hip.fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.slope_scale = (negative_slope, scale)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        slope, scale = ctx.slope_scale
        grad_input, grad_bias = _LeakyReLUGrad.apply(grad_output, out, slope, scale)
        return grad_input, grad_bias, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return _LeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
