"""torch.autograd through the modules of a styled convolution, on the HIP kernels.

The reference's ``insert`` is plain autograd over whatever ``target_model`` a rewriter defines
(rewrite/ganrewrite.py:254-298): ``loss.backward()`` runs through ApplyStyle, DemodulatedConv2dF
(utils/stylegan2/models.py:313-329 -- whose demodulation factor is recomputed from the weight and is part of
the graph, quirk Q3), BlurF, NoiseInjectionF and FusedLeakyReLUF.  The fused HIP solver (rewrite/hipsolve.py)
restates that arithmetic for the three targets the reference's rewriters define; everything else -- a target
spanning several layers, a hooked module, a goal batch larger than one -- takes this path: the same loop, the
same torch.optim.Adam, with every forward AND backward product on the kernels of librewriting_hip.so:

* d fmap of the convolution = the forward kernels on the transposed weights (stride 1: flipped taps,
  ``conv3x3``; the stride-2 transposed convolution: the stride-1 correlation of the gradient map sampled at the
  odd positions -- four times the minimal multiply count, accepted on this fallback path);
* d weight = ``rw_conv_wgrad_f32`` (the split-K MFMA GEMM of the solver's K3 without its Adam epilogue) minus the
  demodulation term  s^2 W sigma^2 demod^2 sum_p g y  (``rw_rowdot_f32`` for the per-(image, channel) sums);
* the blur and the leaky ReLU already carry their adjoints (op/upfirdn2d.py, op/fused_act.py).

Gradients the reference computes and never uses (noise strength, activation bias, modulation weights: quirk
Q5) are produced only when autograd asks for them.
"""
import torch
from torch.autograd import Function

from ... import hip


class StyleMul(Function):
    """ApplyStyle: fmap * style[:, :, None, None]  (models.py:616-620)."""

    @staticmethod
    def forward(ctx, fmap, style):
        ctx.save_for_backward(fmap, style)
        return hip.style_mul(fmap, style)

    @staticmethod
    def backward(ctx, g):
        fmap, style = ctx.saved_tensors
        g = g.contiguous()
        gf = hip.style_mul(g, style) if ctx.needs_input_grad[0] else None
        gs = hip.rowdot(g, fmap) if ctx.needs_input_grad[1] else None
        return gf, gs


class NoiseAdd(Function):
    """NoiseInjectionF: fmap + weight * noise  (models.py:539-546); noise is a constant of the graph."""

    @staticmethod
    def forward(ctx, fmap, noise, weight):
        ctx.save_for_backward(noise)
        return hip.noise_add(fmap, noise, weight)

    @staticmethod
    def backward(ctx, g):
        noise, = ctx.saved_tensors
        gw = None
        if ctx.needs_input_grad[2]:
            b = g.shape[0]
            gw = (g.sum(1).reshape(b, -1) * noise.reshape(b, -1)).sum().reshape(1)
        return (g if ctx.needs_input_grad[0] else None), None, gw


def transposed_weight(weight, flip):
    """(1, O, I, 3, 3) -> (1, I, O, 3, 3), taps flipped for the stride-1 backward-to-input."""
    w = weight.detach()[0].transpose(0, 1)
    if flip:
        w = w.flip(2, 3)
    return w.contiguous()[None]


class DemodConv(Function):
    """DemodulatedConv2dF.forward (models.py:313-329): y = conv(x, s W) * rsqrt(sum (s W sigma)^2 + eps), with x the
    already modulated map.  `module` supplies the kernels' dispatch (algorithm, packed weights) and the constants."""

    @staticmethod
    def forward(ctx, fmap, weight, style, module):
        y = module.run(fmap, style, style_on_load=False, weight_changes=bool(weight.requires_grad))
        ctx.module = module
        ctx.save_for_backward(fmap, weight, style, y)
        return y

    @staticmethod
    def backward(ctx, g):
        fmap, weight, style, y = ctx.saved_tensors
        m = ctx.module
        g = g.contiguous()
        s = m.scale
        demod = hip.demod(hip.weight_sqsum(weight, s), style) if m.demodulate else None
        gx = gw = gs = None
        if ctx.needs_input_grad[0]:
            wt = hip.pack_conv_weight(transposed_weight(weight, flip=not m.upsample), 0)
            # the MFMA kernels take in_ch % 16 == 0 and out_ch % 32 == 0 (every layer of the generators); odd test
            # shapes go to the one-thread-per-output kernel
            impl = 0 if (m.out_channel % 16 == 0 and m.in_channel % 32 == 0) else 1
            c = hip.conv3x3(g, wt, m.in_channel, s, style=demod, impl=impl)   # demod rides in as the on-load factor of g
            gx = c[:, :, 1::2, 1::2].contiguous() if m.upsample else c
        need_r = m.demodulate and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        if need_r:
            rd = hip.rowdot(g, y) * demod * demod                         # (B, O): sum_p g conv * demod^3, conv = y / demod
        if ctx.needs_input_grad[1]:
            dw = hip.conv_wgrad(g, fmap, m.upsample, scale=s, gscale=demod)
            if m.demodulate:
                dw = dw - (s * s) * weight.detach()[0] * torch.mm(rd.t(), style.detach() ** 2)[:, :, None, None]
            gw = dw[None]
        if ctx.needs_input_grad[2] and m.demodulate:
            gs = -style.detach() * torch.mm(rd, hip.weight_sqsum(weight, s))
        return gx, gw, gs, None
