"""Progressive GAN generator (BASELINE.json config 1: "reference plumbing" on PyTorch CPU).

Same module tree and state-dict keys as utils/proggan.py:65-193 (``layerN.{norm,[up],conv,
wscale,relu}``, ``output_RxR.{norm,conv,wscale,clamp}``) so that ``ProgressiveGanRewriter`` and
``nethook.subsequence`` address it identically.  The edited layer is a plain ``nn.Conv2d``; no
HIP kernels are involved.
"""
from collections import OrderedDict

import numpy
import torch
from torch import nn

_DEFAULT_SIZES = {
    8: [512, 512, 512], 16: [512, 512, 512, 512], 32: [512, 512, 512, 512, 256],
    64: [512, 512, 512, 512, 256, 128], 128: [512, 512, 512, 512, 256, 128, 64],
    256: [512, 512, 512, 512, 256, 128, 64, 32],
    1024: [512, 512, 512, 512, 512, 256, 128, 64, 32, 16]}


class PixelNormLayer(nn.Module):
    def forward(self, x):
        return x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


class DoubleResolutionLayer(nn.Module):
    def forward(self, x):
        return nn.functional.interpolate(x, scale_factor=2, mode='nearest')


class WScaleLayer(nn.Module):
    def __init__(self, size, fan_in, gain=numpy.sqrt(2)):
        super().__init__()
        self.scale = gain / numpy.sqrt(fan_in)
        self.b = nn.Parameter(torch.randn(size))
        self.size = size

    def forward(self, x):
        return x * self.scale + self.b.view(1, -1, 1, 1)


def _block(cin, cout, kernel_size, padding, upscale):
    steps = [('norm', PixelNormLayer())]
    if upscale:
        steps.append(('up', DoubleResolutionLayer()))
    steps += [('conv', nn.Conv2d(cin, cout, kernel_size, 1, padding, bias=False)),
              ('wscale', WScaleLayer(cout, cin, gain=numpy.sqrt(2) / kernel_size)),
              ('relu', nn.LeakyReLU(inplace=True, negative_slope=0.2))]
    return steps


class NormConvBlock(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size, padding):
        super().__init__(OrderedDict(_block(in_channels, out_channels, kernel_size, padding, False)))


class NormUpscaleConvBlock(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size, padding):
        super().__init__(OrderedDict(_block(in_channels, out_channels, kernel_size, padding, True)))


class OutputConvBlock(nn.Sequential):
    def __init__(self, in_channels, tanh=False):
        super().__init__(OrderedDict([
            ('norm', PixelNormLayer()),
            ('conv', nn.Conv2d(in_channels, 3, kernel_size=1, padding=0, bias=False)),
            ('wscale', WScaleLayer(3, in_channels, gain=1)),
            ('clamp', nn.Hardtanh() if tanh else nn.Identity())]))


class ProgressiveGenerator(nn.Sequential):
    def __init__(self, resolution=None, sizes=None, modify_sequence=None, output_tanh=True):
        assert (resolution is None) != (sizes is None)
        if sizes is None:
            sizes = _DEFAULT_SIZES[resolution]
        seq = []

        def add(layer, name=None):
            seq.append((name or 'layer%d' % (len(seq) + 1), layer))
        add(NormConvBlock(sizes[0], sizes[1], kernel_size=4, padding=3))
        add(NormConvBlock(sizes[1], sizes[1], kernel_size=3, padding=1))
        for cin, cout in zip(sizes[1:-1], sizes[2:]):
            add(NormUpscaleConvBlock(cin, cout, kernel_size=3, padding=1))
            add(NormConvBlock(cout, cout, kernel_size=3, padding=1))
        dim = 4 * (2 ** (len(seq) // 2 - 1))
        add(OutputConvBlock(sizes[-1], tanh=output_tanh), name='output_%dx%d' % (dim, dim))
        if modify_sequence is not None:
            seq = modify_sequence(seq)
        super().__init__(OrderedDict(seq))

    def forward(self, x):
        return super().forward(x.view(x.shape[0], x.shape[1], 1, 1))
