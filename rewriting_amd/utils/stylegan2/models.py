"""Sequential StyleGANv2 generator on MI355X kernels.

Drop-in for utils/stylegan2/models.py: the same class names, constructor arguments, child-module
names and state-dict keys (level B2), so ``nethook.subsequence``,
``InstrumentedModel`` and ``ganrewrite.SeqStyleGanRewriter`` split and hook it exactly as they
split the reference.  Every non-leaf is an ``nn.Sequential``; data flows as ``DataBag`` dicts.

What differs is underneath: every leaf calls a hand-written gfx950 kernel through the C ABI
(``rewriting_amd.hip``), and a ``StyledConvSeq`` whose children are all present and un-hooked
runs as ONE fused block (style multiply folded into the implicit-GEMM gather; demodulation,
noise, bias and leaky-ReLU in its epilogue) -- invisible at the module boundaries the
rewriter observes, because any hook or split turns the fusion off for that block.
"""
import math
import os
import re
import threading
import warnings
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import grad, op
from ... import hip

# ------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------

_weight_epoch = [0]


def bump_weight_epoch():
    """Called by code that rewrites parameters through raw device pointers (the HIP solver),
    which torch's version counters cannot see; invalidates every derived-weight cache."""
    _weight_epoch[0] += 1


class _DerivedWeights:
    """Cache of tensors derived from a parameter (repacked weights, squared sums), keyed on the
    parameter's storage, torch version counter and the package-wide epoch."""

    def __init__(self):
        self.store = {}

    def get(self, name, param, make):
        key = (param.data_ptr(), param._version, _weight_epoch[0], str(param.device))
        hit = self.store.get(name)
        if hit is None or hit[0] != key:
            hit = (key, make())
            self.store[name] = hit
        return hit[1]


_noise_streams = {}


def reference_noise(batch, hw, device):
    """The reference regenerates ``np.random.RandomState(0).randn(batch, H*W)`` on the host for
    every noise layer of every call (utils/stylegan2/models.py:542-545; quirk Q1).  That is a
    fixed stream prefix, so it is generated once and kept on the device."""
    need = batch * hw
    key = str(device)
    stream = _noise_streams.get(key)
    if stream is None or stream.numel() < need:
        n = max(need, 1 << 16)
        host = np.random.RandomState(0).randn(n).astype('float32')
        stream = torch.from_numpy(host).to(device)
        _noise_streams[key] = stream
    period = _noise_period[0]
    if period == 1 and batch > 1:
        return reference_noise_row0(hw, device).expand(batch, hw)
    if period and batch > period:
        # Several reference-sized batches run as one launch: image j takes the noise row it would have
        # had in its own batch of `period` (row j mod period), see noise_batch_period().
        if batch % period:
            raise ValueError('batch %d is not a multiple of the noise period %d' % (batch, period))
        rows = reference_noise(period, hw, device)
        return rows.repeat(batch // period, 1)
    return stream[:need].view(batch, hw)


def reference_noise_row0(hw, device):
    _noise_period[0], saved = 0, _noise_period[0]
    try:
        return reference_noise(1, hw, device)
    finally:
        _noise_period[0] = saved


_noise_period = [0]


class noise_batch_period:
    """Context manager.  The reference's noise depends on an image's ROW WITHIN ITS BATCH (quirk Q1),
    and its statistics sweeps use batches of 10 (utils/tally.py:631-647).  Inside this context a
    batch of k*period images is treated as k consecutive reference batches, so a sweep can run
    large launches and still give every seed exactly the noise the reference gives it."""

    def __init__(self, period):
        self.period = int(period)

    def __enter__(self):
        self.old = _noise_period[0]
        _noise_period[0] = self.period
        return self

    def __exit__(self, *exc):
        _noise_period[0] = self.old


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


def _unhooked(*modules):
    """nethook hooks a layer by planting a ``forward`` attribute on the instance
    (utils/nethook.py:197-201)."""
    return all('forward' not in m.__dict__ for m in modules)


def fusion_enabled():
    return os.environ.get('RW_FUSE', '1') != '0'


# The RGB branch (up_rgbK, to_rgbK) is HBM-bound and hangs off the feature-map trunk, whose convolutions
# are MFMA-bound: when the WHOLE generator runs un-hooked (SeqStyleGAN2.forward below) the branch is
# issued on a second HIP stream so that it overlaps the next styled convolutions, and is joined
# before the image is returned.  Hooked / sliced models (nethook) never see this: the mode is on only
# inside that forward.
class _RgbBranch(threading.local):      # per thread: two threads may run generators concurrently
    def __init__(self):
        self.stream = None
        self.aux = None
        self.keep = []
        self.final = None        # (last StyledConvSeq, its ToRGBF, latent index of the ToRGB) of the running forward
        self.image_path = False  # inside the un-hooked forward of a whole generator (see conv_algo)
        self.successor = {}      # id(upsampling StyledConvSeq) -> (the StyledConvSeq that reads its result, latent index)
        self.reader = {}         # id(StyledConvSeq) -> the StyledConvSeq that reads its feature map (any kind), if any
        self.torgb = {}          # id(StyledConvSeq) -> (the ToRGBF that reads its feature map, its latent index), if any
        self.pre = {}            # id(StyledConvSeq) -> (style, demod factors), id(ToRGBF) -> style: computed up front
        self.pre_join = None     # the stream they were computed on, until the trunk has waited for it


_rgb_branch = _RgbBranch()
_rgb_side_streams = {}          # one per device, module-level: models are deep-copied by the rewriters


def _side_stream(device):
    """A stream for the work beside the trunk (RGB branch, border strips, prefetch).  RW_SIDE_PRIORITY (default 0): the
    priority torch gives it (positive = below the trunk's stream; clamped to the device's range)."""
    return torch.cuda.Stream(device=device, priority=int(os.environ.get('RW_SIDE_PRIORITY', '0')))


def _rgb_stream():
    return _rgb_branch.stream


def _prefetched(module):
    """(style, demod) of a StyledConvSeq / the style of a ToRGBF computed at the start of the un-hooked forward
    (SeqStyleGAN2._prefetch_modulations), or None.  The first reader makes the trunk wait for the stream they
    were computed on."""
    entry = _rgb_branch.pre.get(id(module))
    if entry is not None and _rgb_branch.pre_join is not None:
        torch.cuda.current_stream().wait_stream(_rgb_branch.pre_join)
        _rgb_branch.pre_join = None
    return entry


_CONV_IMPLS = {'auto': 0, 'mfma': 0, 'direct': 1, 'generic': 2, 'halo': 3, 'nosplitk': 5}


def conv_impl():
    """Kernel choice for the 3x3 convolutions (include/rewriting_hip.h, rw_conv3x3_f32 impl):
    auto = halo-tile MFMA kernel where the map is >= 24 wide else the im2col MFMA kernel;
    'generic' / 'halo' force one of them, 'direct' is the VALU cross-check."""
    return _CONV_IMPLS[os.environ.get('RW_CONV_IMPL', 'auto')]


def conv_precision():
    """'f32' (default: exact fp32 MFMA) or 'bf16x6' (RW_CONV_PRECISION=bf16x6, opt-in): the stride-1 3x3
    convolutions of eligible shapes run on the bf16 matrix pipe with exact three-way operand splits
    (hip.conv3x3_bf16x6, fp32-product accuracy); every other kernel is unchanged."""
    return os.environ.get('RW_CONV_PRECISION', 'f32')


def conv_algo():
    """Algorithm of the stride-1 3x3 convolutions where more than one exists.

    Without RW_CONV_ALGO the answer depends on HOW the modules are being run:
      * inside the un-hooked forward of the whole generator (SeqStyleGAN2.forward sets a thread-local flag; image
        generation): 'winograd4' = F(4x4,3x3) where hip.wino4_supported says so -- 4x fewer matrix FLOPs at ~1e-5
        relative error per layer, 3e-5 measured on the 1024^2 image against the reference (the path's image
        tolerance is 1e-3) -- and F(2x2,3x3) elsewhere;
      * every other way -- a hooked model, a nethook.subsequence slice (key statistics, goal maps, the solve's
        context and its rendering), RW_FUSE=0: 'winograd' = F(2x2,3x3) (hip.conv3x3_wino: 2.25x fewer matrix FLOPs,
        the fp32 error class of the direct sum) -- which since round 5 is what runs BELOW 32^2 only: from 32^2 up these
        models' stride-1 layers are direct sums on the 16-bit pipe (DemodulatedConv2dF.hooked_direct16: exact f16
        operand pairs, 4e-7 from the fp32 direct sum; not a layer whose weight is being optimised) and their upsampling layers
        the fused split kernel (fused_upsample); the statistics goldens hold at 4e-6 - 5e-6 from the reference's.
    Consequence: model(z) and the same weights run through rewriter.sample_image_from_latent differ by 1e-5 .. 1e-4
    on the image.  RW_CONV_ALGO=direct|winograd|winograd4 forces one algorithm everywhere; shapes an algorithm
    does not take always run the next one down ('direct' = the implicit GEMM takes everything)."""
    explicit = os.environ.get('RW_CONV_ALGO')
    if explicit:
        return explicit
    return _IMAGE_CONV_ALGO if _rgb_branch.image_path else _DEFAULT_CONV_ALGO


def up_conv_algo():
    """Algorithm of the stride-2 transposed convolutions: 'winograd' = F(2,2) on the four output-parity phases
    (hip.conv_transpose3x3s2_wino: 25 instead of 36 multiplies per 2x2 block of quads; coefficients 0, +-1, the direct
    sum's error class -- the default everywhere it applies) or 'direct'.  RW_UP_ALGO selects."""
    return os.environ.get('RW_UP_ALGO', 'winograd')


def matrix_mode(kind=None):
    """Which matrix pipe multiplies inside the F(4x4,3x3) and F(2,2) kernels: 'f32' (fp32 MFMAs) or 'split' -- every
    transformed operand as an exact pair of f16 numbers, the piece products on the 16-bit pipe, fp32 accumulation
    (hip.pack_conv_weight_wino4(split=True) ...: per-product error <= 2^-21, the kernels pass their parity tests at the
    fp32 bars).  Inside the un-hooked forward of the whole generator (image generation): 'split' for every kind of
    kernel.  Hooked / sliced models -- key statistics, goal maps, the solve's context -- run their stride-1
    convolutions on F(2x2,3x3) in fp32 below 32^2 and as direct sums on the 16-bit pipe from there up (F(4x4,3x3) never
    runs there: conv_algo; DemodulatedConv2dF.hooked_direct16), and run the F(2,2) transposed
    convolutions (kind 'up') in the split form too, which holds the DIRECT kernels' bars (test_transposed_conv_f22_...:
    3e-6 relative) and is 1.4 - 1.5 x faster on the 64^2 ... 512^2 maps of a layer-10 / layer-14 sweep; RW_MM_HOOKED=f32
    keeps them on fp32.  Where that split form applies, an upsampling layer without a hook inside runs as ONE launch of the
    fused kernel (DemodulatedConv2dF.fused_upsample; +13 - 16 % on those sweeps).  RW_MM=f32|split forces one everywhere."""
    explicit = os.environ.get('RW_MM')
    if explicit:
        return explicit
    if _rgb_branch.image_path:
        return 'split'
    return 'split' if kind == 'up' and os.environ.get('RW_MM_HOOKED', 'split') != 'f32' else 'f32'


def matrix_mode_of_image_path():
    """matrix_mode() as the un-hooked forward of a whole generator sees it (bench.py labels its line with it)."""
    return os.environ.get('RW_MM') or 'split'


def _split_part(kind):
    """Diagnostics: RW_MM_PARTS=w4,up,up1 restricts the split form to the stride-1 F(4x4,3x3) kernels / the F(2,2)
    transposed convolutions / the one-pass upsampling layer."""
    parts = os.environ.get('RW_MM_PARTS')
    return matrix_mode(kind) == 'split' and (parts is None or kind in parts.split(','))


def _direct16(dconv, h, w, kind):
    """Inside the split form: the DIRECT sums on the 16-bit matrix pipe (csrc/rw_dconv.hip) in place of the
    split-operand F(4x4,3x3) kernels, kind 'conv' / 'up' / 'rgb'.  RW_MM_DIRECT16: 'auto' (the default) = 'conv,up', the
    two that measure faster INSIDE the forward of the 1024 generator (same box, batch 64, profiles/r05c_ab.jsonl: 1262
    against 1221 img/s; per launch the stride-1 layers 3.4 / 4.3 ms against 3.7 - 4.7, the one-pass upsampling layer 10.2
    against 10.4; the last layer + ToRGB is slower as a direct sum: 7.0 against 5.8); '1' = all three, '0' = none, or a
    comma-separated list of kinds.  (Round 4 left them opt-in because sequences of un-synced forwards were occasionally
    wrong with them -- the device scalars of that round's bound hand-over; tests/test_gpu_zz_sequences.py holds both
    selections to bit-identical sequences now.)"""
    mode = os.environ.get('RW_MM_DIRECT16', 'auto')
    kinds = ('conv', 'up') if mode == 'auto' else ('conv', 'up', 'rgb') if mode == '1' else mode.split(',')
    if dconv.in_channel < 32 or kind not in kinds:
        return False
    if kind == 'up':
        return hip.dconv_transpose_blur_supported(dconv.out_channel, dconv.in_channel, h, w)
    if kind == 'rgb':
        return hip.dconv_to_rgb_supported(dconv.out_channel, dconv.in_channel, h, w)
    return hip.dconv_supported(dconv.out_channel, dconv.in_channel, h, w)


def _amax_of(fmap):
    """The bound of |fmap| that its producer -- a fused layer of the running un-hooked forward -- left ON the tensor
    (attribute rw_amax: (bound, the tensor's version counter when it was written)), or None: the kernels then measure
    the map themselves (hip.absmax).  It travels with the tensor object, not with an address or a bag key: a slice, a
    copy or a map that was edited in place since (hooks) carries no usable bound.  Trusted only inside the un-hooked
    forward of a whole generator, where producer and consumer are both ours."""
    if fmap is None or not _rgb_branch.image_path:
        return None
    entry = getattr(fmap, 'rw_amax', None)
    if entry is not None and entry[1] == fmap._version:
        return entry[0]
    return None


# Default: F(2x2,3x3) wherever a model is hooked, sliced (nethook.subsequence: the key statistics, the goal maps,
# the solve's context) or run module by module; F(4x4,3x3) inside the un-hooked forward of the whole generator --
# image generation, where the measured deviation from the reference image is the same 2e-5 with either.
_DEFAULT_CONV_ALGO = 'winograd'
_IMAGE_CONV_ALGO = 'winograd4'
_ONE_PASS_UP_MAX_IN = 64       # see DemodulatedConv2dF.one_pass_upsample
_FUSED_UP_MAX_IN = 512         # see DemodulatedConv2dF.fused_upsample


def micro_batch():
    """(images per slice, first resolution run in slices) of SeqStyleGAN2._forward_micro, from
    RW_MICRO_BATCH = "k" or "k:res" (0 = one launch per step for the whole batch)."""
    spec = os.environ.get('RW_MICRO_BATCH', _DEFAULT_MICRO_BATCH)
    k, _, res = spec.partition(':')
    return int(k or 0), int(res or 256)


_DEFAULT_MICRO_BATCH = '0'


class DataBag(dict):
    """dict with attribute access, carrying latent / style / fmap / output / noise through the
    sequential generator (reference: utils/stylegan2/models.py:204-230)."""

    def __init__(self, rep=None, **kwargs):
        super().__init__()
        self.update(rep, **kwargs)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name)

    def update(self, rep=None, **kwargs):
        if rep is not None:
            super().update(rep)
        super().update(kwargs)


# ------------------------------------------------------------------------------------------
# leaves
# ------------------------------------------------------------------------------------------

class InputLatent(nn.Module):
    def forward(self, z):
        return DataBag(latent=z)


class ReturnOutput(nn.Module):
    def forward(self, d):
        return d.output


class PixelNormL(nn.Module):
    def forward(self, d):
        return DataBag(d, latent=hip.pixel_norm(d.latent))


class EqualLinear(nn.Linear):
    """Equalised-learning-rate linear layer (reference: models.py:487-517)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        self.bias_init = bias_init
        self.lr_mul = lr_mul
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        super().__init__(in_dim, out_dim, bias)
        self.activation = activation

    def reset_parameters(self):
        nn.init.normal_(self.weight, std=1.0 / self.lr_mul)
        if self.bias is not None:
            nn.init.constant_(self.bias, self.bias_init)

    def forward(self, input):
        return hip.equal_linear(input, self.weight, self.bias, self.scale, self.lr_mul,
                                act=bool(self.activation))

    def __repr__(self):
        return '%s(%d, %d)' % (type(self).__name__, self.weight.shape[1], self.weight.shape[0])


class EqualLinearL(EqualLinear):
    def forward(self, d):
        return DataBag(d, latent=super().forward(d.latent))


class EqualLinearS(EqualLinear):
    def forward(self, d):
        return DataBag(d, style=super().forward(d.style))


class AdjustLatent(nn.Module):
    """Truncation toward ``latent_avg`` and broadcast to one row per layer (models.py:570-583).
    As in the reference the buffer starts as a 0-d placeholder and truncation only applies once
    it has been given a real vector."""

    def __init__(self, n_latent, truncation=1.0):
        super().__init__()
        self.n_latent = n_latent
        self.truncation = truncation
        self.register_buffer('latent_avg', torch.tensor(0.0))

    def forward(self, d):
        truncate = self.truncation != 1.0 and self.latent_avg.ndim > 0
        lat = hip.adjust_latent(d.latent, self.latent_avg if truncate else None, self.n_latent,
                                self.truncation)
        return DataBag(d, latent=lat)


class PickLatent(nn.Module):
    def __init__(self, index):
        super().__init__()
        self.index = index

    def __repr__(self):
        return '%s(%d)' % (type(self).__name__, self.index)

    def forward(self, d):
        return DataBag(d, style=d.latent[:, self.index])


class NoiseBuffers(nn.Module):
    def __init__(self, replace_input=False):
        super().__init__()
        self.replace_input = replace_input

    def forward(self, d):
        for name, buf in self.named_buffers(recurse=False):
            if name.startswith('noise_') and (self.replace_input or name not in d):
                d[name] = buf
        return d


class FixedNoiseBuffers(NoiseBuffers):
    """Per-layer fixed noise images noise_0.. (models.py:342-352).  Present in the state dict for
    checkpoint compatibility; NoiseInjectionF never reads them (quirk Q1)."""

    def __init__(self, num_layers, seed, replace_input=False):
        super().__init__(replace_input=replace_input)
        self.num_layers = num_layers
        rng = np.random.RandomState(seed)
        for idx in range(num_layers):
            res = 2 ** ((idx + 5) // 2)
            self.register_buffer('noise_%d' % idx,
                                 torch.from_numpy(rng.randn(1, 1, res, res).astype('float32')))


class ConstantInputF(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, d):
        return DataBag(d, fmap=self.input.detach().repeat(d.latent.shape[0], 1, 1, 1))


class ApplyStyle(nn.Module):
    """fmap * style -- its output is the rewriter's key (rewrite/ganrewrite.py:662-665)."""

    def forward(self, d):
        return DataBag(d, fmap=grad.StyleMul.apply(d.fmap, d.style))


class DemodulatedConv2dF(nn.Module):
    """The plain linear convolution the rewriter edits, followed by the demodulation factor
    (models.py:291-329).  Stride 1: 3x3 conv, pad 1.  upsample: stride-2 transposed conv to
    (2H+1, 2W+1)."""

    def __init__(self, in_channel, out_channel, kernel_size, demodulate=True, upsample=False):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError('the gfx950 kernels implement 3x3 styled convolutions')
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.demodulate = demodulate
        self.upsample = upsample
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self._derived = _DerivedWeights()

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_derived'] = _DerivedWeights()       # caches are not copied or pickled
        return state

    def __repr__(self):
        return '%s(%d, %d, %d, upsample=%s)' % (type(self).__name__, self.in_channel,
                                                self.out_channel, self.kernel_size, self.upsample)

    def packed_weight(self):
        return self._derived.get('packed', self.weight,
                                 lambda: hip.pack_conv_weight(self.weight, 1 if self.upsample else 0))

    def wino_weight(self):
        return self._derived.get('wino', self.weight, lambda: hip.pack_conv_weight_wino(self.weight))

    def up_wino_weight(self, split=False):
        if split:
            return self._derived.get('upwino_split', self.weight,
                                     lambda: hip.pack_conv_transpose_weight_wino(self.weight, split=True))
        return self._derived.get('upwino', self.weight, lambda: hip.pack_conv_transpose_weight_wino(self.weight))

    def up_blur_wino4_weight(self, k4, split=False):
        if split:
            return self._derived.get('upblur4_split', self.weight,
                                     lambda: hip.pack_conv_transpose_blur_weight_wino4(self.weight, k4, split=True))
        return self._derived.get('upblur4', self.weight,
                                 lambda: hip.pack_conv_transpose_blur_weight_wino4(self.weight, k4))

    def one_pass_upsample(self, fmap, blur):
        """Transposed conv + blur + noise + activation in ONE pass (hip.conv_transpose3x3s2_blur_wino4: the four
        output-parity phases as virtual channels of the F(4x4,3x3) kernel) -- where F(4x4,3x3) runs at all (the
        un-hooked whole-generator forward, conv_algo() == 'winograd4') and where it pays: it does 1.44x the matrix work
        of the F(2,2) kernel and saves writing + reading the (2H+1)x(2W+1) map and the blur pass, which wins at
        <= 64 input channels (layer 17 of the 1024 model: 12.4 -> 9.x ms).  RW_UP_ALGO=winograd4 forces it wherever
        the shape allows, RW_UP_ALGO=direct / RW_UP_FUSED=0 turn it off."""
        algo = up_conv_algo()
        if not self.upsample or conv_impl() != 0 or conv_precision() != 'f32' or algo == 'direct':
            return False
        if os.environ.get('RW_UP_FUSED', '1') == '0' or tuple(blur.pad) != (1, 1) or tuple(blur.kernel.shape) != (4, 4):
            return False
        if not hip.conv_transpose_blur_wino4_supported(self.out_channel, self.in_channel, fmap.shape[-2],
                                                       fmap.shape[-1]):
            return False
        if algo == 'winograd4':
            return True
        return conv_algo() == 'winograd4' and self.in_channel <= _ONE_PASS_UP_MAX_IN

    def fused_upsample(self, fmap, blur):
        """Transposed conv + blur + noise + activation in one pass at the transposed convolution's OWN multiply count
        (hip.conv_transpose3x3s2_blur_fused, csrc/rw_tconv.hip: a direct sum on the 16-bit matrix pipe, the (2H+1)^2 map
        kept in LDS) -- the default for every upsampling layer it takes (w % 32 == 0, h % 16 == 0: 32^2 maps and up; at most
        RW_UP_FUSED2_MAX_IN input channels) inside the un-hooked whole-generator forward in split mode, and of hooked / sliced
        models too (the statistics sweeps; RW_UP_FUSED2_HOOKED=0: not there); RW_UP_FUSED2=0 brings back the two-pass /
        phase-kernel routes.  History (round 5): the first forms were
        +1.6 % on the forward and stayed opt-in because to_rgb_kernel on the second stream came back wrong beside them (its
        packed fp32 FMAs; the streaming kernels are compiled without packed fp32 math since: csrc/rw_ops.hip, first line);
        the persistent form with specialised waves on the layers of few channels and the one-workgroup-per-CU form on the
        others (both fill a CU's register file: nothing else runs beside them) are +7 % (profiles/r05r), and the sequence /
        stress / parity tests of the forward run with them."""
        if not self.upsample or conv_impl() != 0 or conv_precision() != 'f32' or up_conv_algo() == 'direct':
            return False
        if os.environ.get('RW_UP_FUSED2', '1') != '1':
            return False
        # the un-hooked forward: with the one-pass kinds of the split form; a hooked / sliced model (the statistics sweeps,
        # the rewriter's sub-models: no hook INSIDE this layer, or StyledConvSeq would not be on its fused path): where
        # the F(2,2) transposed convolutions run in the split form (matrix_mode('up')), unless RW_UP_FUSED2_HOOKED=0
        if not (_split_part('up1') if _rgb_branch.image_path
                else _split_part('up') and os.environ.get('RW_UP_FUSED2_HOOKED', '1') == '1'):
            return False
        if tuple(blur.pad) != (1, 1) or tuple(blur.kernel.shape) != (4, 4):
            return False
        if self.in_channel > int(os.environ.get('RW_UP_FUSED2_MAX_IN', _FUSED_UP_MAX_IN)):
            return False
        return hip.tconv_blur_supported(self.out_channel, self.in_channel, fmap.shape[-2], fmap.shape[-1])

    def direct16_weight(self):
        return self._derived.get('direct16', self.weight, lambda: hip.pack_conv_weight_direct16(self.weight))

    def up_blur_direct16_weight(self, k4):
        return self._derived.get('upblur_direct16', self.weight,
                                 lambda: hip.pack_conv_transpose_blur_weight_direct16(self.weight, k4))

    def wino4_weight(self, split=False):
        if split:
            return self._derived.get('wino4_split', self.weight,
                                     lambda: hip.pack_conv_weight_wino4(self.weight, split=True))
        return self._derived.get('wino4', self.weight, lambda: hip.pack_conv_weight_wino4(self.weight))

    def runs_split_wino4(self, h, w):
        """Will run() on a map of h x w execute the split-operand F(4x4,3x3) kernel (which reports max |result|)?"""
        return (not self.upsample and _split_part('w4') and conv_algo() == 'winograd4' and conv_impl() == 0
                and conv_precision() == 'f32' and hip.wino4_supported(self.out_channel, self.in_channel, h, w))

    def runs_small_direct16(self, h, w):
        """Inside the un-hooked forward: the 32^2 stride-1 layer -- too narrow for F(4x4,3x3); the fp32 F(2x2,3x3) kernel until
        round 6 -- as a direct sum on the 16-bit pipe (0.8 against 1.4 ms at batch 64; the forward +1.7 %, same box,
        interleaved: profiles/r06aq).  It measures its input itself (the layer in front runs an fp32 kernel that reports
        nothing) and reports the bound of its result like the other split-operand kernels.  RW_DIRECT16_SMALL=0: off."""
        return (not self.upsample and _rgb_branch.image_path and _split_part('w4')
                and os.environ.get('RW_DIRECT16_SMALL', '1') != '0' and conv_algo() == 'winograd4' and conv_impl() == 0
                and conv_precision() == 'f32' and min(h, w) >= 32
                and not hip.wino4_supported(self.out_channel, self.in_channel, h, w) and _direct16(self, h, w, 'conv'))

    def hooked_direct16(self, h, w):
        """A hooked / sliced model (statistics sweeps, goal maps, the solve's context): the stride-1 convolutions of maps from
        32^2 up as DIRECT sums on the 16-bit matrix pipe (exact f16 operand pairs, fp32 accumulation: 4e-7 from the fp32
        direct sum, the error class these models are held to) in place of the fp32 F(2x2,3x3) kernel -- where their F(2,2)
        transposed convolutions run in the split form too (matrix_mode('up')).  The kernel measures its input itself: nobody
        hands a bound over outside the un-hooked forward.  RW_DIRECT16_HOOKED=0: off.  Not where the weight itself is being
        optimised (run(weight_changes=True), from grad.DemodConv: an autograd `insert` changes it every step, and each
        re-packing reads the weights' maximum back to the host -- hip._split_scale -- two launches and a sync per layer and
        iteration that the fp32 kernel does not have) and not while the stream is being captured (the read-back would fail)."""
        if self.weight.is_cuda and torch.cuda.is_current_stream_capturing():
            return False
        return (not self.upsample and not _rgb_branch.image_path and os.environ.get('RW_DIRECT16_HOOKED', '1') == '1'
                and os.environ.get('RW_CONV_ALGO') is None and matrix_mode('up') == 'split' and conv_impl() == 0
                and conv_precision() == 'f32' and _direct16(self, h, w, 'conv'))

    def squared_sums(self):
        return self._derived.get('wsq', self.weight, lambda: hip.weight_sqsum(self.weight, self.scale))

    def demod_factors(self, style):
        return hip.demod(self.squared_sums(), style) if self.demodulate else None

    def specialised_direct16(self, h, w):
        """Inside the un-hooked forward: will this stride-1 layer run the direct sum with SPECIALISED waves (rw_dconv.hip's
        dconv_ws_w2 kernels: one persistent twelve-wave workgroup per CU; 64 out-channels and 64 columns per tile, a style
        on load)?  Then the layer in front does not pre-scale its result for it.  RW_DCONV_WS_FWD=0: the one-role kernels
        on a pre-scaled map, as in round 5."""
        return (not self.upsample and _rgb_branch.image_path and os.environ.get('RW_DCONV_WS_FWD', '1') != '0'
                and os.environ.get('RW_DCONV_V') != '1' and _split_part('w4') and conv_algo() == 'winograd4'
                and conv_impl() == 0 and conv_precision() == 'f32'
                and self.in_channel >= 32 and self.out_channel % 64 == 0 and h % 8 == 0 and w % 64 == 0
                and hip.wino4_supported(self.out_channel, self.in_channel, h, w) and _direct16(self, h, w, 'conv'))

    def leaves_rgb_partials(self, h, w):
        """Will run(..., rgb=...) on a map of h x w execute the direct-sum kernel that also leaves the channel sums of the
        ToRGB reading its result (hip.conv3x3_direct16_rgb_partial)?  Inside the un-hooked forward only; RW_RGB_PARTIAL=0:
        off (ToRGB then re-reads the feature map on the RGB stream, as before round 6)."""
        return (not self.upsample and _rgb_branch.image_path and os.environ.get('RW_RGB_PARTIAL', '1') != '0'
                and _split_part('w4') and conv_algo() == 'winograd4' and conv_impl() == 0 and conv_precision() == 'f32'
                and self.out_channel % 32 == 0
                and hip.wino4_supported(self.out_channel, self.in_channel, h, w) and _direct16(self, h, w, 'conv'))

    def run(self, fmap, style, style_on_load, demod=None, x_amax=None, y_amax=None, rgb=None, weight_changes=False,
            **epilogue):
        """x_amax: the bound of |fmap| if the producer of fmap left one (hip.new_bound; split-operand kernels -- they
        measure the map themselves otherwise); y_amax: a hip.new_bound buffer that receives the bound of the result
        where the split-operand F(4x4,3x3) kernel runs (`runs_split_wino4` says whether it will); rgb: only where
        `leaves_rgb_partials` says so; weight_changes: the caller differentiates with respect to the weight (see hooked_direct16)."""
        if rgb is not None and not self.leaves_rgb_partials(fmap.shape[-2], fmap.shape[-1]):
            raise RuntimeError('run(rgb=...) on a layer that does not leave ToRGB partial sums')
        if demod is None:
            demod = self.demod_factors(style)
        load_style = style if style_on_load else None
        split = _split_part('up' if self.upsample else 'w4')
        if self.upsample:
            # the border strips go to an auxiliary stream beside the tiles: the one of the running whole-generator
            # forward, else (a sliced or hooked model: the statistics sweeps, the rewriter's sub-models) the
            # device's own -- in a 250-seed sweep launch the three strip kernels were 17 % of the time, in line
            aux = _rgb_branch.aux
            if (aux is None and fmap.is_cuda and os.environ.get('RW_STRIP_STREAM', '1') != '0'
                    and not torch.cuda.is_current_stream_capturing()):
                aux = _rgb_side_streams.get((fmap.device, 'aux'))
                if aux is None:
                    aux = _rgb_side_streams[(fmap.device, 'aux')] = _side_stream(fmap.device)
            b, _, h, w = fmap.shape
            f22 = (up_conv_algo() == 'winograd' and conv_impl() == 0
                   and hip.up_strips_applicable(self.out_channel, self.in_channel)
                   and hip.conv_transpose_wino_supported(self.out_channel, self.in_channel, h, w))
            if conv_impl() == 0 and (f22 or (aux is not None and hip.up_halo_applicable(
                    self.out_channel, self.in_channel, w))):
                # quad tiles (F(2,2) where it applies, else the direct kernel) and the border row / column strips
                # write disjoint elements of the same map; in the un-hooked full forward the strips
                # (latency-bound, 2 % of the step) go to a third stream beside the tiles
                out = torch.empty(b, self.out_channel, 2 * h + 1, 2 * w + 1, device=fmap.device, dtype=fmap.dtype)
                wp = self.packed_weight()      # (re)packed on the trunk's stream BEFORE the fork
                f22_split = f22 and split and hip.conv_transpose_wino_split_supported(self.out_channel, self.in_channel, h, w)
                uf = self.up_wino_weight(f22_split) if f22 else None
                if f22_split and x_amax is None:
                    x_amax = hip.absmax(fmap)
                main = torch.cuda.current_stream() if aux is not None else None
                if aux is not None:
                    aux.wait_stream(main)
                    with torch.cuda.stream(aux):
                        hip.conv_transpose3x3s2(fmap, wp, self.out_channel, self.scale,
                                                style=load_style, demod=demod, impl=8, out=out)
                else:
                    hip.conv_transpose3x3s2(fmap, wp, self.out_channel, self.scale,
                                            style=load_style, demod=demod, impl=8, out=out)
                if f22_split:
                    hip.conv_transpose3x3s2_wino(fmap, uf, self.out_channel, self.scale, style=load_style,
                                                 demod=demod, out=out, x_amax=x_amax)
                elif f22:
                    hip.conv_transpose3x3s2_wino(fmap, uf, self.out_channel, self.scale, style=load_style,
                                                 demod=demod, out=out)
                else:
                    hip.conv_transpose3x3s2(fmap, wp, self.out_channel, self.scale,
                                            style=load_style, demod=demod, impl=7, out=out)
                if aux is not None:
                    main.wait_stream(aux)      # queued while fmap / style / demod / out are still referenced
                return out
            return hip.conv_transpose3x3s2(fmap, self.packed_weight(), self.out_channel, self.scale,
                                           style=load_style, demod=demod, impl=conv_impl())
        if (conv_algo() == 'winograd4' and conv_impl() == 0 and conv_precision() == 'f32'
                and hip.wino4_supported(self.out_channel, self.in_channel, fmap.shape[-2], fmap.shape[-1])):
            if split and _direct16(self, fmap.shape[-2], fmap.shape[-1], 'conv'):
                if rgb is not None:             # (weight (3, out), style (B, out), scale): returns (map, partial images)
                    return hip.conv3x3_direct16_rgb_partial(fmap, self.direct16_weight(), self.out_channel, self.scale,
                                                            rgb[0], rgb[1], rgb[2], style=load_style, demod=demod,
                                                            x_amax=x_amax, y_amax=y_amax, **epilogue)
                return hip.conv3x3_direct16(fmap, self.direct16_weight(), self.out_channel, self.scale, style=load_style,
                                            demod=demod, x_amax=x_amax, y_amax=y_amax, **epilogue)
            if split:
                return hip.conv3x3_wino4(fmap, self.wino4_weight(True), self.out_channel, self.scale, style=load_style,
                                         demod=demod, x_amax=x_amax, y_amax=y_amax, **epilogue)
            return hip.conv3x3_wino4(fmap, self.wino4_weight(), self.out_channel, self.scale, style=load_style,
                                     demod=demod, **epilogue)
        if self.runs_small_direct16(fmap.shape[-2], fmap.shape[-1]):
            return hip.conv3x3_direct16(fmap, self.direct16_weight(), self.out_channel, self.scale, style=load_style,
                                        demod=demod, x_amax=x_amax, y_amax=y_amax, **epilogue)
        if not weight_changes and self.hooked_direct16(fmap.shape[-2], fmap.shape[-1]):
            return hip.conv3x3_direct16(fmap, self.direct16_weight(), self.out_channel, self.scale, style=load_style,
                                        demod=demod, x_amax=x_amax, **epilogue)
        if (conv_algo() in ('winograd', 'winograd4') and conv_impl() == 0 and conv_precision() == 'f32'
                and hip.wino_supported(self.out_channel, self.in_channel, fmap.shape[-2], fmap.shape[-1])):
            return hip.conv3x3_wino(fmap, self.wino_weight(), self.out_channel, self.scale, style=load_style,
                                    demod=demod, **epilogue)
        if (conv_precision() == 'bf16x6' and conv_impl() == 0
                and hip.bf16x6_supported(self.out_channel, self.in_channel, fmap.shape[-1])):
            wb = self._derived.get('packed_bf16x3', self.weight, lambda: hip.pack_conv_weight_bf16x3(self.weight))
            return hip.conv3x3_bf16x6(fmap, wb, self.out_channel, self.scale, style=load_style, demod=demod,
                                      **epilogue)
        return hip.conv3x3(fmap, self.packed_weight(), self.out_channel, self.scale,
                           style=load_style, demod=demod, impl=conv_impl(), **epilogue)

    def forward(self, d):
        # through torch.autograd (grad.DemodConv: backward to the input map, the weight -- both terms, quirk Q3 --
        # and the style); without a graph this is run() and nothing else
        return DataBag(d, fmap=grad.DemodConv.apply(d.fmap, self.weight, d.style, self))


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        k = make_kernel(kernel)
        if upsample_factor > 1:
            k = k * (upsample_factor ** 2)
        self.register_buffer('kernel', k)
        self.pad = pad

    def forward(self, input):
        return op.upfirdn2d(input, self.kernel, pad=self.pad)


class BlurF(Blur):
    def forward(self, d):
        return DataBag(d, fmap=super().forward(d.fmap))


class Upsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return op.upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class UpsampleF(Upsample):
    def forward(self, d):
        return DataBag(d, fmap=super().forward(d.fmap))


class UpsampleO(Upsample):
    def __init__(self, kernel=[1, 3, 3, 1], factor=2):
        super().__init__(kernel, factor)

    def forward(self, d):
        side = _rgb_stream()
        if side is None:
            return DataBag(d, output=super().forward(d.output))
        with torch.cuda.stream(side):              # the previous RGB image was produced on this stream
            return DataBag(d, output=super().forward(d.output))


class NoiseInjectionF(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def noise_for(self, d, batch, height, width, device):
        noise = d.get('noise', None)
        if noise is None:
            rows = d.get('batch_rows', None)
            if rows is not None:        # a slice [start, start + batch) of a launch of `total` images keeps its rows
                start, total = rows
                return reference_noise(total, height * width, device)[start:start + batch]
            return reference_noise(batch, height * width, device)
        return noise.reshape(batch, height * width)

    def forward(self, d):
        b, _, h, w = d.fmap.shape
        return DataBag(d, fmap=grad.NoiseAdd.apply(d.fmap, self.noise_for(d, b, h, w, d.fmap.device), self.weight))


class FusedLeakyReLUF(op.FusedLeakyReLU):
    def forward(self, d):
        return DataBag(d, fmap=super().forward(d.fmap))


class ModulatedConv2d(nn.Module):
    """Style-modulated convolution taking (input, style) tensors (models.py:354-425).  3x3 uses
    the styled implicit-GEMM kernels (style folded into the gather, demodulation in the
    epilogue: the same arithmetic as modulating the weights first); 1x1 without demodulation is
    the ToRGB projection."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True,
                 upsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1),
                             upsample_factor=factor)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self._derived = _DerivedWeights()

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_derived'] = _DerivedWeights()
        return state

    def __repr__(self):
        return '%s(%d, %d, %d, upsample=%s, downsample=False)' % (
            type(self).__name__, self.in_channel, self.out_channel, self.kernel_size, self.upsample)

    def forward(self, input, style):
        style = self.modulation(style)
        if self.kernel_size == 1:
            if self.demodulate or self.upsample or self.out_channel != 3:
                raise NotImplementedError('1x1 modulated conv is implemented for ToRGB only')
            return hip.to_rgb(input, self.weight.view(3, self.in_channel), style, None, None, self.scale)
        if self.kernel_size != 3:
            raise NotImplementedError('kernel_size %d' % self.kernel_size)
        wp = self._derived.get('packed', self.weight,
                               lambda: hip.pack_conv_weight(self.weight, 1 if self.upsample else 0))
        demod = None
        if self.demodulate:
            wsq = self._derived.get('wsq', self.weight, lambda: hip.weight_sqsum(self.weight, self.scale))
            demod = hip.demod(wsq, style)
        if self.upsample:
            out = hip.conv_transpose3x3s2(input, wp, self.out_channel, self.scale, style=style,
                                          demod=demod, impl=conv_impl())
            return self.blur(out)
        return hip.conv3x3(input, wp, self.out_channel, self.scale, style=style, demod=demod,
                           impl=conv_impl())


class ModulatedConv2dF(ModulatedConv2d):
    def forward(self, d):
        return DataBag(d, fmap=super().forward(d.fmap, d.style))


class ToRGBF(nn.Module):
    """Modulated 1x1 projection to RGB + bias + running skip image (models.py:628-655)."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1], skip=False):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))
        self.skip = skip

    def forward(self, d):
        if d.get('fused_rgb') is not None:          # already computed in the epilogue of the last styled conv
            return DataBag(d, output=d.fused_rgb, fused_rgb=None)
        skip = d.output if self.skip else None
        if skip is not None and tuple(skip.shape[2:]) != tuple(d.fmap.shape[2:]):
            up = self.upsample if hasattr(self, 'upsample') else Upsample([1, 3, 3, 1]).to(skip.device)
            if _rgb_stream() is None:
                skip = up(skip)
            else:
                with torch.cuda.stream(_rgb_stream()):
                    skip = up(skip)
        conv = self.conv
        side = _rgb_stream()
        partials = d.get('rgb_partials')
        if partials is not None:                    # the producing convolution left this ToRGB's channel sums
            if side is None:
                raise RuntimeError('ToRGB partial sums outside the forward that produces them')
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = hip.rgb_combine(partials, self.bias.view(3), skip)
            _rgb_branch.keep.append((partials,))
            nd = DataBag(d, output=out)
            nd.pop('rgb_partials', None)        # bags that callers see never carry the key
            return nd
        if side is None:
            style = conv.modulation(d.style)
            out = hip.to_rgb(d.fmap, conv.weight.view(3, conv.in_channel), style, self.bias.view(3), skip,
                             conv.scale)
            return DataBag(d, output=out)
        ahead = _prefetched(self)
        side.wait_stream(torch.cuda.current_stream())      # the feature map and the latent come from the trunk
        with torch.cuda.stream(side):
            style = ahead if ahead is not None else conv.modulation(d.style)
            out = hip.to_rgb(d.fmap, conv.weight.view(3, conv.in_channel), style, self.bias.view(3), skip,
                             conv.scale)
        # The branch reads trunk tensors from another stream: they stay referenced until the join
        # (Tensor.record_stream would do, but it defers the allocator's reuse of multi-GB blocks
        # unpredictably and shows up as intermittent hipMalloc stalls at large batch).
        _rgb_branch.keep.append((d.fmap, d.style))
        return DataBag(d, output=out)


# ------------------------------------------------------------------------------------------
# containers
# ------------------------------------------------------------------------------------------

class ModulatedConv2dSeq(nn.Sequential):
    """modulation -> adain -> dconv [-> blur], explicitly separated so that the learned linear
    convolution can be rewritten as an associative memory (models.py:259-289)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True,
                 upsample=False, blur_kernel=[1, 3, 3, 1]):
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        steps = [
            ('modulation', EqualLinearS(style_dim, in_channel, bias_init=1)),
            ('adain', ApplyStyle()),
            ('dconv', DemodulatedConv2dF(in_channel, out_channel, kernel_size,
                                         demodulate=demodulate, upsample=upsample)),
        ]
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            steps.append(('blur', BlurF(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1),
                                        upsample_factor=factor)))
        super().__init__(OrderedDict(steps))


class StyledConvSeq(nn.Sequential):
    """mconv -> noise -> activate (models.py:232-257).  When nothing inside is hooked the block
    runs fused; otherwise child by child, so hooks and splits observe reference semantics."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False,
                 blur_kernel=[1, 3, 3, 1], demodulate=True, mconv=None):
        assert mconv in [None, 'seq', 'fast']
        conv_cls = ModulatedConv2dSeq if mconv == 'seq' else ModulatedConv2dF
        super().__init__(OrderedDict([
            ('mconv', conv_cls(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                               blur_kernel=blur_kernel, demodulate=demodulate)),
            ('noise', NoiseInjectionF()),
            ('activate', FusedLeakyReLUF(out_channel)),
        ]))

    def _fusable(self, d=None):
        if not fusion_enabled() or set(self._modules) != {'mconv', 'noise', 'activate'}:
            return False
        if torch.is_grad_enabled() and (
                any(p.requires_grad for p in self.mconv.dconv.parameters())
                or (d is not None and any(torch.is_tensor(t) and t.requires_grad for t in (d.get('fmap'), d.get('style'),
                                                                                           d.get('latent'))))):
            # somebody may differentiate through this layer (an `insert` whose target spans it, rewrite/ganrewrite.py:
            # 265-283; a linear_insert that froze every parameter and edits a layer in FRONT of this one: then only
            # the incoming map carries the graph): module by module, where every step carries its adjoint (grad.py) --
            # the fused path calls raw-pointer kernels that record no grad_fn.  Image generation and the statistics
            # sweeps run under no_grad and never come here.
            return False
        mconv = self.mconv
        if not isinstance(mconv, ModulatedConv2dSeq):
            return False
        want = {'modulation', 'adain', 'dconv'} | ({'blur'} if mconv.upsample else set())
        if set(mconv._modules) != want:
            return False
        return _unhooked(mconv, self.noise, self.activate, *mconv._modules.values())

    def _reads_bound(self, h, w):
        """Would this layer, fed a map of h x w inside the un-hooked forward, run a split-operand kernel -- i.e. use a
        bound on its input if the producer left one?  A wrong 'yes' costs the producer one small reduction launch, a wrong
        'no' makes this layer measure its input itself (hip.absmax): neither changes a result."""
        if not self._fusable() or matrix_mode() != 'split' or conv_impl() != 0 or conv_precision() != 'f32':
            return False
        dconv = self.mconv.dconv
        if not self.mconv.upsample:
            return dconv.runs_split_wino4(h, w) or dconv.runs_small_direct16(h, w)
        probe = torch.empty(0, dconv.in_channel, h, w, device='meta')
        if dconv.fused_upsample(probe, self.mconv.blur):
            return True
        if _split_part('up1') and dconv.one_pass_upsample(probe, self.mconv.blur):
            return True
        return (_split_part('up') and up_conv_algo() == 'winograd' and hip.up_strips_applicable(dconv.out_channel, dconv.in_channel)
                and hip.conv_transpose_wino_supported(dconv.out_channel, dconv.in_channel, h, w)
                and hip.conv_transpose_wino_split_supported(dconv.out_channel, dconv.in_channel, h, w))

    def _hands_over_prescaled(self, h, w):
        """Would this (stride-1) layer, fed a map of h x w, run an F(4x4,3x3) kernel with the style applied on load?
        Then the layer in front may multiply the style into its own result (see forward)."""
        mconv, act = self.mconv, self.activate
        if (os.environ.get('RW_PRESCALE', '1') == '0' or not self._fusable() or mconv.upsample
                or act.negative_slope != 0.2 or abs(act.scale - 2 ** 0.5) > 1e-12):
            return False
        dconv = mconv.dconv
        if dconv.specialised_direct16(h, w):
            return False            # that kernel's staging waves apply the style themselves (it needs one on load)
        return (conv_algo() == 'winograd4' and conv_impl() == 0 and conv_precision() == 'f32'
                and hip.wino4_supported(dconv.out_channel, dconv.in_channel, h, w))

    def forward(self, d):
        pre = d.get('prescaled')
        if not self._fusable(d):
            if pre is not None:
                raise RuntimeError('a pre-scaled feature map reached a layer that runs module by module')
            return super().forward(d)
        mconv, act = self.mconv, self.activate
        # `pre`: the layer in front already multiplied this layer's style into fmap (and computed it)
        ahead = _prefetched(self)
        style = pre if pre is not None else ahead[0] if ahead is not None else \
            mconv.modulation(DataBag(style=d.style)).style
        demod = ahead[1] if ahead is not None else None        # else: computed where it is used
        on_load = pre is None
        fmap = d.fmap
        b = fmap.shape[0]
        dconv = mconv.dconv
        if act.negative_slope != 0.2 or abs(act.scale - 2 ** 0.5) > 1e-12:
            if pre is not None:
                raise RuntimeError('a pre-scaled feature map reached a layer that runs module by module')
            return super().forward(d)
        d_in = d                    # as received (with the hand-over key, if any): what a re-entry must see
        x_amax = _amax_of(fmap)     # the producer's bound on |fmap| (split-operand kernels), else None
        if os.environ.get('RW_MM_NO_HANDOVER') == '1':
            x_amax = None
        if pre is not None:
            d = DataBag(d)
            d.pop('prescaled', None)
        split = matrix_mode() == 'split'
        # ... and this layer's bound for the next one, inside the un-hooked forward only (a hooked model under RW_MM=split
        # lets the kernels measure their inputs)
        reader = _rgb_branch.reader.get(id(self)) if _rgb_branch.image_path else None
        want_amax = split and reader is not None

        def y_bound(height, width):
            if not want_amax or not reader._reads_bound(height, width):
                return None
            return hip.new_bound(b * dconv.out_channel * height * width, fmap.device)
        y_amax = None
        y_amax_set = False
        post = None
        rgb_partials = None
        if mconv.upsample and pre is not None:
            raise RuntimeError('a pre-scaled feature map reached an upsampling layer')
        if mconv.upsample:
            # inside the un-hooked whole-generator forward the result is read by exactly one consumer, the next
            # styled convolution: where that one runs F(4x4,3x3) -- whose loop is bound by vector instructions beside
            # the MFMAs -- its style multiply (18 packed multiplies per 6x6 item) moves into this layer's epilogue
            nxt = _rgb_branch.successor.get(id(self)) if _rgb_branch.image_path and pre is None else None
            if nxt is not None and nxt[0]._hands_over_prescaled(2 * fmap.shape[2], 2 * fmap.shape[3]):
                nxt_ahead = _prefetched(nxt[0])
                post = nxt_ahead[0] if nxt_ahead is not None else \
                    nxt[0].mconv.modulation(DataBag(style=d.latent[:, nxt[1]])).style
            h, w = 2 * fmap.shape[2], 2 * fmap.shape[3]
            noise = self.noise.noise_for(d, b, h, w, fmap.device)
            y_amax = y_bound(h, w)
            if dconv.fused_upsample(fmap, mconv.blur):
                mm = dict(y_amax=y_amax) if y_amax is not None else {}
                if os.environ.get('RW_UP_FUSED2_JOIN') == '1' and _rgb_branch.stream is not None:
                    torch.cuda.current_stream().wait_stream(_rgb_branch.stream)      # see fused_upsample
                out = hip.conv_transpose3x3s2_blur_fused(
                    fmap, dconv.direct16_weight(), mconv.blur.kernel, dconv.out_channel, dconv.scale,
                    style=style, demod=demod if demod is not None else dconv.demod_factors(style), noise=noise,
                    noise_w=self.noise.weight, bias=act.bias, act=True, post_scale=post, x_amax=x_amax, **mm)
                y_amax_set = y_amax is not None
            elif dconv.one_pass_upsample(fmap, mconv.blur):
                split1 = _split_part('up1')
                mm = dict(x_amax=x_amax, y_amax=y_amax) if split1 else {}
                if split1 and _direct16(dconv, fmap.shape[2], fmap.shape[3], 'up'):
                    one_pass, packed = hip.conv_transpose3x3s2_blur_direct16, dconv.up_blur_direct16_weight(mconv.blur.kernel)
                else:
                    one_pass, packed = hip.conv_transpose3x3s2_blur_wino4, dconv.up_blur_wino4_weight(mconv.blur.kernel, split1)
                out = one_pass(
                    fmap, packed, dconv.out_channel, dconv.scale,
                    style=style, demod=demod if demod is not None else dconv.demod_factors(style), noise=noise,
                    noise_w=self.noise.weight, bias=act.bias, act=True, post_scale=post, **mm)
                y_amax_set = split1 and y_amax is not None
            else:
                wide = dconv.run(fmap, style, style_on_load=True, demod=demod, x_amax=x_amax)
                mm = dict(y_amax=y_amax) if y_amax is not None else {}
                out = hip.blur_noise_act(wide, mconv.blur.kernel, noise, self.noise.weight, act.bias, post_scale=post,
                                         **mm)
                y_amax_set = y_amax is not None
        else:
            h, w = fmap.shape[2:]
            noise = self.noise.noise_for(d, b, h, w, fmap.device)
            fin = _rgb_branch.final
            if (fin is not None and fin[0] is self and conv_impl() == 0 and conv_precision() == 'f32'
                    and hip.to_rgb_fusable(dconv.out_channel, dconv.in_channel, w)):
                # last styled conv of the un-hooked generator: ToRGB runs in its epilogue and the feature
                # map, which nothing else reads, is never written (models.py:639-655 fused)
                torgb, idx = fin[1], fin[2]
                skip = d.output if torgb.skip else None
                if skip is not None and tuple(skip.shape[2:]) != (h, w):
                    return self._unfused_final(d_in)
                main = torch.cuda.current_stream()
                if _rgb_branch.stream is not None:
                    main.wait_stream(_rgb_branch.stream)           # the running image comes from the RGB stream
                rgb_ahead = _prefetched(torgb)
                rgb_style = rgb_ahead if rgb_ahead is not None else torgb.conv.modulation(d.latent[:, idx])
                wino = (conv_algo() in ('winograd', 'winograd4') and dconv.out_channel == 32
                        and hip.wino_supported(dconv.out_channel, dconv.in_channel, h, w))
                wino4 = (conv_algo() == 'winograd4' and os.environ.get('RW_RGB_F4', '1') != '0'
                         and hip.wino4_to_rgb_supported(dconv.out_channel, dconv.in_channel, h, w))
                fused = (hip.conv3x3_wino4_to_rgb if wino4 else
                         hip.conv3x3_wino_to_rgb if wino else hip.conv3x3_to_rgb)
                split4 = wino4 and _split_part('w4')
                mm = dict(x_amax=x_amax) if split4 else {}
                direct = split4 and _direct16(dconv, h, w, 'rgb')
                if direct:
                    fused = hip.conv3x3_direct16_to_rgb
                _, rgb = fused(
                    fmap, dconv.direct16_weight() if direct else dconv.wino4_weight(split4) if wino4 else dconv.wino_weight() if wino
                    else dconv.packed_weight(),
                    dconv.out_channel, dconv.scale,
                    torgb.conv.weight.view(3, torgb.conv.in_channel), rgb_style, torgb.bias.view(3), skip,
                    torgb.conv.scale, style=style if on_load else None,
                    demod=demod if demod is not None else dconv.demod_factors(style), noise=noise,
                    noise_w=self.noise.weight, bias=act.bias, act=True, **mm)
                return DataBag(d, style=style, fmap=None, fused_rgb=rgb)
            y_amax = y_bound(h, w) if want_amax and (dconv.runs_split_wino4(h, w) or dconv.runs_small_direct16(h, w)) else None
            y_amax_set = y_amax is not None
            # the ToRGB that reads this layer's result (to_rgbK follows layer 2K): its channel sums are left by the
            # convolution itself where the direct-sum kernel runs -- the RGB stream then adds a few small images instead
            # of re-reading the feature map (4.3 GB for layer 16 at batch 64)
            tr = _rgb_branch.torgb.get(id(self)) if _rgb_branch.stream is not None else None
            rgb = None
            if tr is not None and dconv.leaves_rgb_partials(h, w):
                rgb_ahead = _prefetched(tr[0])
                rgb_style = rgb_ahead if rgb_ahead is not None else tr[0].conv.modulation(d.latent[:, tr[1]])
                rgb = (tr[0].conv.weight.view(3, tr[0].conv.in_channel), rgb_style, tr[0].conv.scale)
            out = dconv.run(fmap, style, style_on_load=on_load, demod=demod, x_amax=x_amax,
                            y_amax=y_amax if y_amax_set else None, rgb=rgb, noise=noise,
                            noise_w=self.noise.weight, bias=act.bias, act=True)
            if rgb is not None:
                out, rgb_partials = out
        # hand-overs, only inside the un-hooked forward: the bound rides on the tensor itself (_amax_of)
        extra = {}
        if rgb_partials is not None:
            extra['rgb_partials'] = rgb_partials
        if y_amax_set:
            out.rw_amax = (y_amax, out._version)
        if post is not None:        # bags that callers see never carry the key
            extra['prescaled'] = post
        return DataBag(d, style=style, fmap=out, **extra)

    def _unfused_final(self, d):
        saved, _rgb_branch.final = _rgb_branch.final, None
        try:
            return self.forward(d)
        finally:
            _rgb_branch.final = saved


class SeqStyleGAN2(nn.Sequential):
    """The whole generator as named sequential steps (models.py:31-141): bag_in, style, latents,
    noises, input, layer2, to_rgb1, then per resolution up_rgbK, layer(2j+1), layer(2j+2),
    to_rgbK, and output."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1],
                 lr_mlp=0.01, truncation=1.0, mconv=None, bag_input=False, bag_output=False):
        self.size = size
        self.style_dim = style_dim
        self.mconv = mconv
        self.bag_input = bag_input
        self.bag_output = bag_output
        cm = channel_multiplier
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm,
                         256: 64 * cm, 512: 32 * cm, 1024: 16 * cm}
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.n_latent = self.log_size * 2 - 2

        def styled(cin, cout, upsample=False):
            return StyledConvSeq(cin, cout, 3, style_dim, upsample=upsample,
                                 blur_kernel=blur_kernel, mconv=mconv)

        def picked(index, name, module):
            return nn.Sequential(OrderedDict([('lat%d' % index, PickLatent(index)), (name, module)]))

        mapping = [PixelNormL()] + [
            EqualLinearL(style_dim, style_dim, lr_mul=lr_mlp, activation='fused_lrelu')
            for _ in range(n_mlp)]
        c4 = self.channels[4]
        steps = [] if bag_input else [('bag_in', InputLatent())]
        steps += [
            ('style', nn.Sequential(*mapping)),
            ('latents', AdjustLatent(self.n_latent, truncation)),
            ('noises', FixedNoiseBuffers(self.num_layers, 1, replace_input=False)),
            ('input', ConstantInputF(c4)),
            ('layer2', picked(0, 'conv', styled(c4, c4))),
            ('to_rgb1', picked(1, 'rgb', ToRGBF(c4, style_dim, upsample=False))),
        ]
        cin, lat = c4, 1
        for level in range(3, self.log_size + 1):
            cout = self.channels[2 ** level]
            steps += [
                ('up_rgb%d' % (level - 2), UpsampleO()),
                ('layer%d' % (lat + 2), picked(lat, 'sconv', styled(cin, cout, upsample=True))),
                ('layer%d' % (lat + 3), picked(lat + 1, 'sconv', styled(cout, cout))),
                ('to_rgb%d' % (level - 1),
                 picked(lat + 2, 'rgb', ToRGBF(cout, style_dim, skip=True, upsample=False))),
            ]
            cin, lat = cout, lat + 2
        if not bag_output:
            steps.append(('output', ReturnOutput()))
        super().__init__(OrderedDict(steps))

    def forward(self, input):
        whole = (fusion_enabled() and torch.is_tensor(input) and not self.bag_output and not self.bag_input
                 and not _rgb_branch.image_path and _unhooked(*self.modules()))
        if not whole:
            return self._forward(input)
        _rgb_branch.image_path = True
        _rgb_branch.successor = self._successors()
        _rgb_branch.reader = self._readers()
        _rgb_branch.torgb = self._torgbs()
        try:
            return self._forward(input)
        finally:
            _rgb_branch.image_path = False
            _rgb_branch.successor = {}
            _rgb_branch.reader = {}
            _rgb_branch.torgb = {}

    def _forward(self, input):
        mb, from_res = micro_batch()
        if (mb and fusion_enabled() and torch.is_tensor(input) and not self.bag_output
                and not self.bag_input and input.shape[0] > mb and 'up_rgb%d' % (int(math.log2(from_res)) - 2)
                in self._modules and _rgb_branch.stream is None and _unhooked(*self.modules())):
            return self._forward_micro(input, mb, from_res)
        side_ok = (fusion_enabled() and os.environ.get('RW_RGB_STREAM', '1') != '0' and torch.is_tensor(input)
                   and input.is_cuda and not self.bag_output and _rgb_branch.stream is None
                   and not torch.cuda.is_current_stream_capturing()
                   and _unhooked(*self.modules()))
        if not side_ok:
            return super().forward(input)
        main = torch.cuda.current_stream()
        side = _rgb_side_streams.get(input.device)
        if side is None:
            side = _rgb_side_streams[input.device] = _side_stream(input.device)
        aux = _rgb_side_streams.get((input.device, 'aux'))
        if aux is None:
            aux = _rgb_side_streams[(input.device, 'aux')] = _side_stream(input.device)
        _rgb_branch.stream = side
        _rgb_branch.aux = aux
        _rgb_branch.final = self._final_pair()
        try:
            out = input
            for name, module in self._modules.items():
                out = module(out)
                if name == 'latents' and _rgb_branch.image_path and os.environ.get('RW_PREFETCH_STYLES', '1') != '0':
                    self._prefetch_modulations(out, aux)
        finally:
            _rgb_branch.stream = None
            _rgb_branch.aux = None
            _rgb_branch.final = None
            _rgb_branch.pre = {}
            _rgb_branch.pre_join = None
            main.wait_stream(side)                          # join: the image is complete on the caller's stream
            del _rgb_branch.keep[:]                      # freed to the trunk's pool AFTER the join is queued
        if torch.is_tensor(out):
            out.record_stream(main)
        return out

    def _forward_micro(self, z, mb, from_res):
        """The un-hooked generator on a large batch: the low-resolution steps on the whole batch (their launches
        need it to fill 256 CUs), the steps from resolution `from_res` up on `mb` images at a time, so that the
        feature maps handed from one kernel to the next (134 MB per image at 1024^2) are still in the 256 MB
        memory-side cache when the next kernel reads them instead of making a round trip through HBM.  The slices
        reuse the same allocator blocks, image rows keep their noise rows (`batch_rows`), results are those of
        the one-launch path."""
        mods = list(self._modules.values())
        k = list(self._modules).index('up_rgb%d' % (int(math.log2(from_res)) - 2))
        d = z
        for m in mods[:k]:
            d = m(d)
        total = d.latent.shape[0]
        out = None
        _rgb_branch.final = self._final_pair()
        try:
            for s in range(0, total, mb):
                e = min(s + mb, total)
                part = DataBag({key: (v[s:e] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == total else v)
                                for key, v in d.items()})
                part['batch_rows'] = (s, total)
                for m in mods[k:]:
                    part = m(part)
                if out is None:
                    out = part.new_empty((total,) + tuple(part.shape[1:]))
                out[s:e] = part
                del part
        finally:
            _rgb_branch.final = None
        return out

    def _prefetch_modulations(self, d, aux):
        """Every styled convolution's style (EqualLinearS of its latent row) and demodulation factors, and every
        ToRGB's style, depend on the latents and the weights only: inside the un-hooked forward they are all computed
        HERE, right after the mapping network, on the auxiliary stream beside the 4x4 / 8x8 layers -- instead of as
        ~60 launches of a few microseconds each BETWEEN the large convolutions, where each one is a kernel boundary on
        the trunk and two of them (in front of layers 15 / 17) sat 0.3 - 0.6 ms behind the grid-stride workgroups of
        the RGB branch waiting for a wave slot (rocprofv3 trace of round 2).  The first layer's are computed inline
        (it needs them at once); the trunk waits for the rest at its first use (_prefetched).  Values are those of
        the per-layer path: same kernels, same inputs."""
        pre = {}
        todo = []
        for name, mod in self._modules.items():
            if not isinstance(mod, nn.Sequential) or isinstance(mod, StyledConvSeq):
                continue
            kids = list(mod.children())
            if len(kids) != 2 or not isinstance(kids[0], PickLatent):
                continue
            if isinstance(kids[1], StyledConvSeq) and kids[1]._fusable() and isinstance(kids[1].mconv, ModulatedConv2dSeq):
                todo.append((kids[0].index, kids[1]))
            elif isinstance(kids[1], ToRGBF):
                todo.append((kids[0].index, kids[1]))
        if len(todo) < 3:
            return
        main = torch.cuda.current_stream()
        aux.wait_stream(main)                       # the latents come from the trunk
        with torch.cuda.stream(aux):
            for index, mod in todo[2:]:
                lat = d.latent[:, index]
                if isinstance(mod, ToRGBF):
                    pre[id(mod)] = mod.conv.modulation(lat)
                else:
                    style = mod.mconv.modulation(DataBag(style=lat)).style
                    pre[id(mod)] = (style, mod.mconv.dconv.demod_factors(style))
        _rgb_branch.pre = pre
        _rgb_branch.pre_join = aux

    def _successors(self):
        """{id(upsampling StyledConvSeq): (next StyledConvSeq, its latent index)} for the layer pairs
        'layer(2j+1)' (upsample) -> 'layer(2j+2)' that follow each other directly in this sequence."""
        out = {}
        names = list(self._modules)
        for a, b in zip(names, names[1:]):
            ma, mb = self._modules[a], self._modules[b]
            sa, sb = getattr(ma, 'sconv', None), getattr(mb, 'sconv', None)
            if not (isinstance(sa, StyledConvSeq) and isinstance(sb, StyledConvSeq)):
                continue
            if not getattr(sa.mconv, 'upsample', False) or getattr(sb.mconv, 'upsample', False):
                continue
            picks = [m for m in mb.children() if isinstance(m, PickLatent)]
            if len(picks) == 1 and list(mb.children())[0] is picks[0]:
                out[id(sa)] = (sb, picks[0].index)
        return out

    def _readers(self):
        """{id(StyledConvSeq): the next StyledConvSeq of this sequence} -- the layer that reads its feature map (the ToRGB
        / up_rgb steps in between pass it on untouched)."""
        out, prev = {}, None
        for mod in self._modules.values():
            conv = getattr(mod, 'sconv', None) or getattr(mod, 'conv', None)
            if isinstance(conv, StyledConvSeq):
                if prev is not None:
                    out[id(prev)] = conv
                prev = conv
        return out

    def _torgbs(self):
        """{id(StyledConvSeq): (ToRGBF, latent index)} for every 'layerN' directly followed by a 'to_rgbK' step whose first
        child picks the latent and whose second is the ToRGB (models.py:126-131)."""
        out = {}
        names = list(self._modules)
        for a, b in zip(names, names[1:]):
            sconv = getattr(self._modules[a], 'sconv', None) or getattr(self._modules[a], 'conv', None)
            rgbseq = self._modules[b]
            torgb = getattr(rgbseq, 'rgb', None)
            if not isinstance(sconv, StyledConvSeq) or not isinstance(torgb, ToRGBF):
                continue
            kids = list(rgbseq.children())
            if len(kids) == 2 and isinstance(kids[0], PickLatent) and kids[1] is torgb:
                out[id(sconv)] = (torgb, kids[0].index)
        return out

    def _final_pair(self):
        """(last StyledConvSeq, the ToRGBF that consumes it, latent index of that ToRGB) or None."""
        if os.environ.get('RW_FUSE_FINAL_RGB', '1') == '0':
            return None
        names = list(self._modules)
        last_rgb = 'to_rgb%d' % (self.log_size - 1)
        last_layer = 'layer%d' % (self.num_layers + 1)
        if last_rgb not in names or last_layer not in names or names.index(last_rgb) != names.index(last_layer) + 1:
            return None
        layer, rgbseq = self._modules[last_layer], self._modules[last_rgb]
        sconv = getattr(layer, 'sconv', None)
        torgb = getattr(rgbseq, 'rgb', None)
        picks = [m for m in rgbseq.children() if isinstance(m, PickLatent)]
        if not isinstance(sconv, StyledConvSeq) or not isinstance(torgb, ToRGBF) or len(picks) != 1:
            return None
        return sconv, torgb, picks[0].index

    def bag_from_z(self, z):
        return InputLatent()(z)

    def output_from_bag(self, bag):
        return ReturnOutput()(bag)

    def load_state_dict(self, data, latent_avg=None, **kwargs):
        """Accepts this class's own keys, or a rosinality/stylegan2-pytorch generator
        checkpoint (``{'g_ema': ..., 'latent_avg': ...}`` or its ``g_ema`` dict), renaming
        keys the way the reference does (models.py:149-202)."""
        try:
            return super().load_state_dict(data, **kwargs)
        except Exception:
            pass
        if len(data) < 10 and 'g_ema' in data and 'latent_avg' in data:
            latent_avg = data['latent_avg']
            data = data['g_ema']
        converted = convert_rosinality_keys(data, seq=(self.mconv == 'seq'))
        current = self.state_dict()
        if latent_avg is not None:
            converted['latents.latent_avg'] = latent_avg
        elif 'latents.latent_avg' not in converted:
            if self.latents.truncation != 1.0:
                warnings.warn('Need to provide latent_avg to use truncation.')
            converted['latents.latent_avg'] = current['latents.latent_avg']
        for key in current:
            if key.startswith('noises') and key not in converted:
                converted[key] = current[key]
        return super().load_state_dict(converted, **kwargs)


_ROSINALITY_RULES = [
    (r'^conv1\.conv\.', lambda m: 'layer2.conv.mconv.'),
    (r'^conv1\.', lambda m: 'layer2.conv.'),
    (r'^convs\.(\d+)\.conv', lambda m: 'layer%d.sconv.mconv' % (int(m.group(1)) + 3)),
    (r'^convs\.(\d+)\.', lambda m: 'layer%d.sconv.' % (int(m.group(1)) + 3)),
    (r'^to_rgb1\.(conv\.|bias$)', lambda m: 'to_rgb1.rgb.%s' % m.group(1)),
    (r'^to_rgbs\.(\d+)\.upsample\.', lambda m: 'up_rgb%d.' % (int(m.group(1)) + 1)),
    (r'^to_rgbs\.(\d+)\.', lambda m: 'to_rgb%d.rgb.' % (int(m.group(1)) + 2)),
]


def convert_rosinality_keys(data, seq=True):
    out = {}
    for key, value in data.items():
        for pattern, repl in _ROSINALITY_RULES:
            key = re.sub(pattern, repl, key)
        if seq:
            key = re.sub(r'mconv\.weight$', 'mconv.dconv.weight', key)
        else:
            key = re.sub(r'mconv\.dconv\.weight$', 'mconv.weight', key)
        out[key] = value
    return out
