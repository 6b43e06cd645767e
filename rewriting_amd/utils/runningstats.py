"""Streaming statistics kept on the GPU: the two objects the rewriting hot path uses.

* ``RunningSecondMoment`` -- uncentred second moment ``sum a^T a / n`` (the key covariance
  ``C``; reference: utils/runningstats.py:1072-1120).  The reference accumulates with
  ``addbmm_`` over rows x 1 x C outer products; here ``add`` calls the fp32-MFMA split-K GEMM
  ``rw_second_moment_f32``, and ``add_nchw`` consumes the key map in the layout the generator
  produced it (no permute copy).
* ``RunningVariance`` -- per-unit mean/variance with Chan's parallel merge
  (reference: :748-819); the per-batch reductions run in ``rw_channel_sums_f32``.

``state_dict`` / ``set_state_dict`` keep the reference's npz schema (keys ``constructor``,
``count``, ``mom2`` / ``batchcount``, ``mean``, ``cmom2``) so caches written by either
implementation load in the other (checkpoint/resume).

Device rule: GPU samples always go through the HIP kernels (a missing library raises).  CPU
samples -- BASELINE.json's config 1, "ProgGAN-256 ... on PyTorch CPU, no GPU" -- use torch ops.
"""
import numpy
import torch

from .. import hip


def resolve_state_dict(s):
    if isinstance(s, str):
        return numpy.load(s, allow_pickle=True)
    return s


def _item(v):
    return v.item() if hasattr(v, 'item') else v


class RunningSecondMoment:
    def __init__(self, state=None):
        if state is not None:
            self.set_state_dict(resolve_state_dict(state))
            return
        self.count = 0
        self.mom2 = None

    def _ensure(self, channels, like):
        if self.mom2 is None:
            self.mom2 = torch.zeros(channels, channels, dtype=like.dtype, device=like.device)

    def add(self, a):
        """a: (rows, C) samples."""
        if a.dim() == 1:
            a = a[None, :]
        self._ensure(a.shape[1], a)
        self.count += a.shape[0]
        if hip.on_device(a):
            hip.second_moment_accumulate(self.mom2, a, nchw=False)
        else:
            self.mom2.addmm_(a.t(), a)          # PyTorch-CPU configuration (ProgGAN, config 1)

    def add_nchw(self, acts):
        """acts: (B, C, H, W) key map; rows are its pixels."""
        self._ensure(acts.shape[1], acts)
        self.count += acts.shape[0] * acts.shape[2] * acts.shape[3]
        if hip.on_device(acts):
            hip.second_moment_accumulate(self.mom2, acts, nchw=True)
        else:
            a = acts.permute(0, 2, 3, 1).reshape(-1, acts.shape[1])
            self.mom2.addmm_(a.t(), a)

    def merge_(self, mom2, count):
        """Adds another shard's raw sums (used after an all-reduce)."""
        self._ensure(mom2.shape[0], mom2)
        self.mom2 += mom2.to(self.mom2.device)
        self.count += int(count)

    def cpu_(self):
        self.to_('cpu')

    def cuda_(self):
        self.to_('cuda')

    def to_(self, device):
        if self.mom2 is not None:
            self.mom2 = self.mom2.to(device)

    def size(self):
        return self.count

    def moment(self):
        return self.mom2 / self.count

    def state_dict(self):
        return dict(constructor=self.__module__ + '.' + self.__class__.__name__ + '()',
                    count=self.count, mom2=self.mom2.cpu().numpy())

    def set_state_dict(self, dic):
        self.count = _item(dic['count'])
        self.mom2 = torch.from_numpy(numpy.asarray(dic['mom2']))


class RunningVariance:
    def __init__(self, state=None):
        if state is not None:
            self.set_state_dict(resolve_state_dict(state))
            return
        self.count = 0
        self.batchcount = 0
        self._mean = None
        self.v_cmom2 = None

    @staticmethod
    def _batch_stats(a, nchw, square_input):
        """(n, batch mean, batch centred sum of squares) per unit."""
        if hip.on_device(a):
            sums = hip.channel_sums(a, nchw=nchw, square_input=square_input)
            n = a.numel() // sums.shape[1]
            mean = sums[0] / n
            return n, mean, (sums[1] - n * mean * mean).clamp_(min=0)
        if nchw:
            a = a.permute(0, 2, 3, 1).reshape(-1, a.shape[1])
        if square_input:
            a = a.pow(2)
        n = a.shape[0]
        mean = a.sum(0) / n
        return n, mean, (a - mean).pow(2).sum(0)

    def add(self, a, nchw=False, square_input=False):
        if not nchw:
            if a.dim() == 1:
                a = a[None, :]
            if a.dim() > 2:
                a = a.reshape(a.shape[0], a.shape[1], -1).permute(0, 2, 1).reshape(-1, a.shape[1])
        n, mean, cm2 = self._batch_stats(a, nchw, square_input)
        self.batchcount += 1
        if self._mean is None:
            self.count, self._mean, self.v_cmom2 = n, mean, cm2
            return
        old = self.count
        self.count += n
        frac = float(n) / self.count
        delta = (mean - self._mean) * frac
        self._mean = self._mean + delta
        # bug-compatible with the reference (utils/runningstats.py:786-788): its cross term uses the
        # already frac-scaled delta, i.e. d^2 f^3 n_old rather than Chan's d^2 f n_old.  Only
        # mean() is consumed on the hot path; the formula is kept so unit_rs.npz caches agree.
        self.v_cmom2 = self.v_cmom2 + cm2 + delta.pow(2) * (frac * old)

    def size(self):
        return self.count

    def mean(self):
        return self._mean

    def variance(self):
        return self.v_cmom2 / (self.count - 1)

    def stdev(self):
        return self.variance().sqrt()

    def to_(self, device):
        self._mean = self._mean.to(device)
        self.v_cmom2 = self.v_cmom2.to(device)

    def state_dict(self):
        return dict(constructor=self.__module__ + '.' + self.__class__.__name__ + '()',
                    count=self.count, batchcount=self.batchcount,
                    mean=self._mean.cpu().numpy(), cmom2=self.v_cmom2.cpu().numpy())

    def set_state_dict(self, dic):
        self.count = _item(dic['count'])
        self.batchcount = _item(dic['batchcount'])
        self._mean = torch.from_numpy(numpy.asarray(dic['mean']))
        self.v_cmom2 = torch.from_numpy(numpy.asarray(dic['cmom2']))
