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


class RunningTopK:
    """Running top-k values (and the global sample indexes they came from) for every feature
    (reference surface: utils/runningstats.py:31-146 -- ``add(data[, index])``, ``result()`` ->
    (values, indexes) with k last, ``size()``, ``to_``, ``state_dict``).  Candidates are kept in a
    (features, <= 5k) buffer that is cut back to the best k whenever it fills."""

    def __init__(self, k=100, state=None):
        if state is not None:
            self.set_state_dict(resolve_state_dict(state))
            return
        self.k = k
        self.count = 0
        self.data_shape = None
        self.top_data = None
        self.top_index = None

    def add(self, data, index=None):
        size = data.shape[0]
        if self.data_shape is None:
            self.data_shape = tuple(data.shape[1:])
        flat = data.detach().reshape(size, -1).t()                # (features, size)
        take = min(size, self.k)
        vals, idx = flat.topk(take, dim=1, sorted=False)
        idx = index.to(idx.device)[idx] if index is not None else idx + self.count
        if self.top_data is None:
            self.top_data, self.top_index = vals.clone(), idx.clone()
        else:
            self.top_data = torch.cat([self.top_data, vals], dim=1)
            self.top_index = torch.cat([self.top_index, idx], dim=1)
        if self.top_data.shape[1] > max(10, 5 * self.k):
            self.top_data, self.top_index = self.result(sorted=False, flat=True)
        self.count += size

    def size(self):
        return self.count

    def result(self, sorted=True, flat=False):
        k = min(self.k, self.top_data.shape[1])
        vals, pos = self.top_data.topk(k, dim=1, sorted=sorted)
        idx = self.top_index.gather(1, pos)
        if flat:
            return vals, idx
        shape = tuple(self.data_shape) + (-1,)
        return vals.reshape(shape), idx.reshape(shape)

    def to_(self, device):
        self.top_data = self.top_data.to(device)
        self.top_index = self.top_index.to(device)

    def state_dict(self):
        vals, idx = self.result(sorted=True, flat=True)
        return dict(constructor=self.__module__ + '.' + self.__class__.__name__ + '()',
                    k=self.k, count=self.count, data_shape=tuple(self.data_shape),
                    top_data=vals.cpu().numpy(), top_index=idx.cpu().numpy())

    def set_state_dict(self, dic):
        self.k = _item(dic['k'])
        self.count = _item(dic['count'])
        self.data_shape = tuple(int(v) for v in dic['data_shape'])
        self.top_data = torch.from_numpy(numpy.asarray(dic['top_data']))
        self.top_index = torch.from_numpy(numpy.asarray(dic['top_index']))


class RunningQuantile:
    """Per-unit quantiles of a stream of (samples, units) batches.

    The reference keeps a randomised, resolution-``r`` reservoir sketch while it streams (utils/runningstats.py:
    269-620) because a 2020 GPU could not hold the samples.  With 288 GB of HBM the samples of every sweep on
    this path fit (1000 seeds x 32x32 x 512 units = 2 GB), so this class keeps them while the sweep runs, sorts
    ONCE, and then holds the statistic in the reference's own form: levels of retained samples, a sample of
    level l standing for 2^l samples, plus the exact extremes (:426-438).  ``compress_()`` picks the retained
    samples deterministically (``retained_order_statistics``): at most 2r per unit (the reference's
    ``resolution``), dense in the tails and strided in the body.  Up to 2r samples that is the sample itself and
    the answers are exact; beyond it the rank error is below 2^(cap-1)/n in the body (1.2e-4 for 1000 seeds at
    32x32) and at most 1/256 of the distance to the nearer end in the tails, where the reference's sketch is
    random with ~1e-3 everywhere.

    ``state_dict()`` / ``RunningQuantile(state=...)`` speak the reference's schema (keys resolution, depth,
    buffersize, samplerate, data, sizes, extremes, size, batchcount), so a ``unit_rq.npz`` written by either
    implementation loads in the other.  Read-out follows :550-575 and :598-620: piecewise linear through
    (0, min), (midpoint cumulative weight, sample), (1, max).  A ``max_bytes`` guard raises instead of silently
    approximating while samples are being collected."""

    def __init__(self, r=4096, buffersize=None, seed=None, state=None, max_bytes=64 << 30):
        if state is not None:
            self.set_state_dict(resolve_state_dict(state))
            return
        self.resolution = 2 * r          # the reference's name for the retained-sample budget (:298-299)
        self.buffersize = buffersize if buffersize is not None else min(128, (self.resolution + 7) // 8)
        self.max_bytes = max_bytes
        self.count = 0
        self.batchcount = 0
        self.depth = None
        self._chunks = []               # exact samples, (units, n) pieces, while collecting
        self._sorted = None
        self._levels = None             # after compress_() / a cache load: [(units, n_l)] with weight 2^l
        self._extremes = None
        self._table = None

    # ------------------------------------------------------------------ collecting
    def add(self, incoming):
        """More samples.  Into a fresh statistic: kept exactly.  Into one that was compressed or loaded from a
        cache (the reference's sketch keeps streaming, utils/runningstats.py:340-383): the new samples are kept
        exactly BESIDE the levels, each with weight 1 like a level-0 sample of the reference's representation;
        read-outs merge the two, and the next compress_() reduces the weighted union to the budget."""
        if incoming.dim() == 1:
            incoming = incoming[:, None]
        assert incoming.dim() == 2
        if self.depth is None:
            self.depth = incoming.shape[1]
        assert incoming.shape[1] == self.depth
        self._chunks.append(incoming.detach().t().contiguous())        # (units, samples)
        self._sorted = None
        self._table = None
        self.count += incoming.shape[0]
        self.batchcount += 1
        if sum(c.shape[1] for c in self._chunks) * self.depth * 4 > self.max_bytes:
            raise MemoryError('RunningQuantile holds %d x %d samples (> max_bytes); pass a smaller '
                              'sample_size or raise max_bytes' % (self.count, self.depth))

    def size(self):
        return self.count

    def _data(self):
        if self._sorted is None:
            self._sorted = torch.cat(self._chunks, dim=1).sort(dim=1)[0]
            self._chunks = [self._sorted]
        return self._sorted

    @staticmethod
    def retained_order_statistics(n, budget):
        """Which order statistics of a sorted sample of n stand for it, per level: the sample is cut into
        consecutive groups of 2^l ranks and the middle member of each group is retained at level l (weight 2^l,
        so the weights sum to n exactly).  From each end: T = budget/32 groups of 1, T of 2, T of 4, ... up to
        2^(cap-1); the body between the tails in groups of 2^cap (its remainder in binary, smaller groups); cap
        is the smallest that keeps the retained count within the budget.  Rank error of a read-out: below
        2^(cap-1) in the body, and in the tails at most 1/T of the distance to the nearer end (exact for the T
        extreme values of each side) -- the tails are where -log(1 - rank) scores and the 0.99 / 0.999
        thresholds of this path read."""
        if n <= budget:
            return [numpy.arange(n, dtype=numpy.int64)]
        per = max(budget // 32, 1)
        while True:
            for cap in range(1, 62):
                tail = per * ((1 << cap) - 1)
                if 2 * tail > n:
                    break
                body = n - 2 * tail
                if 2 * per * cap + (body >> cap) + bin(body & ((1 << cap) - 1)).count('1') <= budget:
                    break
            else:
                cap = None
            if cap is not None and 2 * per * ((1 << cap) - 1) <= n:
                break
            per //= 2
            if per == 0:
                raise ValueError('no retained set of %d order statistics for a sample of %d' % (budget, n))
        levels = [[] for _ in range(cap + 1)]
        steps = numpy.arange(per, dtype=numpy.int64)
        pos = 0
        for l in range(cap):                                   # low tail, upper-middle member
            levels[l].append(pos + (1 << l) // 2 + (steps << l))
            pos += per << l
        body = n - 2 * per * ((1 << cap) - 1)
        levels[cap].append(pos + (1 << cap) // 2 + (numpy.arange(body >> cap, dtype=numpy.int64) << cap))
        pos += (body >> cap) << cap
        for l in reversed(range(cap)):
            if body & (1 << l):
                levels[l].append(numpy.array([pos + (1 << l) // 2], dtype=numpy.int64))
                pos += 1 << l
        for l in reversed(range(cap)):                         # high tail, mirrored: lower-middle member
            levels[l].append(pos + ((1 << l) - 1) // 2 + (steps << l))
            pos += per << l
        assert pos == n
        return [numpy.concatenate(lv) for lv in levels]

    def _representation(self):
        """(levels, extremes) of the reference's representation for the current state, WITHOUT changing it:
        exact sorted sample -> retained order statistics (see retained_order_statistics); levels plus samples
        added since -> the same order statistics of the weighted union, a level-l value standing for 2^l equal
        samples (so the result depends only on the union, not on when compress_() ran)."""
        if self.count == 0:
            return None, None
        if self._levels is None:
            s = self._data()
            return ([s[:, torch.from_numpy(idx).to(s.device)].contiguous()
                     for idx in self.retained_order_statistics(self.count, self.resolution)],
                    torch.stack([s[:, 0], s[:, -1]], dim=1))
        if not self._chunks:
            return self._levels, self._extremes
        new = self._data()
        dev = new.device
        vals = torch.cat([lv.to(dev, new.dtype) for lv in self._levels] + [new], dim=1)
        wts = torch.cat([torch.full((lv.shape[1],), 1 << l, dtype=torch.int64, device=dev)
                         for l, lv in enumerate(self._levels)]
                        + [torch.ones(new.shape[1], dtype=torch.int64, device=dev)])
        vals, order = vals.sort(dim=1, stable=True)
        cum = wts[order].cumsum(dim=1)                           # (units, m): ranks < cum[j] belong to value j
        total = int(cum[0, -1].item())                           # = the samples represented (self.count)
        ext = self._extremes.to(dev, new.dtype)
        ext = torch.stack([torch.minimum(ext[:, 0], new[:, 0]), torch.maximum(ext[:, 1], new[:, -1])], dim=1)
        levels = []
        for idx in self.retained_order_statistics(total, self.resolution):
            ranks = torch.from_numpy(idx).to(dev)[None, :].expand(vals.shape[0], -1).contiguous()
            pick = torch.searchsorted(cum, ranks, right=True).clamp_(max=vals.shape[1] - 1)
            levels.append(vals.gather(1, pick).contiguous())
        return levels, ext

    def compress_(self):
        """Current state -> levels of the reference's representation; the exact samples are dropped."""
        if self.count == 0 or (self._levels is not None and not self._chunks):
            return self
        self._levels, self._extremes = self._representation()
        self._chunks, self._sorted, self._table = [], None, None
        return self

    # ------------------------------------------------------------------ read-out
    def _weighted_table(self):
        """(values, position): per unit the sorted retained samples between the extremes and the cumulative
        weight at the middle of each, normalised to [0, 1] (utils/runningstats.py:530-563)."""
        if self._table is None:
            parts = list(self._levels)
            weights = [2.0 ** l for l in range(len(parts))]
            ext = self._extremes
            if self._chunks:                                     # samples added after the compression: weight 1
                new = self._data()
                parts = [p.to(new.device, new.dtype) for p in parts] + [new]
                weights.append(1.0)
                ext = ext.to(new.device, new.dtype)
                ext = torch.stack([torch.minimum(ext[:, 0], new[:, 0]), torch.maximum(ext[:, 1], new[:, -1])], dim=1)
            vals = torch.cat(parts, dim=1)
            wts = torch.cat([torch.full((lv.shape[1],), wt, dtype=torch.float64, device=vals.device)
                             for wt, lv in zip(weights, parts)])
            vals, order = vals.sort(dim=1)
            wts = wts[order]
            ext = ext.to(vals.device, vals.dtype)
            zero = wts.new_zeros(self.depth, 1)
            vals = torch.cat([ext[:, :1], vals, ext[:, 1:]], dim=1)
            wts = torch.cat([zero, wts, zero], dim=1)
            pos = (wts.cumsum(dim=1) - wts / 2) / wts.sum(dim=1, keepdim=True)
            self._table = (vals.contiguous(), pos.contiguous())
        return self._table

    def quantiles(self, quantiles, old_style=False):
        """old_style is accepted for call compatibility: with the zero-weight extremes at both ends the
        reference's two conventions (:556-561) coincide."""
        q = torch.as_tensor(quantiles, dtype=torch.float64)
        qshape = tuple(q.shape)
        if self.count == 0:
            return torch.full((self.depth or 0,) + qshape, float('nan'))
        if self._levels is None:
            s = self._data()
            n = self.count
            # piecewise linear through (0, min), ((i+.5)/n, s_i), (1, max); s_0 and s_{n-1} ARE the extremes
            t = (q.reshape(-1).to(s.device) * n - 0.5).clamp(0, n - 1)
            i0 = t.floor().long().clamp(0, n - 1)
            i1 = (i0 + 1).clamp(0, n - 1)
            frac = (t - i0.double()).to(s.dtype)
            out = s[:, i0] * (1 - frac) + s[:, i1] * frac
            return out.reshape((self.depth,) + qshape)
        vals, pos = self._weighted_table()
        m = vals.shape[1]
        qq = q.reshape(1, -1).to(vals.device).clamp(0, 1).expand(self.depth, -1).contiguous()
        hi = torch.searchsorted(pos, qq, right=True).clamp(1, m - 1)
        lo = hi - 1
        p0, p1 = pos.gather(1, lo), pos.gather(1, hi)
        frac = ((qq - p0) / (p1 - p0).clamp_min(1e-300)).clamp(0, 1).to(vals.dtype)
        out = vals.gather(1, lo) * (1 - frac) + vals.gather(1, hi) * frac
        return out.reshape((self.depth,) + qshape)

    def percentiles(self, percentiles):
        return self.quantiles(percentiles, old_style=True)

    def minmax(self):
        if self._levels is not None:
            vals, _ = self._weighted_table()
            return torch.stack([vals[:, 0], vals[:, -1]], dim=1)
        s = self._data()
        return torch.stack([s[:, 0], s[:, -1]], dim=1)

    def median(self):
        return self.quantiles([0.5])[:, 0]

    def integrate(self, fun):
        """sum over the represented samples of fun(sample) (utils/runningstats.py:577-591)."""
        if self._levels is None:
            return fun(self._data()).sum(dim=-1)
        total = sum(fun(lv).sum(dim=-1) * (2.0 ** l) for l, lv in enumerate(self._levels) if lv.shape[1])
        return total + fun(self._data()).sum(dim=-1).to(total.device) if self._chunks else total

    def mean(self):
        return self.integrate(lambda x: x) / self.count

    def variance(self):
        mean = self.mean()[:, None]
        return self.integrate(lambda x: (x - mean).pow(2)) / (self.count - 1)

    def stdev(self):
        return self.variance().sqrt()

    def readout(self, count=1001, old_style=True):
        return self.quantiles(torch.linspace(0.0, 1.0, count), old_style=old_style)

    def normalize(self, data):
        """data (units, ...) -> its quantile in [0, 1] within each unit's distribution."""
        assert self.count > 0 and data.shape[0] == self.depth
        if self._levels is None:
            s = self._data()
            n = self.count
            flat = data.detach().reshape(self.depth, -1).to(s.device, s.dtype).contiguous()
            # numpy.interp's bracket (:598-620): the LAST sample <= x on the left, so ties resolve upward
            hi = torch.searchsorted(s, flat, right=True).clamp(1, max(n - 1, 1)).clamp(max=n - 1)
            lo = (hi - 1).clamp(min=0)
            x0, x1 = s.gather(1, lo), s.gather(1, hi)
            frac = ((flat - x0) / (x1 - x0).clamp_min(1e-30)).clamp(0, 1)
            q = ((lo.to(s.dtype) + 0.5) + frac) / n
            q = torch.where(flat < s[:, :1], torch.zeros_like(q), q)
            q = torch.where(flat >= s[:, -1:], torch.ones_like(q), q)
            return q.clamp_(0, 1).float().reshape(data.shape).to(data.device)
        vals, pos = self._weighted_table()
        m = vals.shape[1]
        flat = data.detach().reshape(self.depth, -1).to(vals.device, vals.dtype).contiguous()
        hi = torch.searchsorted(vals, flat, right=True).clamp(1, m - 1)
        lo = hi - 1
        x0, x1 = vals.gather(1, lo), vals.gather(1, hi)
        frac = ((flat - x0) / (x1 - x0).clamp_min(1e-30)).clamp(0, 1).double()
        q = pos.gather(1, lo) * (1 - frac) + pos.gather(1, hi) * frac
        q = torch.where(flat < vals[:, :1], torch.zeros_like(q), q)
        q = torch.where(flat >= vals[:, -1:], torch.ones_like(q), q)
        return q.clamp_(0, 1).float().reshape(data.shape).to(data.device)

    def to_(self, device):
        self._chunks = [c.to(device) for c in self._chunks]
        self._sorted = None if self._sorted is None else self._sorted.to(device)
        if self._levels is not None:
            self._levels = [lv.to(device) for lv in self._levels]
            self._extremes = self._extremes.to(device)
            self._table = None

    # ------------------------------------------------------------------ the reference's cache schema
    def state_dict(self):
        """utils/runningstats.py:422-437.  `data` holds one (retained, units) array per level; it is stored as an
        object array because the levels differ in length.  Saving does not change the statistic: the exact
        samples collected so far stay, and more can be added afterwards."""
        levels, extremes = self._representation()
        levels = levels if levels is not None else []
        data = numpy.empty(len(levels), dtype=object)
        for l, lv in enumerate(levels):
            data[l] = lv.t().contiguous().cpu().numpy()
        extremes = (extremes.cpu().numpy() if extremes is not None
                    else numpy.zeros((self.depth or 0, 2), dtype=numpy.float32))
        return dict(constructor=self.__module__ + '.' + self.__class__.__name__ + '()',
                    resolution=self.resolution, depth=self.depth, buffersize=self.buffersize, samplerate=1.0,
                    data=data, sizes=[max(self.resolution, lv.shape[1]) for lv in levels],
                    extremes=extremes, size=self.count, batchcount=self.batchcount)

    def set_state_dict(self, dic):
        self.resolution = int(_item(dic['resolution']))
        self.depth = int(_item(dic['depth']))
        self.buffersize = int(_item(dic['buffersize']))
        self.count = int(_item(dic['size']))
        self.batchcount = int(_item(dic['batchcount'])) if 'batchcount' in dic else 0
        self.max_bytes = 64 << 30
        self._chunks, self._sorted, self._table = [], None, None
        levels = []
        for d in dic['data']:           # a list, an object array, or (equal lengths) one stacked 3-d array
            d = numpy.asarray(d)
            levels.append(torch.from_numpy(numpy.ascontiguousarray(d.reshape(-1, self.depth).T)))
        self._levels = levels
        # the reference folds the samples still in its level-0 buffer into the extremes lazily, at read-out
        # (:527-528, :459-461); a saved state may not have seen them yet
        ext = torch.from_numpy(numpy.asarray(dic['extremes'])).clone()
        for lv in levels:
            if lv.shape[1]:
                ext[:, 0] = torch.minimum(ext[:, 0], lv.min(dim=1)[0].to(ext.dtype))
                ext[:, 1] = torch.maximum(ext[:, 1], lv.max(dim=1)[0].to(ext.dtype))
        self._extremes = ext
