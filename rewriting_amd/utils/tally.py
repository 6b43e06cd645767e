"""Batched dataset sweeps feeding running statistics, with .npz caching.

The part of utils/tally.py on the rewriting hot path: ``tally_second_moment`` (:424-443),
``tally_mean`` (:252-272), the fixed-order loader (``make_loader`` :631-647, batches of 10 in
index order -- the batch grouping decides which noise row a seed receives, quirk Q1) and the
cache protocol (``load_cached_state`` / ``save_cached_state`` :703-730: numpy.savez of the
statistic's state_dict plus the call arguments; a cache is used only if the arguments match).

``shard=(rank, world)`` is new: whole batches are dealt round-robin to ranks (so every seed keeps
the batch row it has in the single-process sweep) and the raw sums are combined with ONE
all-reduce over RCCL/xGMI (``rewriting_amd.parallel``) before the cache is written.
"""
import os

import numpy
import torch

from . import pbar, runningstats
from .sampler import FixedSubsetSampler


def call_compute(compute, batch):
    if isinstance(batch, list):
        return compute(*batch)
    if isinstance(batch, dict):
        return compute(**batch)
    return compute(batch)


def make_loader(dataset, sample_size=None, batch_size=10, sampler=None, **kwargs):
    if isinstance(dataset, torch.Tensor):
        dataset = torch.utils.data.TensorDataset(dataset)
    if sampler is None and sample_size is not None:
        if sample_size > len(dataset):
            pbar.print('Warning: sample size %d > dataset size %d' % (sample_size, len(dataset)))
            sample_size = len(dataset)
        sampler = FixedSubsetSampler(list(range(sample_size)))
    return torch.utils.data.DataLoader(dataset, sampler=sampler, batch_size=batch_size, **kwargs)


def load_cached_state(cachefile, args, construct=None, shard=None):
    """The cached statistic, or None.  A cache counts only if its stored arguments match (utils/tally.py:703-718),
    if `construct(state)` can rebuild the statistic from it (a file of an unknown schema is a miss, not an error;
    the statistics of this path -- c_matrix / unit_rq / mean caches -- use the reference's schemas both ways)
    and, in a sharded sweep,
    if EVERY rank found it: the decision is collective, so no rank returns early while the others enter the
    sweep's all-reduce."""
    from .. import parallel
    result = None
    if cachefile is not None:
        try:
            dat = numpy.load(cachefile, allow_pickle=True)
            ok = True
            for a, v in args.items():
                if a not in dat or dat[a] != v:
                    pbar.print('%s %s changed from %s to %s' % (cachefile, a, dat[a] if a in dat else None, v))
                    ok = False
                    break
            if ok:
                result = construct(dat) if construct is not None else dat
        except FileNotFoundError:
            result = None                      # no cache yet: the ordinary miss, nothing to report
        except Exception as e:
            # a file that does not parse or a schema `construct` cannot rebuild is a miss, not an error -- but a
            # SILENT miss would turn a bug in set_state_dict into a full re-sweep on every run: say why
            pbar.print('%s not used as a cache (%s: %s); recomputing' % (cachefile, type(e).__name__, e))
            result = None
    if shard is not None and shard[1] > 1 and not parallel.all_agree(result is not None):
        return None
    if result is not None:
        pbar.print('Loading cached %s' % cachefile)
    return result


def save_cached_state(cachefile, obj, args, shard=None):
    """numpy.savez of the state dict plus the call arguments (utils/tally.py:721-730).  Written by rank 0 only,
    to a temporary file that is renamed into place (readers never see a partial file), and followed by a
    barrier in a sharded sweep: when any rank goes on, the cache exists."""
    from .. import parallel
    if cachefile is not None and (shard is None or shard[0] == 0):
        os.makedirs(os.path.dirname(cachefile) or '.', exist_ok=True)
        dat = obj.state_dict()
        for a, v in args.items():
            if a in dat:
                assert dat[a] == v
            dat[a] = v
        tmp = '%s.tmp.%d.npz' % (cachefile, os.getpid())
        numpy.savez(tmp, **dat)
        os.replace(tmp, cachefile)
    if shard is not None and shard[1] > 1:
        parallel.barrier()


def _sharded(loader, shard):
    if shard is None:
        yield from loader
        return
    rank, world = shard
    for i, batch in enumerate(loader):
        if i % world == rank:
            yield batch


def tally_second_moment(compute, dataset, sample_size=None, batch_size=10, cachefile=None,
                        shard=None, nchw=False, **kwargs):
    """compute(batch) -> (rows, C) samples [or (B, C, H, W) when nchw]; returns RunningSecondMoment
    (on the CPU, like the reference)."""
    args = dict(sample_size=sample_size)
    cached = load_cached_state(cachefile, args, lambda st: runningstats.RunningSecondMoment(state=st), shard)
    if cached is not None:
        return cached
    loader = make_loader(dataset, sample_size, batch_size, **kwargs)
    r2mom = runningstats.RunningSecondMoment()
    for batch in pbar(_sharded(loader, shard)):
        sample = call_compute(compute, batch)
        if nchw:
            r2mom.add_nchw(sample)
        else:
            r2mom.add(sample)
    if shard is not None and shard[1] > 1:
        from .. import parallel
        parallel.allreduce_second_moment(r2mom)
    r2mom.to_('cpu')
    save_cached_state(cachefile, r2mom, args, shard)
    return r2mom


def tally_mean(compute, dataset, sample_size=None, batch_size=10, cachefile=None, nchw=False,
               square_input=False, shard=None, **kwargs):
    args = dict(sample_size=sample_size)
    cached = load_cached_state(cachefile, args, lambda st: runningstats.RunningVariance(state=st), shard)
    if cached is not None:
        return cached
    loader = make_loader(dataset, sample_size, batch_size, **kwargs)
    rv = runningstats.RunningVariance()
    for batch in pbar(_sharded(loader, shard)):
        rv.add(call_compute(compute, batch), nchw=nchw, square_input=square_input)
    if shard is not None and shard[1] > 1:
        from .. import parallel
        parallel.allreduce_variance(rv)
    rv.to_('cpu')
    save_cached_state(cachefile, rv, args, shard)
    return rv


def _replicated():
    """tally_quantile / tally_topk_and_quantile are not sharded: under torch.distributed every rank that calls
    them computes the whole statistic for itself, takes its own cache decision and may write the (identical)
    cache -- through its own temporary file and an atomic rename, so concurrent writers cannot tear it."""
    return None


def tally_quantile(compute, dataset, sample_size=None, batch_size=10, r=4096, cachefile=None, **kwargs):
    """compute(batch) -> (samples, units); returns RunningQuantile (reference: utils/tally.py:132-154)."""
    args = dict(sample_size=sample_size, r=r)
    cached = load_cached_state(cachefile, args, lambda st: runningstats.RunningQuantile(state=st), _replicated())
    if cached is not None:
        return cached
    loader = make_loader(dataset, sample_size, batch_size, **kwargs)
    rq = runningstats.RunningQuantile(r=r)
    for batch in pbar(loader):
        rq.add(call_compute(compute, batch))
    rq.compress_()          # sorted once on the device; fresh and cached answers are the same statistic
    rq.to_('cpu')
    save_cached_state(cachefile, rq, args, _replicated())
    return rq


def tally_topk_and_quantile(compute, dataset, sample_size=None, batch_size=10, k=100, r=4096,
                            cachefile=None, **kwargs):
    """One pass computing both; compute(batch) -> (sample for top-k, sample for quantiles)
    (reference: utils/tally.py:157-181; its cached branch has two typos, quirk Q11 -- here the
    cache simply stores both state dicts under the prefixes 'rtk.' and 'rq.')."""
    args = dict(sample_size=sample_size, k=k, r=r)

    def both(st):
        pick = lambda pre: {key[len(pre):]: st[key] for key in st.files if key.startswith(pre)}
        return (runningstats.RunningTopK(state=pick('rtk.')), runningstats.RunningQuantile(state=pick('rq.')))
    cached = load_cached_state(cachefile, args, both, _replicated())
    if cached is not None:
        return cached
    loader = make_loader(dataset, sample_size, batch_size, **kwargs)
    rtk, rq = runningstats.RunningTopK(k=k), runningstats.RunningQuantile(r=r)
    for batch in pbar(loader):
        sample_tk, sample_q = call_compute(compute, batch)
        rtk.add(sample_tk)
        rq.add(sample_q)
    rtk.to_('cpu')
    rq.compress_()
    rq.to_('cpu')

    class _Both:
        def state_dict(self):
            d = {'rtk.' + a: b for a, b in rtk.state_dict().items()}
            d.update({'rq.' + a: b for a, b in rq.state_dict().items()})
            return d
    save_cached_state(cachefile, _Both(), args, _replicated())
    return rtk, rq
