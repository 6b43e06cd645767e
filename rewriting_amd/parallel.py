"""Multi-GPU pieces of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed code on this path; what shards
naturally is (section 8e):

* the key-statistics sweep -- whole batches of 10 consecutive seeds are dealt round-robin to
  ranks (every seed keeps its reference noise row), each rank accumulates its own C x C sums,
  and ONE sum all-reduce of (mom2, count) makes every rank hold the same ``C``.  The message
  is 2 MiB at C=512 (float64): latency-bound on xGMI, so a plain ``all_reduce`` is the right collective.
  The reduction is done in float64 so the result does not depend on the ring order;
* per-seed generator passes (sample sets): seed i -> rank i mod world, no collective;
* the solve does not shard (2001 sequential steps on 9.4 MB of state): replicas only.

Backend "nccl" IS RCCL on ROCm; CPU tests use "gloo" with world_size 2.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* if a launcher set them.
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


_independent = [0]


class replicas:
    """Context manager: inside it the ranks run INDEPENDENT jobs (different edits / watermark variants on
    different GPUs -- the part of the path that does not shard), so ``shard()`` is None and every
    sweep a rewriter runs is local to its rank: no collective is entered."""

    def __enter__(self):
        _independent[0] += 1
        return self

    def __exit__(self, *exc):
        _independent[0] -= 1


def shard():
    """(rank, world) for ``tally.tally_second_moment(shard=...)``; None when single-process or inside
    ``replicas()``."""
    if _independent[0] == 0 and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    return None


def collective_device():
    """Where tensors handed to a collective have to live: this rank's GPU under nccl (= RCCL), the host under gloo."""
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


_coll_device = collective_device


def all_agree(flag):
    """True only if `flag` is true on EVERY rank (one tiny MIN all-reduce); the local value when single-process.
    Used for decisions that must be collective -- e.g. "everybody found the statistics cache" -- so that no rank
    enters a sweep's all-reduce while another returns early."""
    if shard() is None:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], device=_coll_device())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def barrier():
    if shard() is not None:
        dist.barrier()


def balanced_batch(n, cap, world):
    """Seeds per launch of a sweep over n seeds that is dealt round-robin to `world` ranks: a multiple of 10 (the
    reference's batch: every seed keeps its noise row), at most `cap`, and such that the number of launches is a
    multiple of the number of ranks wherever n allows it -- 10 000 seeds at 510 per launch are 20 launches, i.e. three
    for four of eight ranks and two for the others; 24 launches of 420 are three each -- and never fewer launches than
    ranks."""
    cap = max(10, cap // 10 * 10)
    if world <= 1:
        return max(10, min(cap, n // 10 * 10))       # one rank: the cap itself (it is chosen for whole rounds of the chip)
    launches = -(-n // cap)
    launches = -(-launches // world) * world
    b = min(cap, max(10, -(-n // launches) + 9) // 10 * 10)
    b = max(10, b)
    while b > 10 and -(-n // b) < min(world, max(1, n // 10)):
        b -= 10
    return b


def batches_for_rank(n_batches, rank, world):
    return list(range(rank, n_batches, world))


def allreduce_second_moment(r2mom):
    """In place: every rank ends with the global (mom2, count).  A rank whose shard was empty (fewer batches
    than ranks) contributes zeros: the channel count is agreed first, so nobody blocks while another raises."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return r2mom
    dev = _coll_device()
    have = r2mom.mom2 is not None
    shape = torch.tensor([float(r2mom.mom2.shape[0]) if have else 0.0], device=dev)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX)
    c = int(shape.item())
    if c == 0:
        raise RuntimeError('second-moment sweep: no rank received a batch')
    if have:
        home_dev, home_dtype = r2mom.mom2.device, r2mom.mom2.dtype
        local = r2mom.mom2.to(dev, torch.float64).reshape(-1)
    else:
        home_dev, home_dtype = dev, torch.float32
        local = torch.zeros(c * c, dtype=torch.float64, device=dev)
    packed = torch.cat([local, torch.tensor([float(r2mom.count)], dtype=torch.float64, device=dev)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    r2mom.mom2 = packed[:-1].reshape(c, c).to(home_dtype).to(home_dev)
    r2mom.count = int(round(packed[-1].item()))
    return r2mom


def allreduce_variance(rv):
    """In place: pooled (count, mean, centred second moment) of a RunningVariance over all ranks
    (exact pooling: M2 = sum M2_r + sum n_r (mean_r - mean)^2), reduced in float64.  Empty shards contribute
    zeros (see allreduce_second_moment)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rv
    dev = _coll_device()
    have = rv._mean is not None
    shape = torch.tensor([float(rv._mean.numel()) if have else 0.0], device=dev)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX)
    c = int(shape.item())
    if c == 0:
        raise RuntimeError('mean/variance sweep: no rank received a batch')
    if have:
        home_dev, home_dtype = rv._mean.device, rv._mean.dtype
        n = float(rv.count)
        mean = rv._mean.to(dev, torch.float64)
        local = torch.cat([mean * n, rv.v_cmom2.to(dev, torch.float64) + n * mean * mean,
                           torch.tensor([n, float(rv.batchcount)], dtype=torch.float64, device=dev)])
    else:
        home_dev, home_dtype = dev, torch.float32
        local = torch.zeros(2 * c + 2, dtype=torch.float64, device=dev)
    dist.all_reduce(local, op=dist.ReduceOp.SUM)
    packed = local
    total = packed[-2].item()
    gmean = packed[:c] / total
    m2 = packed[c:2 * c] - total * gmean * gmean
    rv.count = int(round(total))
    rv.batchcount = int(round(packed[-1].item()))
    rv._mean = gmean.to(home_dtype).to(home_dev)
    rv.v_cmom2 = m2.clamp_(min=0).to(home_dtype).to(home_dev)
    return rv


def gather_images(local_images, seeds_total):
    """all_gather of per-rank image batches produced with seed i -> rank i mod world (this rank holds seeds
    rank, rank + world, ... in that order); returns all seeds_total images in seed order on every rank (optional:
    ranks normally write their own files).  The collective wants equal shapes: ranks are padded to
    ceil(seeds_total / world) images -- when the division leaves a remainder r = seeds_total % world, the LAST
    world - r ranks hold one image less.  A rank whose batch has the wrong length does not raise on its own (the others
    would wait in the collective for ever): the verdict is agreed on by all ranks first, then every rank raises."""
    if shard() is None:
        if local_images.shape[0] != seeds_total:
            raise ValueError('one process holds all %d images, got %d' % (seeds_total, local_images.shape[0]))
        return local_images
    rank, world = shard()
    mine = len(range(rank, seeds_total, world))
    bad = torch.tensor([0 if local_images.shape[0] == mine else 1], dtype=torch.int32, device=collective_device())
    dist.all_reduce(bad, op=dist.ReduceOp.SUM)
    if int(bad.item()):
        raise ValueError('gather_images: %d rank(s) hold the wrong number of images (rank %d of %d: %d, expected %d of %d)'
                         % (int(bad.item()), rank, world, local_images.shape[0], mine, seeds_total))
    per_rank = -(-seeds_total // world)
    padded = local_images.contiguous()
    if mine < per_rank:
        padded = torch.cat([padded, padded.new_zeros((per_rank - mine,) + tuple(padded.shape[1:]))])
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    out = torch.empty((seeds_total,) + tuple(local_images.shape[1:]), dtype=local_images.dtype,
                      device=local_images.device)
    for r, part in enumerate(parts):
        idx = torch.arange(r, seeds_total, world, device=out.device)
        out[idx] = part[:idx.numel()]
    return out
