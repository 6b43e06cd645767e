"""world_size-2 gloo test of the sharded key-statistics sweep: batches dealt round-robin, one
all-reduce of (mom2, count); every rank must end with the single-process statistic."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from rewriting_amd import parallel
    from rewriting_amd.utils import tally
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and parallel.shard() == (rank, world)
    torch.manual_seed(0)
    data = torch.randn(57, 6)
    seen = []

    def compute(batch):
        seen.append(int(batch.shape[0]))
        return batch * 2.0
    stat = tally.tally_second_moment(compute, data, shard=parallel.shard())
    rv = tally.tally_mean(lambda b: b * 2.0, data, shard=parallel.shard())
    torch.save(dict(mom2=stat.mom2, count=stat.count, seen=seen, mean=rv.mean(), var=rv.variance(),
                    n=rv.size()), os.path.join(out, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_second_moment_matches_single_process(tmp_path):
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    torch.manual_seed(0)
    data = torch.randn(57, 6) * 2.0
    want = data.t() @ data
    got = [torch.load(str(tmp_path / ('r%d.pt' % r))) for r in range(2)]
    assert got[0]['seen'] == [10, 10, 10] and got[1]['seen'] == [10, 10, 7]     # batches 0,2,4 | 1,3,5
    for g in got:
        assert g['count'] == 57
        assert torch.allclose(g['mom2'], want, rtol=1e-5, atol=1e-4)
    assert torch.equal(got[0]['mom2'], got[1]['mom2'])
    for g in got:       # sharded tally_mean: pooled mean / variance on every rank
        assert g['n'] == 57
        assert torch.allclose(g['mean'], data.mean(0), atol=1e-5)
        # (variance is not compared with the true one: the per-rank merge is bug-compatible with the
        #  reference's cross term, utils/runningstats.py:786-788; only mean() is consumed on the path)
        assert torch.equal(got[0]['var'], got[1]['var'])


def test_batches_for_rank_partition():
    from rewriting_amd import parallel
    all_b = sorted(sum((parallel.batches_for_rank(13, r, 4) for r in range(4)), []))
    assert all_b == list(range(13))


def _uneven_worker(rank, world, port, out):
    """seeds_total % world != 0: gather_images pads the short rank; the sharded sample-set statistics (samples.generate
    with shard=, FeatureStatistics.allreduce_) count every seed once."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from rewriting_amd import parallel, samples
    parallel.init_from_env(backend='gloo')
    total = 7
    mine = list(range(rank, total, world))
    local = torch.stack([torch.full((3, 2, 2), float(s)) for s in mine])
    everything = parallel.gather_images(local, total)
    ok_order = all(torch.equal(everything[s], torch.full((3, 2, 2), float(s))) for s in range(total))
    try:
        parallel.gather_images(local[:-1] if rank == 0 else local, total)     # ONE rank is wrong: every rank raises, none hangs
        refused = False
    except ValueError:
        refused = True

    torch.manual_seed(3)
    net = torch.nn.Linear(4, 5)                          # "generator": z (4) -> a feature row per seed
    stats = samples.FeatureStatistics()
    seeds = list(range(11))                              # 11 seeds, batches of 3, two ranks: ragged everywhere
    seen = []
    for chunk, img in samples.generate(net, net, seeds, batch=3, device='cpu', shard=parallel.shard()):
        seen += list(chunk)
        stats.add(img)
    stats.allreduce_()
    mu, sigma = stats.mean_cov()
    torch.save(dict(ok_order=ok_order, refused=refused, seen=seen, count=stats.count, mu=mu, sigma=sigma),
               os.path.join(out, 'u%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_shards_gather_and_sample_statistics(tmp_path):
    import socket
    import numpy
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_uneven_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(str(tmp_path / ('u%d.pt' % r)), weights_only=False) for r in range(2)]
    assert all(g['ok_order'] and g['refused'] for g in got)
    assert sorted(got[0]['seen'] + got[1]['seen']) == list(range(11))       # every seed on exactly one rank
    sys.path.insert(0, ROOT)
    from rewriting_amd import samples

    torch.manual_seed(3)
    net = torch.nn.Linear(4, 5)
    want = samples.FeatureStatistics()
    for _, img in samples.generate(net, net, list(range(11)), batch=3, device='cpu'):
        want.add(img)
    mu, sigma = want.mean_cov()
    for g in got:
        assert g['count'] == 11
        assert numpy.allclose(g['mu'], mu, atol=1e-9) and numpy.allclose(g['sigma'], sigma, atol=1e-9)
