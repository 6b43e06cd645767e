"""The drivers either side of the hot path (BASELINE.json configs[4]): sample-set generation
with float64 feature statistics, the Frechet distance of metrics/fid.py, and the five-variant watermark job.
CPU tests run the host logic on the kernel stand-ins of tests/hip_emulation.py; `-m gpu` tests run the kernels."""
import os
import socket
import sys

import numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import reference_shim
from tests.conftest import build_stylegan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _folded_request(n):
    from rewriting_amd import workloads
    req = workloads.load_request()
    return {k: ([[a % n, m] for a, m in v] if k == 'key' else [v[0] % n, v[1]]) for k, v in req.items()}


def check_feature_statistics_and_frechet(device):
    from rewriting_amd import samples
    rs = numpy.random.RandomState(0)
    a = (rs.randn(300, 24) * (1 + numpy.arange(24) / 8) + 0.3).astype('float32')
    b = (rs.randn(280, 24) * 1.3 - 0.2).astype('float32')
    sa, sb = samples.FeatureStatistics(), samples.FeatureStatistics()
    for i in range(0, 300, 64):
        sa.add(torch.from_numpy(a[i:i + 64]).to(device))
    sb.add(torch.from_numpy(b).to(device))
    mu, sigma = sa.mean_cov()
    # metrics/fid.py:60-61: mu = mean(act, 0), sigma = cov(act, rowvar=False)
    assert numpy.abs(mu - a.astype('float64').mean(0)).max() < 1e-6
    assert numpy.abs(sigma - numpy.cov(a.astype('float64'), rowvar=False)).max() < 1e-5
    d = samples.frechet_distance(mu, sigma, *sb.mean_cov())
    assert d > 0 and abs(samples.frechet_distance(mu, sigma, mu, sigma)) < 1e-6
    return (mu, sigma) + sb.mean_cov() + (d,)


def test_feature_statistics_and_frechet_cpu():
    check_feature_statistics_and_frechet('cpu')


@pytest.mark.skipif(not reference_shim.available(), reason='metrics/fid.py lives in /root/reference')
def test_frechet_distance_equals_the_reference_function():
    """metrics/fid.py:137-187 (its module imports tensorflow at the top, so the one function is extracted from the
    file like the shim extracts upfirdn2d_native) on the same statistics, incl. the near-singular retry branch."""
    from scipy import linalg
    from rewriting_amd import samples
    ns = {'np': numpy, 'linalg': linalg, 'warnings': __import__('warnings')}
    ref = reference_shim._extract_function(os.path.join(reference_shim.REFERENCE_ROOT, 'metrics', 'fid.py'),
                                           'calculate_frechet_distance', ns)
    mu1, s1, mu2, s2, d = check_feature_statistics_and_frechet('cpu')
    assert abs(d - ref(mu1, s1, mu2, s2)) < 1e-9 * max(1.0, abs(d))
    z = numpy.zeros_like(s1)                                    # singular product -> eps on the diagonal, both sides
    with pytest.warns(Warning):
        d0 = samples.frechet_distance(mu1, z, mu2, z)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert abs(d0 - ref(mu1, z, mu2, z)) < 1e-9


def check_sample_set_and_watermark_variants(device, size=64, nseeds=20, niters=11):
    """sample_set_statistics (launches of k x 10 seeds) equals the reference's batch-of-10 loop
    (metrics/make_watermark_images.py:99-131); the three erase methods run and change what they should."""
    from rewriting_amd import samples, workloads
    from rewriting_amd.utils import zdataset
    model = build_stylegan(size, 0.5, device=device)
    zds = zdataset.z_dataset_for_model(model, size=nseeds + 3)            # ragged last batch
    got = workloads.sample_set_statistics(model, zds, batch=20)
    want = samples.FeatureStatistics()
    with torch.no_grad():
        for i in range(0, len(zds), 10):
            z = torch.stack([zds[j][0] for j in range(i, min(i + 10, len(zds)))]).to(device)
            want.add(workloads.pooled_rgb_features(model(z)))
    assert got.count == want.count == nseeds + 3
    for a, b in zip(got.mean_cov(), want.mean_cov()):
        assert numpy.abs(a - b).max() < 1e-5 * max(1.0, numpy.abs(b).max())
    req = _folded_request(nseeds)
    out = {}
    for v in (dict(erasemethod='none'), dict(erasemethod='ours', nreps=1, drank=30, rank=1),
              dict(erasemethod='gandissect', drank=30)):
        t, stats, gw = workloads.run_watermark_variant(v, device, req, size=size, layer=6, sample_size=nseeds,
                                                       niters=niters)
        assert stats.count == nseeds and t['edit_s'] >= 0 and t['sample_set_s'] > 0
        out[v['erasemethod']] = (stats, gw.target_weights().detach().clone())
    W0 = out['none'][1]
    assert (out['ours'][1] - W0).abs().max().item() > 0 and (out['gandissect'][1] - W0).abs().max().item() > 0
    mu0, s0 = out['none'][0].mean_cov()
    for name in ('ours', 'gandissect'):
        mu, s = out[name][0].mean_cov()
        assert samples.frechet_distance(mu, s, mu0, s0) > 0
    # the un-edited variant's statistics are those of the plain generator
    plain = workloads.sample_set_statistics(build_stylegan(size, 0.5, device=device),
                                            zdataset.z_dataset_for_model(model, size=nseeds))
    assert numpy.abs(plain.mean_cov()[0] - mu0).max() < 1e-5 * max(1.0, numpy.abs(mu0).max())
    return out


def test_sample_set_and_watermark_variants_cpu(emulated_hip):
    check_sample_set_and_watermark_variants('cpu')


def _replica_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import pytest as _pytest
    from tests import hip_emulation
    hip_emulation.install(_pytest.MonkeyPatch())
    from rewriting_amd import parallel, workloads
    parallel.init_from_env(backend='gloo')
    n = 20
    req = _folded_request(n)
    variants = [dict(erasemethod='none'), dict(erasemethod='gandissect', drank=10), dict(erasemethod='none')]
    mine = {}
    with parallel.replicas():
        assert parallel.shard() is None                      # sweeps stay local: no collective inside a replica
        for i, v in enumerate(variants):
            if i % world == rank:
                t, stats, _ = workloads.run_watermark_variant(v, 'cpu', req, size=32, layer=6, sample_size=n, niters=3)
                mine['%d-%s' % (i, v['erasemethod'])] = stats.mean_cov()[0]
    assert parallel.shard() == (rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        torch.save({k: v for part in gathered for k, v in part.items()}, os.path.join(out, 'all.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_watermark_variants_run_as_replicas_on_two_ranks(tmp_path):
    """world_size 2 (gloo): variants dealt round-robin, every rank runs complete, independent rewriters (their
    statistics sweeps must not enter a collective), results gathered at the end."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_replica_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(str(tmp_path / 'all.pt'), weights_only=False)
    assert sorted(got) == ['0-none', '1-gandissect', '2-none']
    assert numpy.abs(got['0-none'] - got['2-none']).max() < 1e-12          # rank 0 twice: deterministic
    assert numpy.abs(got['0-none'] - got['1-gandissect']).max() > 0


# ------------------------------------------------------------------ on the MI355X
@pytest.mark.gpu
def test_feature_statistics_and_frechet_gpu():
    cpu = check_feature_statistics_and_frechet('cpu')
    gpu = check_feature_statistics_and_frechet('cuda')               # fp32 MFMA outer products, fp64 across batches
    for a, b in zip(cpu[:4], gpu[:4]):
        assert numpy.abs(a - b).max() < 1e-5
    assert abs(cpu[4] - gpu[4]) < 1e-4 * max(1.0, abs(cpu[4]))


@pytest.mark.gpu
def test_sample_set_generation_and_watermark_variants_gpu():
    from rewriting_amd import samples
    check_sample_set_and_watermark_variants('cuda')
    # metrics/sample.py:32-37: image n from its own z stream at batch 1 (noise row 0) -- a batched launch must give
    # the same images
    model = build_stylegan(64, 0.5, device='cuda')
    seeds = list(range(7))
    batched = torch.cat([img for _, img in samples.generate(model, model, seeds, batch=4)])
    with torch.no_grad():
        single = torch.cat([model(samples.seed_latents(model, [s]).cuda()) for s in seeds])
    assert (batched - single).abs().max().item() < 1e-5 * max(1.0, single.abs().max().item())
