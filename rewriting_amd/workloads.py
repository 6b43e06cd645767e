"""BASELINE.json configs[4]: the watermark-removal experiment of the reference (``watermark.sh:11-24`` runs
``metrics/make_watermark_images.py`` five times) as ONE job over the GPUs of a node.

What shards: the five variants are independent edits of independent copies of the generator -- the part of the
path SURVEY.md 8e calls "replicas only" (the 2001-step solve is sequential on 9.4 MB of state).  Variants are
dealt round-robin to the ranks inside ``parallel.replicas()``, so every sweep a rewriter runs stays local to its
rank and no collective is entered; the per-variant feature statistics of the generated sample sets are gathered
once at the end (outside the timed region) for the Frechet distances between variants.

Per variant (metrics/make_watermark_images.py:39-84, defaults :13-28): church-256 architecture, layer 6,
1000-seed statistics computed with the truncation-1.0 model and re-used from the shared cache directory by the
truncation-0.5 rewriter (quirk Q7), request multikey_markandbottom.json, then
  'ours'        nreps x apply_erase(rank, drank, 2001 steps, low_rank_gradient=True)
  'gandissect'  multi_key_from_selection(key_method='gandissect', rank=drank) + zero()
  'none'        no edit
and the sample set: the images of the statistics seeds in batches of 10 (:99-131 -- each image with the noise row
of its position in its batch of 10, quirk Q1), reduced on the fly to float64 feature statistics.
The reference's feature extractor (a TensorFlow Inception graph downloaded at run time, metrics/fid.py:16) is
out of scope and pluggable; the default here is a fixed 8x8 average-pooled RGB descriptor (192 features).
"""
import copy
import json
import os
import shutil
import tempfile
import time

import torch

from . import parallel, samples, synthetic
from .utils import zdataset
from .utils.stylegan2 import models
from .utils.stylegan2.models import noise_batch_period

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WATERMARK_VARIANTS = [                  # watermark.sh:11-24, in order
    dict(erasemethod='ours', nreps=2, drank=60, rank=1),
    dict(erasemethod='ours', nreps=2, drank=30, rank=1),
    dict(erasemethod='gandissect', drank=30),
    dict(erasemethod='gandissect', drank=60),
    dict(erasemethod='none'),
]


def variant_name(v):
    return '-'.join(str(v[k]) for k in ('erasemethod', 'drank', 'nreps') if k in v)


def pooled_rgb_features(images):
    """(B,3,S,S) -> (B,192): 8x8 average pooling.  A stand-in for the Inception pool3 features."""
    return torch.nn.functional.adaptive_avg_pool2d(images, 8).reshape(images.shape[0], -1)


_templates = {}


def build_model(size, truncation, device, seed=0):
    """The "checkpoint" of this offline build: seeded synthetic weights.  The module tree is built and its weights are
    drawn ONCE per (size, seed, device) and kept as a template ON THE DEVICE; every variant then takes a copy -- a
    device-to-device copy of 120 MB where constructing a generator draws 30 M random initial weights on the host and a
    host-resident checkpoint pays the PCIe transfer again (0.25 - 0.30 s of the 0.4 s a watermark variant spent before
    its first kernel: profiles/r04j_bench.json, build_models_s)."""
    key = (size, seed, str(device))
    if key not in _templates:
        g = models.SeqStyleGAN2(size, 512, 8, truncation=truncation, mconv='seq')
        synthetic.randomize_(g, seed=seed)
        _templates[key] = g.eval().to(device)
    g = copy.deepcopy(_templates[key])
    g.latents.truncation = truncation
    return g


def sample_set_statistics(model, zds, batch=250, feature_fn=pooled_rgb_features, stats=None):
    """metrics/make_watermark_images.py:99-131 without the PNG writer: model(z) over the dataset in index order,
    launches of `batch` seeds = batch/10 reference batches of 10 (noise_batch_period keeps every image's row)."""
    stats = stats or samples.FeatureStatistics()
    device = next(model.parameters()).device
    n = len(zds)
    batch = max(10, batch // 10 * 10)
    with torch.no_grad(), noise_batch_period(10):
        for i in range(0, n - n % 10, batch):
            z = torch.stack([zds[j][0] for j in range(i, min(i + batch, n - n % 10))]).to(device)
            stats.add(feature_fn(model(z)))
    if n % 10:
        with torch.no_grad():           # the reference's last, short batch
            z = torch.stack([zds[j][0] for j in range(n - n % 10, n)]).to(device)
            stats.add(feature_fn(model(z)))
    return stats


def run_watermark_variant(variant, device, request, size=256, layer=6, sample_size=1000, niters=2001, piters=10,
                          lr=0.05, cachedir=None, feature_fn=pooled_rgb_features, weight_seed=0, callback='loss_only'):
    """One invocation of metrics/make_watermark_images.py main(); returns (timings, FeatureStatistics, rewriter).
    callback: the driver passes a progress-bar hook as update_callback on every iteration (:66-72, `pbar_hook(it)`):
    'loss_only' = that hook, marked ganrewrite.loss_only (it reads `it` only: the solve runs uninterrupted),
    'reference' = the same hook unmarked (the reference's contract: called between the steps, one launch per iteration),
    'none' = no callback."""
    from .rewrite import ganrewrite
    own_cache = cachedir is None
    cachedir = cachedir or tempfile.mkdtemp(prefix='rw_watermark_')
    sync = (lambda: torch.cuda.synchronize(device)) if torch.device(device).type == 'cuda' else (lambda: None)
    t = {}
    t0 = time.perf_counter()
    model_for_covariance = build_model(size, 1.00, device, weight_seed)
    model = build_model(size, 0.50, device, weight_seed)
    zds = zdataset.z_dataset_for_model(model, size=sample_size)
    sync()
    t['build_models_s'] = time.perf_counter() - t0
    gw = None
    for m in (model_for_covariance, model):
        ta = time.perf_counter()
        gw = ganrewrite.SeqStyleGanRewriter(
            m, zds, layer, cachedir=cachedir, low_rank_insert=True, low_rank_gradient=True,
            key_method={'ours': 'zca', 'gandissect': 'gandissect', 'none': 'zca'}[variant['erasemethod']],
            tight_paste=True)
        if m is model_for_covariance:
            gw.collect_2nd_moment()
        sync()
        # the first rewriter sweeps (or finds the sweep in the cache directory the job's variants share, as the five
        # invocations of watermark.sh share their results directory) and factorises; the second loads and factorises
        t['rewriter_covariance_model_s' if m is model_for_covariance else 'rewriter_edited_model_s'] = time.perf_counter() - ta
    t['statistics_s'] = time.perf_counter() - t0
    t1 = time.perf_counter()
    if variant['erasemethod'] == 'ours':
        ticks = [0]

        def pbar_cb(it, loss):
            ticks[0] += 1
        cb = {'none': None, 'reference': pbar_cb, 'loss_only': ganrewrite.loss_only(lambda it, loss: pbar_cb(it, loss))}[callback]
        for _ in range(variant.get('nreps', 2)):
            gw.apply_erase(request, rank=variant.get('rank', 1), drank=variant['drank'], niter=niters, piter=piters,
                           lr=lr, update_callback=cb)
        t['callback'] = callback
    elif variant['erasemethod'] == 'gandissect':
        mkey = gw.multi_key_from_selection(request['key'], rank=variant['drank'])
        gw.zero(mkey)
    else:
        assert variant['erasemethod'] == 'none'
    torch.cuda.synchronize(device) if torch.device(device).type == 'cuda' else None
    t['edit_s'] = time.perf_counter() - t1
    t2 = time.perf_counter()
    stats = sample_set_statistics(gw.model, zds, feature_fn=feature_fn)
    torch.cuda.synchronize(device) if torch.device(device).type == 'cuda' else None
    t['sample_set_s'] = time.perf_counter() - t2
    t['images'] = stats.count
    if own_cache:
        shutil.rmtree(cachedir, ignore_errors=True)
    return t, stats, gw


def load_request(name='multikey_markandbottom.json'):
    with open(os.path.join(ROOT, 'tests', 'golden', 'masks', name)) as f:
        return json.load(f)


def fold_request(request, nseeds):
    """The request with its seed indices folded into [0, nseeds): lets a reduced sample (smoke runs, launcher
    tests) serve the recorded masks; the identity at the reference's sample size."""
    out = dict(request)                 # anything else a request carries (a name, a rank, ...) passes through
    if 'key' in request:
        out['key'] = [[n % nseeds, m] for n, m in request['key']]
    for k in ('object', 'paste'):
        if k in request:
            out[k] = [request[k][0] % nseeds, request[k][1]]
    return out


def watermark_job(device, rank=0, world=1, sample_size=1000, niters=2001, size=256, layer=6, variants=None,
                  callback='loss_only', share_cache=True):
    """The five variants, variant i on rank i mod world.  Returns {variant name: (timings, stats)} of this rank.
    share_cache: the variants of a rank share one statistics cache directory, as the five invocations of watermark.sh
    share their results directory (utils/tally.py:703-730: the first computes the 1000-seed sweep, the others load
    it) -- one sweep per rank instead of one per variant."""
    request = fold_request(load_request(), sample_size)
    variants = WATERMARK_VARIANTS if variants is None else variants
    out = {}
    cachedir = tempfile.mkdtemp(prefix='rw_watermark_job_') if share_cache else None
    try:
        with parallel.replicas():
            for i, v in enumerate(variants):
                if i % world != rank:
                    continue
                t, stats, _ = run_watermark_variant(v, device, request, size=size, layer=layer, sample_size=sample_size,
                                                    niters=niters, cachedir=cachedir, callback=callback)
                out[variant_name(v)] = (t, stats)
    finally:
        if cachedir is not None:
            shutil.rmtree(cachedir, ignore_errors=True)
    return out


def watermark_bench(args, rank, world, device, timed):
    """bench.py --workload watermark: one step = the whole five-variant job (strong scaling: the work is fixed)."""
    import torch.distributed as dist
    results = {}

    def step():
        results.clear()
        results.update(watermark_job(device, rank, world, sample_size=args.seeds,
                                     niters=getattr(args, 'niters', 2001), size=getattr(args, 'wm_size', 256)))
    dt = timed(step, args.steps, args.warmup, world)
    mine = {name: dict(t, mu_sigma=[a.tolist() for a in stats.mean_cov()]) for name, (t, stats) in results.items()}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        mine = {k: v for part in gathered for k, v in part.items()}
    frechet = {}
    if rank == 0 and 'none' in mine:
        import numpy
        mu0, s0 = (numpy.array(a) for a in mine['none']['mu_sigma'])
        for name, rec in mine.items():
            mu, s = (numpy.array(a) for a in rec['mu_sigma'])
            frechet[name] = float(samples.frechet_distance(mu, s, mu0, s0))
    per_variant = {name: {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items() if k != 'mu_sigma'}
                   for name, rec in mine.items()}
    n = args.steps
    images = sum(rec['images'] for rec in per_variant.values())
    return dict(metric='watermark-removal job (watermark.sh: 5 erase variants + their %d-image sample sets), seconds'
                       % args.seeds,
                value=round(dt / n, 4), unit='s', n_gpus=world, steps=n, warmup=args.warmup,
                ms_per_step=round(dt / n * 1e3, 2), higher_is_better=False, scaling='strong', vs_baseline=None,
                dtype='f32', data='synthetic',
                config=dict(workload='church-%d architecture, layer 6, %d-seed statistics, %d-step erase solves '
                                     '(low_rank_gradient), variants dealt round-robin to ranks as independent replicas'
                                     % (getattr(args, 'wm_size', 256), args.seeds, getattr(args, 'niters', 2001)),
                            variants=per_variant, images_per_s=round(images / (dt / n), 1),
                            frechet_vs_unedited_pooled_rgb=frechet))
