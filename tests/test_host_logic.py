"""CPU tests of the host logic that needs no kernels: nethook slicing/hooking, seed contract,
tally loader + npz cache protocol, mask decoding, bounding-box / paste helpers, DataBag, the
state-dict surface, and BASELINE config 1 (ProgGAN rewriter on PyTorch-CPU) end to end against
the reference-generated golden."""
import json
import os
from collections import OrderedDict

import numpy
import pytest
import torch
from torch import nn

from rewriting_amd.rewrite import ganrewrite
from rewriting_amd.utils import nethook, proggan, renormalize, runningstats, tally, zdataset
from rewriting_amd.utils.stylegan2 import models
from rewriting_amd import synthetic
from tests.conftest import golden_meta, load_golden, load_mask_request, subsample, ROOT


def _nest():
    def block(n):
        return nn.Sequential(OrderedDict([('a', nn.Linear(n, n)), ('b', nn.ReLU()), ('c', nn.Linear(n, n))]))
    return nn.Sequential(OrderedDict([('l1', block(4)), ('l2', block(4)), ('l3', block(4))]))


def test_subsequence_inclusive_exclusive_and_dotted_names():
    net = _nest()
    names = lambda m: [n for n, mod in m.named_modules() if n and not list(mod.children())]
    assert names(nethook.subsequence(net, first_layer='l2', last_layer='l2')) == ['l2.a', 'l2.b', 'l2.c']
    assert names(nethook.subsequence(net, upto_layer='l2.b')) == ['l1.a', 'l1.b', 'l1.c', 'l2.a']
    assert names(nethook.subsequence(net, first_layer='l2.b', last_layer='l2.c')) == ['l2.b', 'l2.c']
    assert names(nethook.subsequence(net, after_layer='l2.b')) == ['l2.c', 'l3.a', 'l3.b', 'l3.c']
    assert names(nethook.subsequence(net, after_layer='l1', upto_layer='l3')) == ['l2.a', 'l2.b', 'l2.c']
    assert names(nethook.subsequence(net, single_layer='l3.a')) == ['l3.a']
    shared = nethook.subsequence(net, first_layer='l2.a', share_weights=True)
    assert shared.l2.a.weight is net.l2.a.weight
    copied = nethook.subsequence(net, first_layer='l2.a')
    assert copied.l2.a.weight is not net.l2.a.weight and torch.equal(copied.l2.a.weight, net.l2.a.weight)
    with pytest.raises(ValueError):
        nethook.subsequence(net, first_layer='nope')
    # the three rewriter parts compose to the whole
    x = torch.randn(3, 4)
    parts = [nethook.subsequence(net, upto_layer='l2.b', share_weights=True),
             nethook.subsequence(net, first_layer='l2.b', last_layer='l2.c', share_weights=True),
             nethook.subsequence(net, after_layer='l2.c', share_weights=True)]
    assert torch.allclose(parts[2](parts[1](parts[0](x))), net(x))


def test_instrumented_model_retain_edit_and_restore():
    net = _nest()
    x = torch.randn(2, 4)
    base = net(x)
    with nethook.InstrumentedModel(net) as inst:
        inst.retain_layer('l2.b')
        inst.edit_layer('l2.b', ablation=1.0, replacement=torch.zeros(4))
        out = inst(x)
        kept = inst.retained_layer('l2.b')
        assert 'forward' in net.l2.b.__dict__
        assert kept.shape == (2, 4) and not torch.allclose(out, base)
        sliced = inst(x, first_layer='l1', last_layer='l1')
        assert torch.allclose(sliced, net.l1(x))
        with pytest.raises(ValueError):
            inst.retain_layer('missing')
    assert 'forward' not in net.l2.b.__dict__ and 'forward' not in net.__dict__
    assert torch.allclose(net(x), base)
    nethook.set_requires_grad(False, net)
    assert not any(p.requires_grad for p in net.parameters())


def test_seed_contract_and_noise_stream():
    z = zdataset.standard_z_sample(3, 512, seed=1)
    assert [round(v, 4) for v in z[0, :3].tolist()] == [1.6243, -0.6118, -0.5282]
    assert torch.equal(zdataset.standard_z_sample(7, 512, seed=1)[:3], z)
    g = proggan.ProgressiveGenerator(resolution=8)
    assert tuple(zdataset.z_sample_for_model(g, 4).shape) == (4, 512, 1, 1)
    sg = models.SeqStyleGAN2(8, 512, 2, mconv='seq')
    assert tuple(zdataset.z_dataset_for_model(sg, size=5)[2][0].shape) == (512,)
    ds = zdataset.z_dataset_for_model(sg, indices=[4, 1])
    assert torch.equal(ds[0][0], zdataset.standard_z_sample(5, 512)[4])
    n = models.reference_noise(3, 16, torch.device('cpu'))
    assert numpy.array_equal(n.numpy(), numpy.random.RandomState(0).randn(3, 16).astype('float32'))
    assert numpy.array_equal(models.reference_noise(1, 40, torch.device('cpu')).numpy().ravel(),
                             numpy.random.RandomState(0).randn(1, 40).astype('float32').ravel())


def test_tally_batches_cache_and_sharding(tmp_path):
    data = torch.arange(35, dtype=torch.float32)[:, None] * torch.ones(1, 4)
    seen = []

    def compute(batch):
        seen.append(batch[:, 0].tolist())
        return batch
    cache = str(tmp_path / 'sub' / 'r2m.npz')
    r = tally.tally_second_moment(compute, data, cachefile=cache)
    assert [len(s) for s in seen] == [10, 10, 10, 5] and seen[1][0] == 10.0       # index order, batches of 10
    assert r.count == 35 and torch.allclose(r.moment(), (data.t() @ data) / 35)
    dat = numpy.load(cache, allow_pickle=True)
    assert sorted(dat.files) == ['constructor', 'count', 'mom2', 'sample_size']    # reference schema
    assert 'RunningSecondMoment' in str(dat['constructor'])
    seen.clear()
    r2 = tally.tally_second_moment(compute, data, cachefile=cache)
    assert not seen and torch.equal(r2.mom2, r.mom2)                                # served from cache
    tally.tally_second_moment(compute, data, sample_size=20, cachefile=cache)       # args changed -> recompute
    assert [len(s) for s in seen] == [10, 10]
    # round-robin batch sharding: rank r takes batches r, r+world, ...
    seen.clear()
    a = tally.tally_second_moment(compute, data, shard=(1, 2))
    assert [s[0] for s in seen] == [10.0, 30.0] and a.count == 15
    rv = tally.tally_mean(lambda b: b, data, cachefile=str(tmp_path / 'unit_rs.npz'))
    assert torch.allclose(rv.mean(), data.mean(0)) and rv.batchcount == 4
    rv2 = runningstats.RunningVariance(state=str(tmp_path / 'unit_rs.npz'))
    assert torch.equal(rv2.mean(), rv.mean())


def test_mask_decode_and_paste_helpers():
    req = load_mask_request('recorded_horse_hat.json')
    area = renormalize.from_url(req['object'][1], target='pt', size=(32, 32))[0]
    assert tuple(area.shape) == (32, 32) and 0 <= area.min() and area.max() <= 1
    g = load_golden('rw_s64_l8_horsehat')
    assert list(ganrewrite.positive_bounding_box(area)) == list(g['obj_bounds'])
    assert not ganrewrite.ProgressiveGanRewriter.is_empty_mask(None, req['object'][1])
    m = torch.zeros(6, 8)
    assert ganrewrite.positive_bounding_box(m) == (0, 0, 0, 0)
    m[2:4, 3:7] = 1
    assert ganrewrite.positive_bounding_box(m) == (2, 3, 4, 7)
    assert ganrewrite.centered_location(m) == (3, 5)
    src = torch.zeros(1, 2, 6, 8)
    clip = torch.ones(1, 2, 3, 4)
    out, b = ganrewrite.paste_clip_at_center(src, clip, (0, 7))
    assert b == (0, 4, 3, 8) and out[:, :, 0:3, 4:8].eq(1).all() and out.sum() == 24   # clamped inside
    half = torch.full((3, 4), 0.5)
    out, _ = ganrewrite.paste_clip_at_center(src + 2, clip, (3, 4), half)
    assert torch.allclose(out[:, :, 2:5, 2:6], torch.full((1, 2, 3, 4), 1.5))
    k, v = torch.randn(1, 2, 8, 8), torch.randn(1, 2, 16, 16)
    ck, cv, sb, tb = ganrewrite.crop_clip_to_bounds(k, v, (3, 5, 8, 9))
    assert sb == (1, 2, 4, 5) and tb == (2, 4, 8, 10) and ck.shape[2:] == (3, 3) and cv.shape[2:] == (6, 6)


def test_databag_and_state_dict_surface():
    d = models.DataBag(latent=1)
    e = models.DataBag(d, fmap=2)
    assert e.latent == 1 and e['fmap'] == 2 and 'fmap' not in d and e.get('noise') is None
    e.style = 3
    assert e['style'] == 3 and type(e)({k: v for k, v in e.items()}).style == 3
    with pytest.raises(AttributeError):
        e.missing
    g = models.SeqStyleGAN2(16, 512, 8, mconv='seq')
    keys = set(g.state_dict())
    for k in ['style.1.weight', 'style.8.bias', 'latents.latent_avg', 'noises.noise_0', 'input.input',
              'layer2.conv.mconv.modulation.weight', 'layer2.conv.mconv.dconv.weight',
              'layer2.conv.noise.weight', 'layer2.conv.activate.bias', 'to_rgb1.rgb.bias',
              'to_rgb1.rgb.conv.weight', 'to_rgb1.rgb.conv.modulation.bias', 'up_rgb1.kernel',
              'layer3.sconv.mconv.blur.kernel', 'layer6.sconv.mconv.dconv.weight', 'to_rgb3.rgb.conv.weight']:
        assert k in keys, k
    assert [n for n, _ in g.named_children()][:7] == ['bag_in', 'style', 'latents', 'noises', 'input',
                                                      'layer2', 'to_rgb1']
    assert tuple(g.layer6.sconv.mconv.dconv.weight.shape) == (1, 512, 512, 3, 3)
    # rosinality checkpoint keys are accepted
    ros = {}
    for k, v in g.state_dict().items():
        r = k
        r = r.replace('layer2.conv.mconv.dconv.weight', 'conv1.conv.weight').replace('layer2.conv.mconv.', 'conv1.conv.')
        r = r.replace('layer2.conv.', 'conv1.')
        ros[r] = v
    conv = models.convert_rosinality_keys({'convs.2.conv.weight': 0, 'to_rgbs.1.conv.weight': 0,
                                           'to_rgbs.0.upsample.kernel': 0, 'conv1.activate.bias': 0})
    assert set(conv) == {'layer5.sconv.mconv.dconv.weight', 'to_rgb3.rgb.conv.weight', 'up_rgb1.kernel',
                         'layer2.conv.activate.bias'}
    # deepcopy drops derived-weight caches but keeps parameters
    import copy
    c = copy.deepcopy(g)
    assert torch.equal(c.layer4.sconv.mconv.dconv.weight, g.layer4.sconv.mconv.dconv.weight)
    assert c.layer4.sconv.mconv.dconv._derived.store == {}


def test_stylegan_modules_refuse_cpu_tensors():
    g = models.SeqStyleGAN2(8, 512, 2, mconv='seq')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        g(torch.randn(1, 512))


def test_proggan_rewriter_config1_on_cpu_matches_golden():
    g = load_golden('pg64_l6_spire2tree')
    meta = golden_meta(g)
    model = proggan.ProgressiveGenerator(resolution=meta['resolution'])
    synthetic.randomize_(model, seed=0, kind='proggan')
    model.eval()
    zds = zdataset.z_dataset_for_model(model, size=meta['nseeds'])
    with torch.no_grad():
        assert (model(zds[0][0][None]) - torch.from_numpy(g['image'])).abs().max() < 1e-4
    req = load_mask_request(meta['mask'], meta['nseeds'])

    def fresh():
        return ganrewrite.ProgressiveGanRewriter(model, zds, meta['layernum'])
    gw = fresh()
    assert abs(gw.c_matrix.double().norm().item() / float(g['c_matrix_norm']) - 1) < 1e-5
    assert abs(gw.zca_matrix.double().norm().item() / float(g['zca_norm']) - 1) < 1e-3
    obj_acts, _, obj_area, bounds = gw.object_from_selection(*req['object'])
    goal_in, goal_out, _, pb = gw.paste_from_selection(req['paste'][0], req['paste'][1], obj_acts, obj_area)
    assert list(bounds) == list(g['obj_bounds']) and list(pb) == list(g['paste_bounds'])
    assert (goal_in - torch.from_numpy(g['goal_in'])).abs().max() < 1e-4
    assert (goal_out - torch.from_numpy(g['goal_out'])).abs().max() < 1e-4
    mkey = gw.multi_key_from_selection(req.get('key', [req['paste']]), rank=1)
    assert (mkey - torch.from_numpy(g['mkey'])).abs().max() < 2e-3
    mkey = torch.from_numpy(g['mkey'])
    W0 = gw.target_weights().detach().clone()
    for niter in (1, 11):
        gwn = fresh()
        gwn.insert(torch.from_numpy(g['goal_in']), torch.from_numpy(g['goal_out']), mkey, niter=niter)
        dW = gwn.target_weights().detach() - W0
        r = (torch.einsum('oiyx,di->odyx', dW, mkey) - torch.from_numpy(g['dW_%d_cos' % niter])).norm() \
            / float(g['dW_%d_norm' % niter])
        assert r < 1e-4, (niter, r)


def test_running_topk_and_exact_quantiles():
    torch.manual_seed(0)
    x = torch.randn(3000, 4) * torch.tensor([1., 2., .5, 3.]) + torch.tensor([0., 1., -1., 2.])
    rq, tk = runningstats.RunningQuantile(), runningstats.RunningTopK(k=5)
    for i in range(0, 3000, 700):
        rq.add(x[i:i + 700])
        tk.add(x[i:i + 700])
    qs = [0.0, 0.01, 0.5, 0.999, 1.0]
    got = rq.quantiles(qs).numpy()
    # the reference's read-out convention (utils/runningstats.py:550-575): sample i sits at (i+.5)/n
    srt = numpy.sort(x.numpy(), axis=0)
    grid = (numpy.arange(3000) + 0.5) / 3000
    want = numpy.stack([numpy.interp(qs, numpy.r_[0, grid, 1], numpy.r_[srt[0, u], srt[:, u], srt[-1, u]])
                        for u in range(4)])
    assert numpy.abs(got - want).max() < 1e-5
    assert numpy.abs(got[:, 2] - numpy.quantile(x.numpy(), 0.5, axis=0)).max() < 5e-3
    assert numpy.allclose(got[:, 0], x.min(0)[0].numpy()) and numpy.allclose(got[:, -1], x.max(0)[0].numpy())
    assert tuple(rq.quantiles(0.999).shape) == (4,) and rq.size() == 3000
    rank = rq.normalize(x[:7].t())
    emp = (x[None, :, :] < x[:7, None, :]).float().mean(1).t()
    assert (rank - emp).abs().max() < 1e-3 and rank.min() >= 0 and rank.max() <= 1
    v, i = tk.result()
    assert torch.equal(v, x.topk(5, dim=0)[0].t()) and torch.equal(i, x.topk(5, dim=0)[1].t())
    rq2 = runningstats.RunningQuantile(state=rq.state_dict())
    assert torch.equal(rq2.quantiles(qs), rq.quantiles(qs))


def test_tally_topk_and_quantile_cache(tmp_path):
    data = torch.arange(40, dtype=torch.float32)[:, None].repeat(1, 2)
    f = lambda b: (b.sum(1), b[:, :1])
    cache = str(tmp_path / 'tq.npz')
    tk, rq = tally.tally_topk_and_quantile(f, data, k=3, cachefile=cache)
    assert tk.result()[1].tolist() == [39, 38, 37] and abs(rq.median().item() - 19.5) < 0.51
    tk2, rq2 = tally.tally_topk_and_quantile(lambda b: 1 / 0, data, k=3, cachefile=cache)   # served from cache
    assert tk2.result()[1].tolist() == [39, 38, 37] and torch.equal(rq2.median(), rq.median())


def test_feature_statistics_and_frechet_distance():
    from rewriting_amd import samples
    rs = numpy.random.RandomState(0)
    a = rs.randn(500, 6) @ rs.randn(6, 6)
    b = rs.randn(400, 6) @ rs.randn(6, 6) + 0.5
    stats = []
    for arr in (a, b):
        st = samples.FeatureStatistics()
        for i in range(0, len(arr), 64):
            st.add(torch.from_numpy(arr[i:i + 64]).float())
        mu, sigma = st.mean_cov()
        assert numpy.allclose(mu, arr.astype('float32').mean(0), atol=1e-5)
        assert numpy.allclose(sigma, numpy.cov(arr.astype('float32'), rowvar=False), atol=1e-4)
        stats.append((mu, sigma))
    d = samples.frechet_distance(*stats[0], *stats[1])
    assert d > 0 and abs(samples.frechet_distance(*stats[0], *stats[0])) < 1e-6
    # 1-d closed form: (m1-m2)^2 + (s1-s2)^2
    assert abs(samples.frechet_distance([1.0], [[4.0]], [3.0], [[9.0]]) - (4 + 1)) < 1e-9
    g = proggan.ProgressiveGenerator(resolution=8)
    z = samples.seed_latents(g, [3, 4])
    assert torch.equal(z[1:], zdataset.z_sample_for_model(g, size=1, seed=4))


def test_split_precision_switch_only_takes_eligible_stride1_layers(emulated_hip, monkeypatch):
    """RW_CONV_PRECISION=bf16x6 routes exactly the eligible stride-1 convolutions (map >= 24 wide,
    Cin % 16 == 0, Cout % 64 == 0) to conv3x3_bf16x6 and changes nothing else; off by default."""
    import torch
    from rewriting_amd import hip
    from tests.conftest import build_stylegan
    model = build_stylegan(64, 0.5, device='cpu')
    z = torch.randn(2, 512)
    seen = []
    orig = hip.conv3x3_bf16x6

    def spy(x, wb, out_ch, *a, **k):
        seen.append((x.shape[1], out_ch, x.shape[-1]))
        return orig(x, wb, out_ch, *a, **k)
    monkeypatch.setattr(hip, 'conv3x3_bf16x6', spy)
    monkeypatch.setenv('RW_UP_FUSED2', '0')        # (the fused upsampling kernel is an fp32-precision route: the switch turns it off)
    with torch.no_grad():
        default = model(z)
        monkeypatch.setenv('RW_CONV_ALGO', 'direct')               # the direct sum, which the split path restates
        base = model(z)
        monkeypatch.delenv('RW_CONV_ALGO')
    assert seen == []                                              # default: exact fp32 kernels only
    assert (default - base).abs().max().item() < 1e-4             # (default = Winograd on the same two layers)
    monkeypatch.setenv('RW_CONV_PRECISION', 'bf16x6')
    with torch.no_grad():
        img = model(z)
    assert seen and all(w >= 24 and i % 16 == 0 and o % 64 == 0 for i, o, w in seen), seen
    assert sorted(w for _, _, w in seen) == [32, 64]               # layers 8 and 10 of the 64^2 generator
    assert torch.equal(img, base)                                  # emulated product is exact


def test_sweep_batch_never_starves_a_rank(monkeypatch):
    from rewriting_amd import parallel
    from rewriting_amd.rewrite import ganrewrite

    class Fake(ganrewrite.ProgressiveGanRewriter):
        def __init__(self, n):
            self.zds = list(range(n))
    for world, n, want in ((1, 1000, 510), (8, 1000, 130), (4, 10000, 500), (8, 10000, 420), (8, 100, 10), (2, 35, 20),
                           (1, 300, 300), (1, 2040, 510), (2, 1000, 500)):
        monkeypatch.setattr(parallel, 'shard', lambda world=world: (0, world) if world > 1 else None)
        b = Fake(n)._sweep_batch()
        assert b == want and b % 10 == 0, (world, n, b)
        launches = (n + b - 1) // b
        assert launches >= min(world, n // 10)                     # at least one batch per rank
        if n >= 100 * world:
            assert launches % world == 0                           # ... and the same number for every rank
    # the key map of a launch stays within sweep_bytes: 128 x 256 x 256 floats per seed (layer 14 of the 1024 model)
    monkeypatch.setattr(parallel, 'shard', lambda: None)
    big = Fake(1000)
    big.k_shape = (1, 128, 256, 256)
    assert big._sweep_batch() == 60
    small = Fake(1000)
    small.k_shape = (1, 512, 32, 32)
    assert small._sweep_batch() == 510


def test_final_pair_identifies_last_styled_conv_and_its_to_rgb():
    """SeqStyleGAN2._final_pair: the layer whose ToRGB is fused into its epilogue in the un-hooked forward."""
    from rewriting_amd.utils.stylegan2 import models as sg
    g = sg.SeqStyleGAN2(64, 512, 2, mconv='seq')
    sconv, torgb, idx = g._final_pair()
    assert sconv is g.layer10.sconv and torgb is g.to_rgb5.rgb
    assert idx == g.n_latent - 1 == 9                      # ToRGB of the last resolution takes the last latent
    assert torgb.skip and torgb.conv.in_channel == sconv.mconv.dconv.out_channel
    from rewriting_amd import hip
    assert hip.to_rgb_fusable(32, 32, 1024) and hip.to_rgb_fusable(64, 64, 512)
    assert not hip.to_rgb_fusable(128, 128, 256) and not hip.to_rgb_fusable(32, 32, 16)


def _check_digest(g, prefix, img, bar=1e-4):
    """Image batch against a fixture digest (oracle/make_golden.py image_digest): strided sub-sample, the four
    full-resolution crops, float64 row / column sums (every pixel enters one of each), norm."""
    stride = int(g[prefix + 'stride'])
    assert list(img.shape) == list(g[prefix + 'shape'])
    want = torch.from_numpy(g[prefix + 'strided'])
    scale = max(1.0, want.abs().max().item())
    worst = (img[:, :, ::stride, ::stride].cpu() - want).abs().max().item()
    crops = g[prefix + 'crops']
    c = crops.shape[-1]
    for k, (y, x) in enumerate(g[prefix + 'crop_origin']):
        worst = max(worst, (img[:, :, y:y + c, x:x + c].cpu() - torch.from_numpy(crops[k])).abs().max().item())
    assert worst < bar * scale, (worst, scale)
    width = img.shape[-1]
    assert numpy.abs(img.double().sum(3).cpu().numpy() - g[prefix + 'rowsum']).max() < bar * scale * width ** 0.5 * 4
    assert numpy.abs(img.double().sum(2).cpu().numpy() - g[prefix + 'colsum']).max() < bar * scale * width ** 0.5 * 4
    assert abs(img.double().norm().item() / float(g[prefix + 'norm']) - 1) < 1e-5
    return worst


def test_proggan_rewriter_config1_at_its_own_size_matches_reference_golden(tmp_path):
    """BASELINE.json configs[0] as SURVEY.md 8d states it: ProgressiveGenerator(resolution=256), 1000 seeds, layer 6,
    notebooks/masks/proggan/church/spire2tree.json with its real seed indices (object 971, paste 18), rank 1 --
    against pg256_l6_spire2tree_1000.npz, written by the reference's own utils/proggan.py:65-193 and
    rewrite/ganrewrite.py:24-298 (oracle/make_golden.py golden_proggan_full).  CPU by definition of the config."""
    g = load_golden('pg256_l6_spire2tree_1000')
    meta = golden_meta(g)
    assert meta['resolution'] == 256 and meta['nseeds'] == 1000 and meta['layernum'] == 6
    model = proggan.ProgressiveGenerator(resolution=256)
    synthetic.randomize_(model, seed=0, kind='proggan')
    model.eval()
    zds = zdataset.z_dataset_for_model(model, size=1000)
    req = load_mask_request(meta['mask'])
    assert req['object'][0] == 971 and req['paste'][0] == 18 and 'key' not in req
    zs = torch.stack([zds[i][0] for i in (0, req['paste'][0])])
    with torch.no_grad():
        _check_digest(g, 'image/', model(zs))
    cachedir = str(tmp_path / 'cache')

    def fresh():
        return ganrewrite.ProgressiveGanRewriter(model, zds, 6, cachedir=cachedir)
    gw = fresh()
    assert list(gw.k_shape) == list(g['k_shape']) == [1, 512, 16, 16] and list(gw.v_shape) == list(g['v_shape'])
    # statistics: as close to the reference's C as the reference is to the float64 accumulation of its own keys
    C = gw.c_matrix
    cmax = C.abs().max().item()
    ref_abs, ref_rel = float(g['c_ref_vs_exact_max']), float(g['c_ref_vs_exact'])
    assert (C[::4, ::4].double() - torch.from_numpy(g['c_exact'])).abs().max().item() < 1.2 * ref_abs + 2e-5 * cmax
    assert (C[::4, ::4] - torch.from_numpy(g['c_matrix'])).abs().max().item() < 1.2 * ref_abs + 2e-5 * cmax
    assert (C.diag() - torch.from_numpy(g['c_matrix_diag'])).abs().max().item() < 1.2 * ref_abs + 2e-5 * cmax
    assert abs(C.double().norm().item() / float(g['c_matrix_norm']) - 1) < 1.5 * ref_rel + 1e-5
    Z = gw.zca_matrix
    zbar = 1.2 * float(g['zca_ref_vs_exact_max']) + 2e-4 * Z.abs().max().item()
    assert (Z[::4, ::4] - torch.from_numpy(g['zca'])).abs().max().item() < zbar
    assert (Z[::4, ::4] - torch.from_numpy(g['zca_exact'])).abs().max().item() < zbar
    # goals and the context direction
    obj_acts, _, obj_area, bounds = gw.object_from_selection(*req['object'])
    goal_in, goal_out, _, pb = gw.paste_from_selection(req['paste'][0], req['paste'][1], obj_acts, obj_area)
    assert list(bounds) == list(g['obj_bounds']) and list(pb) == list(g['paste_bounds'])
    assert (obj_area - torch.from_numpy(g['obj_area'])).abs().max() < 1e-6
    for got, nm in ((goal_in, 'goal_in'), (goal_out, 'goal_out')):
        want = torch.from_numpy(g[nm])
        assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item()), nm
    mkey = gw.multi_key_from_selection([req['paste']], rank=1)
    assert ganrewrite.all_obs.shape[0] == int(g['n_sel'])
    assert (mkey * torch.from_numpy(g['mkey'])).sum().item() > 1 - 1e-5
    # the solve on the reference's own goal and direction: 1e-4 relative (north_star) at every recorded horizon
    mkey = torch.from_numpy(g['mkey'])
    gin, gout = torch.from_numpy(g['goal_in']), torch.from_numpy(g['goal_out'])
    W0 = gw.target_weights().detach().clone()
    assert (subsample(W0, 16384) - torch.from_numpy(g['W0_sub'])).abs().max().item() == 0.0

    def rel(W, tag):
        dW = W - W0
        r_cos = ((torch.einsum('oiyx,di->odyx', dW, mkey) - torch.from_numpy(g['dW_%s_cos' % tag])).norm()
                 / float(g['dW_%s_norm' % tag])).item()
        gsub = torch.from_numpy(g['dW_%s_sub' % tag])
        return max(r_cos, ((subsample(dW, 8192) - gsub).norm() / gsub.norm()).item())
    for niter in (1, 11, 101):
        gwn = fresh()
        snaps, losses = {}, []

        def cb(it, loss, snaps=snaps, gwn=gwn, losses=losses):
            losses.append(float(loss))
            if it in (9, 99):
                snaps[it + 1] = gwn.target_weights().detach().clone()
        gwn.insert(gin, gout, mkey, niter=niter, piter=10, lr=0.05, update_callback=cb)
        snaps[niter] = gwn.target_weights().detach().clone()
        for n, W in snaps.items():
            assert rel(W, '%d' % n) < 1e-4, (n, rel(W, '%d' % n))
        if niter == 101:
            assert numpy.abs(numpy.array(losses) - g['losses']).max() < 1e-5
            with torch.no_grad():
                _check_digest(g, 'edited_image/', gwn.sample_image_from_latent(zs))


def test_proggan_linear_insert_learns_lambda_on_a_4d_weight():
    """ProgressiveGanRewriter.linear_insert (rewrite/ganrewrite.py:201-252).  The reference sizes Lambda from
    ws[3], ws[4] of a 5-d weight and raises IndexError on ProgGAN's nn.Conv2d weight, so there is no reference
    output to compare with: the check is the definition -- Adam(lr) on Lambda with weight = W0 + Lambda . d, written
    independently here with a functional convolution -- plus the rank-r structure and the restored module."""
    g = load_golden('pg64_l6_spire2tree')
    meta = golden_meta(g)
    model = proggan.ProgressiveGenerator(resolution=meta['resolution'])
    synthetic.randomize_(model, seed=0, kind='proggan')
    model.eval()
    zds = zdataset.z_dataset_for_model(model, size=meta['nseeds'])
    gw = ganrewrite.ProgressiveGanRewriter(model, zds, meta['layernum'], use_linear_insert=True)
    mkey = torch.from_numpy(g['mkey'])
    gin, gout = torch.from_numpy(g['goal_in']), torch.from_numpy(g['goal_out'])
    W0 = gw.target_weights().detach().clone()
    conv = [m for m in gw.target_model.modules() if getattr(m, 'weight', None) is gw.target_weights()][0]
    losses = []
    gw.insert(gin, gout, mkey, niter=5, lr=0.05, update_callback=lambda it, loss: losses.append(float(loss)))
    W = gw.target_weights()
    assert isinstance(conv._parameters.get('weight'), torch.nn.Parameter) and 'forward' not in conv.__dict__
    dW = (W - W0).detach()
    assert dW.norm() > 0 and ((dW - ganrewrite.projected_conv(dW, mkey)).norm() / dW.norm()).item() < 1e-5
    # the definition, written independently
    lam = torch.zeros(W0.shape[0], 1, 3, 3, requires_grad=True)
    opt = torch.optim.Adam([lam], lr=0.05)
    want_losses = []
    for _ in range(5):
        w = W0 + torch.einsum('odyx,di->oiyx', lam, mkey)
        loss = torch.nn.functional.l1_loss(gout, torch.nn.functional.conv2d(
            gin, w, conv.bias, conv.stride, conv.padding))
        opt.zero_grad()
        loss.backward()
        opt.step()
        want_losses.append(loss.item())
    want = torch.einsum('odyx,di->oiyx', lam.detach(), mkey)
    assert ((dW - want).norm() / want.norm()).item() < 1e-5
    assert numpy.abs(numpy.array(losses) - numpy.array(want_losses)).max() < 1e-6


def test_running_quantile_keeps_streaming_after_a_save_and_after_a_cache_load():
    """utils/runningstats.py:269-620: the reference's sketch can be saved and extended.  Here: state_dict() does not
    finalise the statistic, and a statistic rebuilt from a state dict (or compressed) takes more samples -- kept
    exactly beside the levels, weight 1 each -- with read-outs of the weighted union."""
    torch.manual_seed(1)
    x = torch.randn(6000, 3) * torch.tensor([1., 3., .3]) + torch.tensor([0., -2., 5.])
    qs = [0.0, 0.001, 0.05, 0.5, 0.95, 0.999, 1.0]
    whole = runningstats.RunningQuantile(r=256)
    whole.add(x)
    want = whole.quantiles(qs)
    # saving in the middle of a stream changes nothing
    rq = runningstats.RunningQuantile(r=256)
    rq.add(x[:2500])
    state = rq.state_dict()
    rq.add(x[2500:])
    assert torch.equal(rq.quantiles(qs), want) and rq.size() == 6000
    # a loaded statistic (2r = 512 retained of 2500) extended by the rest of the stream
    loaded = runningstats.RunningQuantile(state=state)
    assert loaded.size() == 2500
    loaded.add(x[2500:4000])
    loaded.add(x[4000:])
    assert loaded.size() == 6000 and loaded.batchcount == 3
    got = loaded.quantiles(qs)
    spread = (want[:, -1] - want[:, 0])[:, None]
    assert ((got - want).abs() / spread).max() < 2e-3            # body: rank error of the compressed part
    assert torch.equal(got[:, 0], want[:, 0]) and torch.equal(got[:, -1], want[:, -1])      # extremes exact
    assert torch.allclose(loaded.mean(), x.mean(0), atol=2e-2) and torch.equal(loaded.minmax(), whole.minmax())
    rank = loaded.normalize(x[:50].t())
    emp = (x[None, :, :] < x[:50, None, :]).float().mean(1).t()
    assert (rank - emp).abs().max() < 5e-3
    # compressing the union, saving and loading it again: the same statistic in the reference's schema
    again = runningstats.RunningQuantile(state=loaded.state_dict())
    assert again.size() == 6000 and sum(lv.shape[1] for lv in again._levels) <= 512
    assert ((again.quantiles(qs) - want).abs() / spread).max() < 4e-3
    assert abs(sum(lv.shape[1] * 2 ** l for l, lv in enumerate(again._levels)) - 6000) == 0
    loaded.compress_()
    assert torch.equal(loaded.quantiles(qs), again.quantiles(qs))


def test_hooked_models_leave_a_layer_whose_weight_requires_grad_on_the_fp32_kernel(emulated_hip, monkeypatch):
    """A hooked / sliced model runs its stride-1 layers of maps from 32^2 up as direct sums on the 16-bit pipe
    (DemodulatedConv2dF.hooked_direct16; tests/test_emulated_path.py holds which layers and the statistics bar) -- but not
    a layer whose weight requires grad (an autograd `insert`): every re-packing of a changing weight would read its maximum
    back to the host, two launches and a sync per layer and optimiser step."""
    import torch
    from rewriting_amd import hip
    from rewriting_amd.utils import nethook
    from tests.conftest import build_stylegan
    model = build_stylegan(64, 0.5, device='cpu')
    z = torch.randn(2, 512)
    seen = []
    orig = hip.conv3x3_direct16

    def spy(x, wp, out_ch, *a, **k):
        seen.append((x.shape[1], out_ch, x.shape[-1]))
        return orig(x, wp, out_ch, *a, **k)
    monkeypatch.setattr(hip, 'conv3x3_direct16', spy)

    def hooked_run():
        del seen[:]
        with nethook.InstrumentedModel(model) as inst:
            inst.retain_layer('layer7', detach=False)
            return inst(z)
    with torch.no_grad():
        want = hooked_run()
    assert sorted(w for _, _, w in seen) == [32, 64], seen         # layers 8 and 10 of the 64^2 generator
    layer = [m for n, m in model.named_modules() if n.startswith('layer10') and hasattr(m, 'hooked_direct16')][0]
    nethook.set_requires_grad(False, model)
    assert layer.hooked_direct16(64, 64)
    layer.weight.requires_grad_(True)
    try:
        got = hooked_run()
        assert sorted(w for _, _, w in seen) == [32], seen          # layer 8 only: layer 10's weight is being optimised
        assert (got.detach() - want).abs().max().item() < 1e-4     # the fp32 kernel's result: the image barely moves
    finally:
        layer.weight.requires_grad_(False)
