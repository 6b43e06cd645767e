"""GPU parity of the assembled path: generator, key statistics, rewriter and fused solver against
the reference-generated goldens and the travelling oracle, plus size-independent properties
at BASELINE.json's full sizes (256^2 / 1024^2 generators, 1000-seed statistics, 2001-step
solve).  Tolerances follow BASELINE.json's north_star: images within 1e-3 L-inf, edited
weight delta within 1e-4 relative at short horizons (1, 10, 11, 100 steps); the 2001-step
divergence is REPORTED next to the oracle's own self-divergence."""
import json
import math
import os

import numpy
import pytest
import torch

from tests.conftest import (build_stylegan, golden_meta, load_golden, load_mask_request,
                            oracle_state_dict, subsample)

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize('fuse', ['1', '0'])
@pytest.mark.parametrize('impl', ['auto', 'generic', 'direct'])
@pytest.mark.parametrize('name', ['gen_s32_t05', 'gen_s64_cm1'])
def test_generator_matches_reference_golden(monkeypatch, name, impl, fuse):
    monkeypatch.setenv('RW_FUSE', fuse)
    monkeypatch.setenv('RW_CONV_IMPL', impl)
    g = load_golden(name)
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'], device=DEV)
    z = torch.from_numpy(g['z']).to(DEV)
    store, handles = {}, []
    if fuse == '0':
        for lname, mod in model.named_modules():
            if lname and len(list(mod.children())) == 0:
                handles.append(mod.register_forward_hook(
                    lambda m, i, o, lname=lname: store.__setitem__(lname, o)))
    with torch.no_grad():
        img = model(z)
    for h in handles:
        h.remove()
    want = torch.from_numpy(g['image'])
    assert torch.isfinite(img).all()
    assert (img.cpu() - want).abs().max().item() < 1e-3          # north_star: 1e-3 L-inf
    assert (img.cpu() - want).abs().max().item() < 1e-4          # what fp32 MFMA actually delivers
    checked = 0
    for key in g.files:
        if not (key.startswith('stage/') and key.endswith('/sub')) or key[6:-4] not in store:
            continue
        lname = key[6:-4]
        out = store[lname]
        if isinstance(out, dict):
            field = 'output' if lname.startswith('up_rgb') else 'style' if lname.endswith('modulation') \
                else 'latent' if (lname.startswith('style.') or lname == 'latents') else 'fmap'
            if field not in out:
                continue
            out = out[field]
        w = torch.from_numpy(g[key])
        assert (subsample(out) - w).abs().max().item() < 1e-4 * max(1.0, w.abs().max().item()), lname
        checked += 1
    assert fuse == '1' or checked >= 40


def test_generator_with_split_bf16x6_convs_matches_reference_golden(monkeypatch):
    """The opt-in bf16x6 convolutions (32^2 and 64^2 layers of this generator) hold the same image bar."""
    g = load_golden('gen_s64_cm1')
    meta = golden_meta(g)
    z = torch.from_numpy(g['z']).to(DEV)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'], device=DEV)
    with torch.no_grad():
        exact = model(z)
    monkeypatch.setenv('RW_CONV_PRECISION', 'bf16x6')
    from rewriting_amd import hip
    calls = []
    orig = hip.conv3x3_bf16x6
    monkeypatch.setattr(hip, 'conv3x3_bf16x6', lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    with torch.no_grad():
        img = model(z)
    assert len(calls) >= 2                                        # the path was actually taken
    want = torch.from_numpy(g['image'])
    assert (img.cpu() - want).abs().max().item() < 1e-4
    assert (img - exact).abs().max().item() < 2e-5


def test_hook_surface_on_gpu():
    from rewriting_amd.utils import nethook
    g = load_golden('gen_s32_t05')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], device=DEV)
    z = torch.from_numpy(g['z']).to(DEV)
    with torch.no_grad():
        base = model(z)
    with nethook.InstrumentedModel(model) as inst:
        inst.retain_layer('layer6.sconv.mconv.adain', detach=False)
        inst.edit_layer('layer6.sconv.activate', rule=lambda x, imodel: type(x)(x, fmap=x.fmap * 0))
        with torch.no_grad():
            edited = inst(z)
        key = inst.retained_layer('layer6.sconv.mconv.adain').fmap
    assert (subsample(key) - torch.from_numpy(g['stage/layer6.sconv.mconv.adain/sub'])).abs().max() < 1e-4
    assert (edited - base).abs().max().item() > 1e-3             # the edit rule was applied
    with torch.no_grad():
        assert torch.equal(model(z), base)                       # hooks removed, fusion back on


def _make_rewriter(meta, **kw):
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset
    model = build_stylegan(meta['size'], meta['truncation'], device=DEV)
    zds = zdataset.z_dataset_for_model(model, size=meta['nseeds'])
    return ganrewrite.SeqStyleGanRewriter(model, zds, meta['layernum'], cachedir=kw.pop('cachedir', None),
                                          low_rank_insert=True, key_method='zca', tight_paste=True, **kw)


def test_rewriter_edit_matches_reference_golden(tmp_path):
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils.stylegan2.models import DataBag
    g = load_golden('rw_s64_l8_horsehat')
    meta = golden_meta(g)
    gw = _make_rewriter(meta, cachedir=str(tmp_path / 'cache'))
    assert list(gw.k_shape) == list(g['k_shape']) and list(gw.v_shape) == list(g['v_shape'])
    C = gw.c_matrix.cpu()
    assert abs(C.double().norm().item() / float(g['c_matrix_norm']) - 1) < 1e-5
    assert (C[::4, ::4] - torch.from_numpy(g['c_matrix'])).abs().max() < 1e-4 * C.abs().max()
    assert (C.diag() - torch.from_numpy(g['c_matrix_diag'])).abs().max() < 1e-4 * C.abs().max()
    assert abs(gw.zca_matrix.double().norm().item() / float(g['zca_norm']) - 1) < 2e-3
    # cache written in the reference's npz schema and re-used by a second rewriter
    cached = numpy.load(str(tmp_path / 'cache' / 'r2m.npz'), allow_pickle=True)
    assert sorted(cached.files) == ['constructor', 'count', 'mom2', 'sample_size']
    assert int(cached['count']) == meta['nseeds'] * 32 * 32
    gw2 = _make_rewriter(meta, cachedir=str(tmp_path / 'cache'))
    assert torch.equal(gw2.c_matrix, gw.c_matrix)
    req = load_mask_request(meta['mask'], meta['nseeds'])
    obj_acts, _, obj_area, bounds = gw.object_from_selection(*req['object'])
    assert list(bounds) == list(g['obj_bounds'])
    goal_in, goal_out, _, pb = gw.paste_from_selection(req['paste'][0], req['paste'][1], obj_acts, obj_area)
    assert list(pb) == list(g['paste_bounds'])
    assert (goal_in.fmap.cpu() - torch.from_numpy(g['goal_in_fmap'])).abs().max() < 1e-4
    assert (goal_out.fmap.cpu() - torch.from_numpy(g['goal_out_fmap'])).abs().max() < 1e-4
    mkey = gw.multi_key_from_selection(req['key'], rank=1)
    assert ganrewrite.all_obs.shape[0] == int(g['n_sel'])
    assert (mkey.cpu() - torch.from_numpy(g['mkey'])).abs().max() < 2e-3
    # ---- the solve on identical inputs (golden goal + golden context), short horizons
    mkey = torch.from_numpy(g['mkey']).to(DEV)
    gin = DataBag(goal_in, fmap=torch.from_numpy(g['goal_in_fmap']).to(DEV),
                  style=torch.from_numpy(g['goal_in_style']).to(DEV))
    gout = DataBag(goal_out, fmap=torch.from_numpy(g['goal_out_fmap']).to(DEV))
    W0 = gw.target_weights().detach().clone()
    report = {}
    for niter in (1, 11):
        gwn = _make_rewriter(meta, cachedir=str(tmp_path / 'cache'))
        gwn.insert(gin, gout, mkey, niter=niter, piter=10, lr=0.05)
        dW = (gwn.target_weights().detach() - W0)[0]
        cos = torch.einsum('oiyx,di->odyx', dW, mkey).cpu()
        r = ((cos - torch.from_numpy(g['dW_%d_cos' % niter])).norm() / float(g['dW_%d_norm' % niter])).item()
        report[niter] = r
        assert r < 1e-4, (niter, r)                                  # north_star: 1e-4 relative
        assert (dW - ganrewrite.projected_conv(dW[None], mkey)[0]).norm() / dW.norm() < 1e-4
    # 10 / 100 (un-projected states) and 101 through the callback path and the graph path.
    # At >= 100 steps the reference's own fp32 trajectory has left the exact (fp64) one by ~4e-4
    # (L1 sign gradients + Adam amplify rounding chaotically; profiles/r01_solve_divergence_301.json),
    # so the bar there is the EXACT arithmetic: the oracle's explicit update in float64.
    from oracle import restatement as R
    sd_w = W0.cpu()
    _, _, exact = R.insert_explicit(
        sd_w, torch.from_numpy(g['goal_in_fmap']), torch.from_numpy(g['goal_in_style']),
        torch.from_numpy(g['goal_out_fmap']), gw.target_model.layer8.sconv.activate.bias.detach().cpu(),
        gw.target_model.layer8.sconv.noise.weight.detach().cpu(), torch.from_numpy(g['mkey']),
        niter=101, snapshots=(10, 100, 101), dtype=torch.float64)
    for use_cb in (True, False):
        gwn = _make_rewriter(meta, cachedir=str(tmp_path / 'cache'))
        snaps, losses = {}, []

        def cb(it, loss):
            losses.append(loss)
            if it in (9, 99):
                snaps[it + 1] = gwn.target_weights().detach().clone()
        t_ms = gwn.insert(gin, gout, mkey, niter=101, piter=10, lr=0.05,
                          update_callback=cb if use_cb else None, return_timing=True)
        assert t_ms > 0
        snaps[101] = gwn.target_weights().detach().clone()
        for n, W in snaps.items():
            dW = (W - W0)[0].cpu()
            ex = (exact[n].float() - sd_w)[0]
            gsub = torch.from_numpy(g['dW_%d_sub' % n])
            r_exact = rel(dW, ex)
            r_gold = ((subsample(dW, 8192) - gsub).norm() / gsub.norm()).item()
            gold_self = ((subsample(ex, 8192) - gsub).norm() / gsub.norm()).item()
            tag = '%s%d' % ('cb' if use_cb else 'graph', n)
            report[tag] = dict(gpu_vs_exact=r_exact, gpu_vs_golden=r_gold, golden_vs_exact=gold_self)
            assert r_exact < 1e-4, (tag, r_exact)                       # north_star bar, exact arithmetic
            assert r_gold < 3 * gold_self + 2e-4, (tag, r_gold, gold_self)
        if use_cb:
            got = torch.stack(losses).cpu().numpy()
            assert numpy.abs(got - g['losses'])[:20].max() < 1e-5
            assert numpy.abs(got - g['losses']).max() < 5e-4
    with torch.no_grad():
        zs = torch.cat([gwn.get_z(i) for i in (0, 1)])
        img = gwn.sample_image_from_latent(zs)
    assert (img.cpu() - torch.from_numpy(g['edited_image'])).abs().max().item() < 1e-3
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/solve_parity.json', 'w') as f:
        json.dump(report, f, indent=1)


def test_rewriter_erase_matches_reference_golden():
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset
    from rewriting_amd.utils.stylegan2.models import DataBag
    g = load_golden('rw_s64_l6_erase')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], device=DEV)
    zds = zdataset.z_dataset_for_model(model, size=meta['nseeds'])
    gw = ganrewrite.SeqStyleGanRewriter(model, zds, meta['layernum'], low_rank_insert=True,
                                        low_rank_gradient=True)
    req = load_mask_request(meta['mask'], meta['nseeds'])
    with torch.no_grad():
        scale = gw.square_scales_for_units().cpu()
        assert (scale - torch.from_numpy(g['unit_scale'])).abs().max() < 1e-4 * float(g['unit_scale'].max())
        units = gw.normdissect_units(req['key'], meta['drank'])
        assert set(units.tolist()) == set(g['d_units'].tolist())
        goal_in, goal_out = gw.erase_from_selection(req['paste'][0], req['paste'][1], req['key'], meta['drank'])
    assert (goal_in.fmap.cpu() - torch.from_numpy(g['goal_in_fmap'])).abs().max() < 1e-4
    assert (goal_out.fmap.cpu() - torch.from_numpy(g['goal_out_fmap'])).abs().max() < 1e-4
    mkey = torch.from_numpy(g['mkey']).to(DEV)
    gin = DataBag(goal_in, fmap=torch.from_numpy(g['goal_in_fmap']).to(DEV),
                  style=torch.from_numpy(g['goal_in_style']).to(DEV))
    gout = DataBag(goal_out, fmap=torch.from_numpy(g['goal_out_fmap']).to(DEV))
    W0 = gw.target_weights().detach().clone()
    gw.insert(gin, gout, mkey, niter=11, piter=10, lr=0.05)       # low_rank_gradient path, 16x16 map
    dW = (gw.target_weights().detach() - W0)[0]
    r = ((torch.einsum('oiyx,di->odyx', dW, mkey).cpu() - torch.from_numpy(g['dW_11_cos'])).norm()
         / float(g['dW_11_norm'])).item()
    assert r < 1e-4, r
    gw.zero(mkey, amount=0.0)                                      # zero(): W <- W - P(W)
    W = gw.target_weights().detach()
    assert ganrewrite.projected_conv(W, mkey).abs().max().item() < 1e-4 * W.abs().max().item()


def test_solver_against_oracle_explicit_arithmetic():
    """Independent of the goldens: random layer, rank-2 context, rectangular crop; HIP solver vs
    oracle.insert_explicit at 1/10/11/30 steps, graph and eager paths."""
    from rewriting_amd.rewrite import hipsolve
    from oracle import restatement as R
    rs = numpy.random.RandomState(4)
    O = I = 128
    h, w = 6, 11
    W0 = torch.from_numpy(rs.randn(1, O, I, 3, 3).astype('float32'))
    key = torch.from_numpy(rs.randn(1, I, h, w).astype('float32'))
    style = torch.from_numpy((1 + 0.3 * rs.randn(1, I)).astype('float32'))
    val = torch.from_numpy(rs.randn(1, O, h, w).astype('float32'))
    bias = torch.from_numpy((0.1 * rs.randn(O)).astype('float32'))
    nw = torch.tensor([0.1])
    ctx = torch.linalg.qr(torch.from_numpy(rs.randn(I, 2).astype('float32')))[0].t().contiguous()
    for lrg in (False, True):
        for mode in ('one_launch', 'graph', 'eager'):
            os.environ['RW_SOLVE_ONE_LAUNCH'] = '1' if mode == 'one_launch' else '0'
            os.environ['RW_SOLVE_GRAPH'] = '0' if mode == 'eager' else '1'
            for n in (1, 10, 11, 30, 41):
                Wd = W0.to(DEV).clone()
                s = hipsolve.run(Wd, key.to(DEV), style.to(DEV), val.to(DEV), bias.to(DEV), nw.to(DEV),
                                 ctx.to(DEV), niter=n, piter=10, lr=0.05, low_rank_insert=True,
                                 low_rank_gradient=lrg)
                assert s.one_launch == (mode == 'one_launch')
                # a run of n iterations projects at its last iteration; compare with the oracle run
                # of the same length
                _, l2, s2 = R.insert_explicit(W0, key, style, val, bias, nw, ctx, niter=n, piter=10,
                                              low_rank_gradient=lrg, snapshots=(n,))
                r = rel(Wd - W0.to(DEV), s2[n] - W0)
                assert r < 1e-4, (lrg, mode, n, r)
                assert numpy.abs(s.losses.cpu().numpy() - numpy.array(l2)).max() < 1e-5
    os.environ.pop('RW_SOLVE_GRAPH', None)
    os.environ.pop('RW_SOLVE_ONE_LAUNCH', None)


# the last four: crops beyond the LDS (streamed row by row: round 4), the watermark erase's own shape first;
# lrg = low_rank_gradient (the erase's setting: the gradient phase then needs no key at all); 'stream': the streaming
# kernel forced on a crop that would fit (RW_SOLVE_STREAM=1) -- odd and even widths, one row, rank 8
@pytest.mark.parametrize('O,I,h,w,rank,plain,lrg,stream', [
    (512, 512, 5, 8, 1, False, False, False), (256, 512, 6, 8, 3, False, False, False),
    (64, 64, 1, 1, 1, False, False, False), (128, 256, 4, 4, 8, True, False, False),
    (64, 192, 3, 17, 2, False, False, False), (128, 256, 8, 12, 1, False, False, False),
    (64, 128, 7, 1, 2, False, False, False),
    (512, 512, 16, 16, 1, False, True, False), (512, 512, 16, 16, 2, False, False, False),
    (128, 512, 12, 12, 3, False, True, False), (64, 256, 13, 15, 1, True, False, False),
    (128, 256, 4, 4, 8, True, True, True), (64, 128, 7, 1, 2, False, False, True), (64, 192, 3, 16, 2, False, True, True),
    (128, 256, 8, 12, 1, False, False, True)])
def test_one_launch_solver_equals_step_solver(O, I, h, w, rank, plain, lrg, stream, monkeypatch):
    """rw_solve_run_f32 (one workgroup per pair of out-channels, the whole solve in one launch) against the step
    kernels on the same problem: the same update in another summation order, so 11 iterations agree to rounding;
    a per-iteration callback drives the same kernel one iteration per launch (state round-trips through HBM) with the
    projection done by the stand-alone kernel after the callback, as the reference orders them."""
    from rewriting_amd import hip
    from rewriting_amd.rewrite import hipsolve
    rs = numpy.random.RandomState(O + I + h)
    W0 = torch.from_numpy(rs.randn(1, O, I, 3, 3).astype('float32')).to(DEV)
    key = torch.from_numpy(rs.randn(1, I, h, w).astype('float32')).to(DEV)
    style = torch.from_numpy((1 + 0.3 * rs.randn(1, I)).astype('float32')).to(DEV)
    val = torch.from_numpy(rs.randn(1, O, h, w).astype('float32')).to(DEV)
    bias = None if plain else torch.from_numpy((0.1 * rs.randn(O)).astype('float32')).to(DEV)
    nw = None if plain else torch.tensor([0.1], device=DEV)
    ctx = torch.linalg.qr(torch.from_numpy(rs.randn(I, rank).astype('float32')))[0].t().contiguous().to(DEV)
    assert hip.solve_run_supported(O, I, h, w, rank, False, False)
    if stream:
        monkeypatch.setenv('RW_SOLVE_STREAM', '1')
    res = {}
    for mode in ('one_launch', 'step', 'callback'):
        os.environ['RW_SOLVE_ONE_LAUNCH'] = '0' if mode == 'step' else '1'
        Wd = W0.clone()
        seen = []
        cb = (lambda it, loss: seen.append(float(loss))) if mode == 'callback' else None
        s = hipsolve.run(Wd, key, style, val, bias, nw, ctx, niter=11, piter=10, lr=0.05, low_rank_insert=True,
                         low_rank_gradient=lrg, upsample=False, update_callback=cb)
        assert s.one_launch == (mode != 'step')
        res[mode] = (Wd, s.losses.clone())
        if mode == 'callback':
            assert numpy.array_equal(numpy.array(seen, dtype='float32'), s.losses.cpu().numpy())
    os.environ.pop('RW_SOLVE_ONE_LAUNCH', None)
    assert rel(res['one_launch'][0] - W0, res['callback'][0] - W0) < 1e-5
    assert (res['one_launch'][1] - res['callback'][1]).abs().max().item() < 1e-6
    r = rel(res['one_launch'][0] - W0, res['step'][0] - W0)
    assert r < 1e-4, r
    assert (res['one_launch'][1] - res['step'][1]).abs().max().item() < 1e-5


def test_one_launch_solver_limits():
    from rewriting_amd import hip
    assert hip.solve_run_supported(512, 512, 6, 8, 1, False, False)          # 8 x 9 padded rows: within the LDS
    assert hip.solve_run_supported(256, 256, 8, 12, 1, False, False)
    assert hip.solve_run_supported(512, 512, 8, 9, 1, False, False)          # beyond the LDS: streamed (w <= 16)
    assert hip.solve_run_supported(512, 512, 16, 16, 8, False, False)
    assert not hip.solve_run_supported(512, 512, 16, 17, 1, False, False)     # wider than the row buffers
    assert hip.solve_run_scratch_elems(512, 512, 5, 8, 2001) == 2001 * 512
    assert hip.solve_run_scratch_elems(512, 512, 16, 16, 2001) == 2001 * 512 + 309 * 512
    assert not hip.solve_run_supported(512, 512, 4, 4, 1, True, False)       # upsampling targets: step path
    assert not hip.solve_run_supported(512, 512, 4, 4, 1, False, True)       # linear_insert: step path
    assert not hip.solve_run_supported(512, 1024, 2, 2, 1, False, False)
    assert not hip.solve_run_supported(512, 96, 2, 2, 1, False, False)
    assert not hip.solve_run_supported(512, 128, 2, 2, 9, False, False)


def test_odd_layer_edit_matches_reference_golden():
    from tests.common_checks import check_odd_layer_edit
    report = check_odd_layer_edit(DEV)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(report, open('gpurun_out/solve_parity_odd_layer.json', 'w'))


@pytest.mark.parametrize('name', ['rw_s64_l8_variants', 'rw_s64_l7_variants'])
def test_tiny_and_pre_rewriters_match_reference_golden(name):
    from tests.common_checks import check_rewriter_variants
    report = check_rewriter_variants(DEV, name)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(report, open('gpurun_out/solve_parity_%s.json' % name, 'w'))


def test_key_methods_linear_insert_and_rank3():
    from tests.common_checks import check_extras
    report = check_extras(DEV)
    json.dump(report, open('gpurun_out/solve_parity_extras.json', 'w'))


def test_fast_mconv_equals_seq():
    from tests.common_checks import check_fast_mconv_equals_seq
    report = check_fast_mconv_equals_seq(DEV)
    json.dump(report, open('gpurun_out/fast_mconv_parity.json', 'w'))


def test_odd_layer_solver_against_oracle_autograd():
    """Random upsampling layer: HIP solver vs the oracle's autograd + torch.optim.Adam loop."""
    from rewriting_amd.rewrite import hipsolve
    from oracle import restatement as R
    rs = numpy.random.RandomState(9)
    O, I, h, w = 64, 128, 5, 7
    W0 = torch.from_numpy(rs.randn(1, O, I, 3, 3).astype('float32'))
    key = torch.from_numpy(rs.randn(1, I, h, w).astype('float32'))
    style = torch.from_numpy((1 + 0.3 * rs.randn(1, I)).astype('float32'))
    val = torch.from_numpy(rs.randn(1, O, 2 * h, 2 * w).astype('float32'))
    bias = torch.from_numpy((0.1 * rs.randn(O)).astype('float32'))
    nw = torch.tensor([0.2])
    k4 = R.make_kernel([1, 3, 3, 1]) * 4
    ctx = torch.linalg.qr(torch.from_numpy(rs.randn(I, 2).astype('float32')))[0].t().contiguous()

    def fwd(W):
        out = R.upfirdn2d(R.demod_conv(key, style, W, True), k4, pad=(1, 1))
        out = out + nw * R.noise_rows(1, 4 * h * w).view(1, 1, 2 * h, 2 * w)
        return R.fused_leaky_relu(out, bias)
    for n in (1, 10, 11, 31):
        Wref, lref, _ = R.insert_autograd(W0, fwd, val, ctx, niter=n, piter=10)
        Wd = W0.to(DEV).clone()
        s = hipsolve.run(Wd, key.to(DEV), style.to(DEV), val.to(DEV), bias.to(DEV), nw.to(DEV), ctx.to(DEV),
                         niter=n, piter=10, lr=0.05, blur_kernel=k4.to(DEV))
        r = rel(Wd - W0.to(DEV), Wref - W0)
        assert r < 1e-4, (n, r)
        assert numpy.abs(s.losses.cpu().numpy() - numpy.array(lref)).max() < 1e-5


# ------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize('size,batch', [(256, 4), (1024, 2)])
def test_full_size_generator_properties(monkeypatch, size, batch):
    """At BASELINE sizes the oracle is too slow to run in a test, so: two independent kernel
    paths (fused blocks vs module-by-module) must agree, outputs must be finite, and a seed's
    image must not depend on which other seeds share its batch beyond its noise row."""
    model = build_stylegan(size, 0.5, device=DEV)
    from rewriting_amd.utils import zdataset
    z = zdataset.standard_z_sample(batch, 512, seed=1).to(DEV)
    # (a host sync after every forward: what back-to-back forwards do is pinned by its own test below)
    with torch.no_grad():
        fused = model(z)
        torch.cuda.synchronize()
        monkeypatch.setenv('RW_FUSE', '0')
        plain = model(z)
        torch.cuda.synchronize()
        monkeypatch.setenv('RW_FUSE', '1')
        first = model(z[:1])
        torch.cuda.synchronize()
        monkeypatch.setenv('RW_CONV_PRECISION', 'bf16x6')          # opt-in split-precision stride-1 convolutions
        split = model(z)
        monkeypatch.delenv('RW_CONV_PRECISION')
    assert fused.shape == (batch, 3, size, size) and torch.isfinite(fused).all()
    assert (fused - plain).abs().max().item() < 1e-4 * max(1.0, plain.abs().max().item())
    assert (first - fused[:1]).abs().max().item() < 1e-4 * max(1.0, plain.abs().max().item())   # row 0 noise
    assert (split - fused).abs().max().item() < 1e-4 * max(1.0, plain.abs().max().item())


def test_full_size_statistics_and_solve_properties():
    """1000-seed layer-8 statistics of the 256^2 generator and the full 2001-step solve:
    additivity of the sums, symmetry, trace == checksum from an independent kernel,
    positive semi-definiteness; after the solve dW is rank-1 in the context direction, the
    part of W orthogonal to the context is untouched, and the loss went down."""
    from rewriting_amd import hip
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset, runningstats
    model = build_stylegan(256, 0.5, device=DEV)
    zds = zdataset.z_dataset_for_model(model, size=1000)
    gw = ganrewrite.SeqStyleGanRewriter(model, zds, 8)
    C = gw.c_matrix
    assert tuple(C.shape) == (512, 512) and torch.isfinite(C).all()
    assert (C - C.t()).abs().max().item() == 0.0
    ev = torch.linalg.eigvalsh(C.double().cpu())
    assert ev.min().item() > -1e-6 * ev.max().item()
    # additivity + trace checksum on 3 batches
    halves = [runningstats.RunningSecondMoment() for _ in range(3)]
    tr = 0.0
    with torch.no_grad():
        for bi in range(3):
            zb = torch.stack([zds[i][0] for i in range(bi * 10, bi * 10 + 10)]).to(DEV)
            acts = gw.context_model(zb).fmap
            halves[2].add_nchw(acts)
            halves[bi % 2].add_nchw(acts)
            tr += hip.channel_sums(acts, nchw=True, square_input=False)[1].double().sum().item()
    assert rel(halves[0].mom2 + halves[1].mom2, halves[2].mom2) < 1e-6
    assert abs(halves[2].mom2.diag().double().sum().item() / tr - 1) < 1e-5
    req = load_mask_request('recorded_horse_hat.json')
    W0 = gw.target_weights().detach().clone()
    losses = []
    t_ms = gw.apply_edit(req, rank=1, niter=2001, piter=10, lr=0.05)
    mkey = gw.multi_key_from_selection(req['key'], rank=1)
    W = gw.target_weights().detach()
    dW = W - W0
    assert torch.isfinite(W).all() and dW.abs().max().item() > 0
    assert rel(ganrewrite.projected_conv(dW, mkey), dW) < 1e-4
    sv = torch.linalg.svdvals(dW[0].permute(0, 2, 3, 1).reshape(-1, 512).cpu())
    assert sv[1].item() < 1e-3 * sv[0].item()


def test_micro_batched_forward_equals_one_launch(monkeypatch):
    """SeqStyleGAN2._forward_micro (RW_MICRO_BATCH): slices of the batch through the high-resolution steps give
    the images of the one-launch path, every image with the noise row of its position in the whole batch."""
    from rewriting_amd.utils.stylegan2.models import noise_batch_period
    model = build_stylegan(256, 0.7, device=DEV)
    z = torch.randn(12, 512, generator=torch.Generator().manual_seed(5)).to(DEV)
    with torch.no_grad():
        want = model(z)
        torch.cuda.synchronize()          # (forwards issued back to back: tests/test_gpu_zz_sequences.py)
        with noise_batch_period(3):
            want_p = model(z)
    for spec in ('2:64', '4:128', '1:256', '5:32'):
        monkeypatch.setenv('RW_MICRO_BATCH', spec)
        with torch.no_grad():
            torch.cuda.synchronize()
            got = model(z)
            torch.cuda.synchronize()
            with noise_batch_period(3):
                got_p = model(z)
        # (not bit-equal: the low-resolution kernels pick their split-K by batch size, and the F(4x4,3x3) layers
        # turn a 1e-7 difference of their input into a fresh draw of their own 1e-5 rounding noise)
        assert (got - want).abs().max().item() < 5e-5, spec
        assert (got_p - want_p).abs().max().item() < 5e-5, spec


def test_premultiplied_style_and_one_pass_layers_on_the_image_path(monkeypatch):
    """The un-hooked forward's extra fusions against the same forward without them: the next layer's style multiplied
    into an upsampling layer's result (RW_PRESCALE: the same products, up to the FMA contraction of the in-loop
    multiply, which the F(4x4,3x3) layers turn into a fresh draw of their 1e-5 rounding noise), the last
    upsampling layer in one pass and the last conv + ToRGB by F(4x4,3x3) (RW_UP_FUSED, RW_RGB_F4: the F(4x4,3x3)
    error class)."""
    model = build_stylegan(256, 0.7, device=DEV)
    z = torch.randn(4, 512, generator=torch.Generator().manual_seed(9)).to(DEV)
    with torch.no_grad():
        got = model(z)
        torch.cuda.synchronize()          # (forwards issued back to back: tests/test_gpu_zz_sequences.py)
        monkeypatch.setenv('RW_PRESCALE', '0')
        same = model(z)
        assert (got - same).abs().max().item() < 5e-5
        monkeypatch.setenv('RW_UP_FUSED', '0')
        monkeypatch.setenv('RW_RGB_F4', '0')
        base = model(z)
        torch.cuda.synchronize()
        monkeypatch.setenv('RW_CONV_ALGO', 'winograd')
        exact = model(z)
    assert (got - base).abs().max().item() < 5e-5
    assert (got - exact).abs().max().item() < 1e-4


@pytest.mark.parametrize('hook', [False, True])
def test_insert_on_a_two_layer_target_matches_reference_golden(hook):
    """The autograd path on the kernels: a target of two styled convolutions (and the same with a hooked module),
    weights after 1 and 11 steps and every loss against the reference's own run (rw_s64_l8l9_twolayer)."""
    from tests.common_checks import check_two_layer_target
    report = check_two_layer_target(DEV, hook=hook)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(report, open('gpurun_out/solve_parity_two_layer_hook%d.json' % int(hook), 'w'))


def test_goal_batch_of_two_takes_the_autograd_path_and_matches_the_oracle():
    """A goal of two key/value pairs (the fused solver is batch-1): loss and d W of the first step against
    torch.autograd of oracle/restatement.py's styled convolution on the host."""
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset
    from rewriting_amd.utils.stylegan2.models import DataBag
    from oracle import restatement as R
    model = build_stylegan(32, 0.5, device=DEV)
    zds = zdataset.z_dataset_for_model(model, size=20)
    gw = ganrewrite.SeqStyleGanRewriter(model, zds, 6, cachedir=None)
    with torch.no_grad():
        k = gw.context_model(torch.cat([gw.get_z(3), gw.get_z(5)]))
        v = gw.target_model(k)
    goal_out = DataBag(v, fmap=v.fmap * 1.1 + 0.05)
    mkey = torch.linalg.qr(torch.randn(512, 1, generator=torch.Generator().manual_seed(0)))[0].t().contiguous().to(DEV)
    assert gw._hip_solvable(k) is None
    W0 = gw.target_weights().detach().clone()
    losses = []
    gw.insert(k, goal_out, mkey, niter=1, piter=10, lr=0.05, update_callback=lambda it, l: losses.append(float(l)))
    # host: the same step by autograd over the restatement (Adam's first step moves every entry by lr * sign(grad),
    # then the projection: compare the projected update)
    sd = {n: p.detach().cpu() for n, p in gw.target_model.state_dict().items()}
    w = W0.cpu().clone().requires_grad_(True)
    key, style = k.fmap.cpu(), k.style.cpu()
    y = R.demod_conv(key, style, w, upsample=False)
    b, _, h, wd = y.shape
    y = y + sd['layer6.sconv.noise.weight'] * R.noise_rows(b, h * wd).view(b, 1, h, wd)
    y = R.fused_leaky_relu(y, sd['layer6.sconv.activate.bias'])
    loss = torch.nn.functional.l1_loss(goal_out.fmap.cpu(), y)
    loss.backward()
    assert abs(losses[0] - loss.item()) < 1e-5 * max(1.0, abs(loss.item()))
    step = -0.05 * w.grad / (w.grad.abs() + 1e-8)               # Adam, t = 1: m / (sqrt(v) + eps) = g / (|g| + eps)
    want = R.projected_conv((W0.cpu() + step) - W0.cpu(), mkey.cpu())
    got = (gw.target_weights().detach() - W0).cpu()
    assert ((got - want).norm() / want.norm()).item() < 1e-3, ((got - want).norm() / want.norm()).item()


def test_two_layer_target_gradients_match_the_oracle_at_every_iteration():
    """What IS well defined on the two-layer target whatever the rounding: the loss and d W of the kernels' autograd
    path against torch.autograd over oracle/restatement.py on the host, FROM THE SAME WEIGHTS, at each of the eleven
    iterations of the reference's loop (projection at 0 and 10).  An output within rounding of its goal may flip one
    sign between the two evaluations (an L1 tie: 3e-4 of the gradient when it happens); everything else agrees to 1e-6."""
    from tests.common_checks import two_layer_rewriter_class
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset
    from rewriting_amd.utils.stylegan2.models import DataBag
    from oracle import restatement as R
    g = load_golden('rw_s64_l8l9_twolayer')
    meta = golden_meta(g)
    model = build_stylegan(64, 0.5, device=DEV)
    zds = zdataset.z_dataset_for_model(model, size=meta['nseeds'])
    gw = two_layer_rewriter_class()(model, zds, 8, cachedir=None)
    sd = {k: v.detach().cpu().clone() for k, v in gw.model.state_dict().items()}
    with torch.no_grad():
        bag = gw.context_model(gw.get_z(0))
    dev = lambda a: torch.from_numpy(g[a]).to(DEV)
    gin = DataBag(bag, fmap=dev('goal_in_fmap'), style=dev('goal_in_style'), latent=dev('goal_in_latent'))
    gin.output = bag.output[:, :, :gin.fmap.shape[2], :gin.fmap.shape[3]].contiguous()
    val, mkey = dev('goal_out_fmap'), dev('mkey')
    weight = gw.target_weights()
    opt = torch.optim.Adam([weight], lr=0.05)
    with torch.no_grad():
        ortho = weight - ganrewrite.projected_conv(weight, mkey)
    key_c, style_c, lat_c, val_c = gin.fmap.cpu(), gin.style.cpu(), gin.latent.cpu(), val.cpu()
    lat_idx = list(gw.model.layer9.children())[0].index
    grad_rel, report = [], []
    for it in range(11):
        with torch.enable_grad():
            out = gw.target_model(gin).fmap
            loss = torch.nn.functional.l1_loss(val, out)
            opt.zero_grad()
            loss.backward()
        Wc = weight.detach().cpu().clone().requires_grad_(True)
        y = R.demod_conv(key_c, style_c, Wc, upsample=False)
        b, _, h, w = y.shape
        y = y + sd['layer8.sconv.noise.weight'] * R.noise_rows(b, h * w).view(b, 1, h, w)
        y = R.fused_leaky_relu(y, sd['layer8.sconv.activate.bias'])
        y9, _ = R.styled_conv(sd, 'layer9.sconv', y, lat_c[:, lat_idx], upsample=True)
        lc = torch.nn.functional.l1_loss(val_c, y9)
        lc.backward()
        assert abs(loss.item() - lc.item()) < 2e-6 * max(1.0, abs(lc.item())), (it, loss.item(), lc.item())
        assert rel(out.detach(), y9.detach()) < 5e-6, it
        grad_rel.append(rel(weight.grad.detach(), Wc.grad))
        report.append(dict(it=it, loss=loss.item(), loss_host=lc.item(), grad_rel=grad_rel[-1]))
        opt.step()
        if it % 10 == 0:
            with torch.no_grad():
                weight[...] = ortho + ganrewrite.projected_conv(weight, mkey)
    assert sorted(grad_rel)[len(grad_rel) // 2] < 5e-6 and max(grad_rel) < 5e-3, grad_rel
    assert sum(r > 1e-5 for r in grad_rel) <= 3, grad_rel               # ties are rare events, not the rule
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(report, open('gpurun_out/two_layer_gradients.json', 'w'), indent=1)
