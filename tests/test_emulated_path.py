"""Host logic above the C ABI, exercised on CPU with the kernel wrappers replaced by the
torch stand-ins of tests/hip_emulation.py: module tree, fused vs. per-module execution, hooks,
rewriter orchestration and the solver driver, all against the reference-generated fixtures."""
import copy

import numpy
import pytest
import torch

from tests.conftest import (build_stylegan, golden_meta, load_golden, load_mask_request, subsample)


def _stage_outputs(model, z):
    store, handles = {}, []
    for name, mod in model.named_modules():
        if name and len(list(mod.children())) == 0:
            handles.append(mod.register_forward_hook(
                lambda m, i, o, name=name: store.__setitem__(name, o)))
    with torch.no_grad():
        img = model(z)
    for h in handles:
        h.remove()
    return img, store


@pytest.mark.parametrize('name', ['gen_s32_t05', 'gen_s64_cm1'])
def test_generator_matches_golden_fused_and_unfused(emulated_hip, monkeypatch, name):
    g = load_golden(name)
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'])
    z = torch.from_numpy(g['z'])
    want = torch.from_numpy(g['image'])
    with torch.no_grad():
        fused = model(z)
    assert (fused - want).abs().max() < 1e-4
    monkeypatch.setenv('RW_FUSE', '0')
    img, store = _stage_outputs(model, z)
    assert (img - want).abs().max() < 1e-4
    checked = 0
    for key in g.files:
        if not (key.startswith('stage/') and key.endswith('/sub')):
            continue
        lname = key[6:-4]
        if lname not in store:
            continue
        out = store[lname]
        if isinstance(out, dict):
            field = 'fmap'
            if lname.startswith('up_rgb'):
                field = 'output'
            if lname.endswith('modulation'):
                field = 'style'
            if lname.startswith('style.') or lname == 'latents':
                field = 'latent'
            if field not in out:
                continue
            out = out[field]
        w = torch.from_numpy(g[key])
        got = subsample(out)
        assert got.shape == w.shape, lname
        assert (got - w).abs().max() < 5e-5 * max(1.0, w.abs().max().item()), lname
        checked += 1
    assert checked >= 40, checked


def test_hooks_disable_fusion_and_see_reference_values(emulated_hip):
    from rewriting_amd.utils import nethook
    g = load_golden('gen_s32_t05')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'])
    z = torch.from_numpy(g['z'])
    with nethook.InstrumentedModel(model) as inst:
        inst.retain_layer('layer6.sconv.mconv.adain', detach=False)
        with torch.no_grad():
            img = inst(z)
        key = inst.retained_layer('layer6.sconv.mconv.adain').fmap
    assert (subsample(key) - torch.from_numpy(g['stage/layer6.sconv.mconv.adain/sub'])).abs().max() < 1e-4
    assert (img - torch.from_numpy(g['image'])).abs().max() < 1e-4
    assert 'forward' not in model.layer6.sconv.mconv.adain.__dict__     # unhooked on close


def _rewriter(meta, **kw):
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset
    model = build_stylegan(meta['size'], meta['truncation'])
    zds = zdataset.z_dataset_for_model(model, size=meta['nseeds'])
    return ganrewrite.SeqStyleGanRewriter(model, zds, meta['layernum'], cachedir=None,
                                          low_rank_insert=True, key_method='zca', tight_paste=True, **kw), model


def test_rewriter_edit_matches_golden(emulated_hip):
    from rewriting_amd.rewrite import ganrewrite
    g = load_golden('rw_s64_l8_horsehat')
    meta = golden_meta(g)
    gw, model = _rewriter(meta)
    assert list(gw.k_shape) == list(g['k_shape']) and list(gw.v_shape) == list(g['v_shape'])
    assert list(gw.x_shape) == list(g['x_shape'])
    assert abs(gw.c_matrix.double().norm().item() / float(g['c_matrix_norm']) - 1) < 1e-5
    assert (gw.c_matrix[::4, ::4] - torch.from_numpy(g['c_matrix'])).abs().max() < 1e-4 * gw.c_matrix.abs().max()
    assert abs(gw.zca_matrix.double().norm().item() / float(g['zca_norm']) - 1) < 2e-3
    req = load_mask_request(meta['mask'], meta['nseeds'])
    o_imgnum, o_mask = req['object']
    p_imgnum, p_mask = req['paste']
    obj_acts, _, obj_area, bounds = gw.object_from_selection(o_imgnum, o_mask)
    assert list(bounds) == list(g['obj_bounds'])
    goal_in, goal_out, _, pb = gw.paste_from_selection(p_imgnum, p_mask, obj_acts, obj_area)
    assert list(pb) == list(g['paste_bounds'])
    assert (goal_in.fmap - torch.from_numpy(g['goal_in_fmap'])).abs().max() < 1e-4
    assert (goal_out.fmap - torch.from_numpy(g['goal_out_fmap'])).abs().max() < 1e-4
    assert (goal_in.style - torch.from_numpy(g['goal_in_style'])).abs().max() < 1e-5
    assert list(goal_out.output.shape) == list(g['goal_out_output_shape'])
    mkey = gw.multi_key_from_selection(req['key'], rank=1)
    assert ganrewrite.all_obs.shape[0] == int(g['n_sel'])
    assert (mkey - torch.from_numpy(g['mkey'])).abs().max() < 2e-3
    # isolate the solve from upstream rounding (L1 + Adam amplifies 1e-6 input noise chaotically)
    mkey = torch.from_numpy(g['mkey'])
    goal_in = type(goal_in)(goal_in, fmap=torch.from_numpy(g['goal_in_fmap']),
                            style=torch.from_numpy(g['goal_in_style']))
    goal_out = type(goal_out)(goal_out, fmap=torch.from_numpy(g['goal_out_fmap']))
    W0 = gw.target_weights().detach().clone()
    assert abs(W0.double().norm().item() / float(g['W0_norm']) - 1) < 1e-6
    seen = []
    for niter in (1, 11):
        gwn, _ = _rewriter(meta)
        gwn.insert(goal_in, goal_out, mkey, niter=niter, piter=10, lr=0.05)
        dW = (gwn.target_weights().detach() - W0)[0]
        cos = torch.einsum('oiyx,di->odyx', dW, mkey)
        rel = (cos - torch.from_numpy(g['dW_%d_cos' % niter])).norm() / float(g['dW_%d_norm' % niter])
        assert rel < 1e-4, (niter, rel)
        # rank-1 in the context direction: dW == P(dW)
        assert (dW - ganrewrite.projected_conv(dW[None], mkey)[0]).norm() / dW.norm() < 1e-4
    gwn, _ = _rewriter(meta)
    gwn.insert(goal_in, goal_out, mkey, niter=101, piter=10, lr=0.05,
               update_callback=lambda it, loss: seen.append((it, float(loss))))
    assert [it for it, _ in seen] == list(range(101))
    assert numpy.abs(numpy.array([l for _, l in seen]) - g['losses'])[:20].max() < 1e-5
    assert numpy.abs(numpy.array([l for _, l in seen]) - g['losses']).max() < 5e-4
    dW = (gwn.target_weights().detach() - W0)[0]
    rel = (torch.einsum('oiyx,di->odyx', dW, mkey) - torch.from_numpy(g['dW_101_cos'])).norm() / float(g['dW_101_norm'])
    assert rel < 2e-4, rel
    with torch.no_grad():
        zs = torch.cat([gwn.get_z(i) for i in (0, 1)])
        img = gwn.sample_image_from_latent(zs)
    assert (img - torch.from_numpy(g['edited_image'])).abs().max() < 2e-3
    # the caller's model is untouched (deepcopy at construction), the rewriter's copy is edited
    assert torch.equal(model.layer8.sconv.mconv.dconv.weight, W0)


def test_rewriter_erase_matches_golden(emulated_hip):
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset
    g = load_golden('rw_s64_l6_erase')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'])
    zds = zdataset.z_dataset_for_model(model, size=meta['nseeds'])
    gw = ganrewrite.SeqStyleGanRewriter(model, zds, meta['layernum'], low_rank_insert=True,
                                        low_rank_gradient=True)
    req = load_mask_request(meta['mask'], meta['nseeds'])
    p_imgnum, p_mask = req['paste']
    with torch.no_grad():
        scale = gw.square_scales_for_units()
        assert (scale - torch.from_numpy(g['unit_scale'])).abs().max() < 1e-4 * float(g['unit_scale'].max())
        units = gw.normdissect_units(req['key'], meta['drank'])
        assert set(units.tolist()) == set(g['d_units'].tolist())
        goal_in, goal_out = gw.erase_from_selection(p_imgnum, p_mask, req['key'], meta['drank'])
    assert list(goal_in.fmap.shape) == list(g['goal_in_fmap_shape'])       # quirk Q8: full map
    assert (goal_in.fmap - torch.from_numpy(g['goal_in_fmap'])).abs().max() < 1e-4
    assert (goal_out.fmap - torch.from_numpy(g['goal_out_fmap'])).abs().max() < 1e-4
    mkey = torch.from_numpy(g['mkey'])
    goal_in = type(goal_in)(goal_in, fmap=torch.from_numpy(g['goal_in_fmap']),
                            style=torch.from_numpy(g['goal_in_style']))
    goal_out = type(goal_out)(goal_out, fmap=torch.from_numpy(g['goal_out_fmap']))
    W0 = gw.target_weights().detach().clone()
    gw.insert(goal_in, goal_out, mkey, niter=11, piter=10, lr=0.05)
    dW = (gw.target_weights().detach() - W0)[0]
    rel = (torch.einsum('oiyx,di->odyx', dW, mkey) - torch.from_numpy(g['dW_11_cos'])).norm() / float(g['dW_11_norm'])
    assert rel < 2e-4, rel


def test_zero_and_projection_helpers(emulated_hip):
    from rewriting_amd.rewrite import ganrewrite
    torch.manual_seed(0)
    w = torch.randn(1, 8, 6, 3, 3)
    d = torch.linalg.qr(torch.randn(6, 2))[0].t().contiguous()
    p = ganrewrite.projected_conv(w, d)
    assert (ganrewrite.projected_conv(p, d) - p).abs().max() < 1e-5     # idempotent
    assert (ganrewrite.rank_one_conv(w[0], d[0]) - ganrewrite.projected_conv(w[0], d[:1])).abs().max() < 1e-5


def test_odd_layer_edit_matches_golden(emulated_hip):
    from tests.common_checks import check_odd_layer_edit
    check_odd_layer_edit('cpu')


@pytest.mark.parametrize('name', ['rw_s64_l8_variants', 'rw_s64_l7_variants'])
def test_tiny_and_pre_rewriters_match_golden(emulated_hip, name):
    from tests.common_checks import check_rewriter_variants
    check_rewriter_variants('cpu', name)


def test_key_methods_linear_insert_and_rank3(emulated_hip):
    from tests.common_checks import check_extras
    check_extras('cpu')


def test_fast_mconv_equals_seq(emulated_hip):
    from tests.common_checks import check_fast_mconv_equals_seq
    check_fast_mconv_equals_seq('cpu')


def test_generator_with_winograd_convolutions_holds_the_golden_bars(emulated_hip, monkeypatch):
    """RW_CONV_ALGO=winograd routes the eligible stride-1 convolutions (16^2, 32^2 and 64^2 maps of this generator)
    through hip.conv3x3_wino -- here the same F(2x2,3x3) arithmetic in torch fp32 -- and the reference golden
    still holds at the bars of the direct path (images 1e-4, stages 5e-5 relative)."""
    from rewriting_amd import hip
    g = load_golden('gen_s64_cm1')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'])
    z = torch.from_numpy(g['z'])
    calls = []
    orig = hip.conv3x3_wino
    monkeypatch.setattr(hip, 'conv3x3_wino', lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1])
    monkeypatch.setenv('RW_CONV_ALGO', 'winograd')
    with torch.no_grad():
        fused = model(z)
    assert sorted(s[-1] for s in calls) == [4, 8, 16, 32, 64]     # layers 2, 4, 6, 8 and 10 (every stride-1 layer)
    assert (fused - torch.from_numpy(g['image'])).abs().max() < 1e-4
    monkeypatch.setenv('RW_FUSE', '0')
    img, store = _stage_outputs(model, z)
    assert (img - torch.from_numpy(g['image'])).abs().max() < 1e-4
    for lname in ('layer6.sconv.mconv.dconv', 'layer8.sconv.mconv.dconv', 'layer10.sconv.mconv.dconv'):
        w = torch.from_numpy(g['stage/%s/sub' % lname])
        assert (subsample(store[lname].fmap) - w).abs().max() < 5e-5 * max(1.0, w.abs().max().item()), lname


def test_micro_batched_forward_equals_one_launch(emulated_hip, monkeypatch):
    """SeqStyleGAN2._forward_micro: high-resolution steps in slices of the batch -- same images, every image with
    the noise row of its position in the whole batch (quirk Q1), also inside noise_batch_period."""
    from rewriting_amd.utils.stylegan2.models import noise_batch_period
    model = build_stylegan(64, 0.7)
    z = torch.randn(6, 512, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want = model(z)
        with noise_batch_period(3):
            want_p = model(z)
    for spec in ('2:16', '4:32', '1:64'):
        monkeypatch.setenv('RW_MICRO_BATCH', spec)
        with torch.no_grad():
            got = model(z)
            with noise_batch_period(3):
                got_p = model(z)
        # (torch's CPU convolutions round differently per batch size: the stand-in kernels, not the path)
        assert (got - want).abs().max() < 3e-5, spec
        assert (got_p - want_p).abs().max() < 3e-5, spec
    monkeypatch.setenv('RW_MICRO_BATCH', '8:16')                 # not larger than the batch: the plain path
    with torch.no_grad():
        assert torch.equal(model(z), want)


def test_generator_with_f4_winograd_is_inside_the_image_tolerance(emulated_hip, monkeypatch):
    """RW_CONV_ALGO=winograd4 (opt-in F(4x4,3x3)): same generator, reference golden, north_star's image tolerance
    1e-3 L-inf (measured ~1e-5 at this size); the default algorithm holds 1e-4."""
    g = load_golden('gen_s64_cm1')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'])
    z = torch.from_numpy(g['z'])
    want = torch.from_numpy(g['image'])
    monkeypatch.setenv('RW_CONV_ALGO', 'winograd4')
    with torch.no_grad():
        got = model(z)
    err = (got - want).abs().max().item()
    assert err < 1e-3, err
    monkeypatch.setenv('RW_CONV_ALGO', 'winograd')
    with torch.no_grad():
        base = model(z)
    assert (base - want).abs().max().item() < 1e-4 and not torch.equal(base, got)


@pytest.mark.parametrize('direct16', ['0', 'auto'])       # F(4x4,3x3) everywhere / the default: direct sums where they are faster
def test_one_pass_upsampling_layers_stay_inside_the_image_tolerance(emulated_hip, monkeypatch, direct16):
    """RW_UP_ALGO=winograd4: the upsampling StyledConvs as ONE pass (transposed conv (*) blur as four phase convolutions
    -- F(4x4,3x3), or, with RW_MM_DIRECT16, direct sums on the 16-bit pipe where the shape allows --, noise + bias + activation in
    the epilogue) instead of conv -> (2H+1)^2 map -> blur pass: same generator, reference golden, image tolerance; a
    hooked layer falls back to the module-by-module route."""
    from rewriting_amd import hip
    from rewriting_amd.utils import nethook
    g = load_golden('gen_s64_cm1')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'])
    z = torch.from_numpy(g['z'])
    want = torch.from_numpy(g['image'])
    calls, direct = [], []
    real, real16 = hip.conv_transpose3x3s2_blur_wino4, hip.conv_transpose3x3s2_blur_direct16
    monkeypatch.setattr(hip, 'conv_transpose_blur_wino4_supported', lambda o, i, h, w: True)
    monkeypatch.setattr(hip, 'conv_transpose3x3s2_blur_wino4', lambda *a, **k: (calls.append(a[0].shape), real(*a, **k))[1])
    monkeypatch.setattr(hip, 'conv_transpose3x3s2_blur_direct16',
                        lambda *a, **k: (calls.append(a[0].shape), direct.append(a[0].shape), real16(*a, **k))[2])
    monkeypatch.setenv('RW_UP_ALGO', 'winograd4')
    monkeypatch.setenv('RW_MM_DIRECT16', direct16)
    monkeypatch.setenv('RW_UP_FUSED2', '0')             # (the default takes the 32^2 layer in rw_tconv.hip's kernel instead)
    with torch.no_grad():
        got = model(z)
    assert len(calls) == 4                              # 4 -> 8 -> 16 -> 32 -> 64
    assert [tuple(sh[2:]) for sh in direct] == ([] if direct16 == '0' else [(32, 32)])       # w % 32 == 0, h % 8 == 0
    assert (got - want).abs().max().item() < 1e-3
    monkeypatch.setenv('RW_UP_ALGO', 'winograd')
    del calls[:]
    with torch.no_grad():
        base = model(z)
    assert not calls and (base - want).abs().max().item() < 1e-4
    assert (got - base).abs().max().item() < 1e-4 and not torch.equal(got, base)
    # a hooked blur inside one layer: that layer runs child by child, the others in one pass
    monkeypatch.setenv('RW_UP_ALGO', 'winograd4')
    with nethook.InstrumentedModel(model) as inst:
        inst.retain_layer('layer5.sconv.mconv.blur', detach=False)
        with torch.no_grad():
            hooked = inst(z)
        assert inst.retained_layer('layer5.sconv.mconv.blur') is not None
    assert len(calls) == 3
    assert (hooked - want).abs().max().item() < 1e-3


def test_style_handed_over_premultiplied_changes_nothing(emulated_hip, monkeypatch):
    """Inside the un-hooked forward an upsampling layer multiplies the NEXT layer's style into its own result where
    that layer runs F(4x4,3x3) (one multiply per element instead of one per 6x6 item and out-channel tile): the same
    products in the same precision, so the image is bit-identical with RW_PRESCALE=0, and bags seen by hooks never
    carry the hand-over key."""
    from rewriting_amd import hip
    from rewriting_amd.utils import nethook
    g = load_golden('gen_s64_cm1')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'])
    z = torch.from_numpy(g['z'])
    seen = []
    real = hip.blur_noise_act
    monkeypatch.setattr(hip, 'blur_noise_act',
                        lambda *a, **k: (seen.append(k.get('post_scale') is not None), real(*a, **k))[1])
    monkeypatch.setenv('RW_MM', 'f32')                   # the fp32 kernels: bit for bit
    with torch.no_grad():
        got = model(z)
    assert any(seen)                                     # the 64 x 64 layer takes F(4x4,3x3)
    monkeypatch.setenv('RW_PRESCALE', '0')
    del seen[:]
    with torch.no_grad():
        base = model(z)
    assert not any(seen) and torch.equal(got, base)
    monkeypatch.delenv('RW_PRESCALE')
    # the split-operand kernels (the default of this forward) scale by a power of two taken from the bound on their input:
    # max |x s| when the style came pre-multiplied, max |x| max |s| otherwise -- a binade apart at most, i.e. other low
    # bits of the f16 pairs
    monkeypatch.delenv('RW_MM')
    with torch.no_grad():
        got_split = model(z)
    monkeypatch.setenv('RW_PRESCALE', '0')
    with torch.no_grad():
        base_split = model(z)
    monkeypatch.delenv('RW_PRESCALE')
    assert (got_split - base_split).abs().max().item() < 2e-6 * got.abs().max().item()
    assert (got_split - got).abs().max().item() < 1e-5 * got.abs().max().item()
    with nethook.InstrumentedModel(model) as inst:
        inst.retain_layer('layer7', detach=False)
        with torch.no_grad():
            hooked = inst(z)
        assert 'prescaled' not in inst.retained_layer('layer7')
    assert (hooked - got).abs().max().item() < 1e-4


def test_successor_pairs_and_hand_over_guards(emulated_hip):
    """SeqStyleGAN2._successors pairs every upsampling layer with the styled convolution that follows it directly,
    and a pre-scaled map can only be consumed by a fused stride-1 layer: anything else refuses it loudly."""
    from rewriting_amd.utils.stylegan2 import models
    g = load_golden('gen_s64_cm1')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'])
    succ = model._successors()
    ups = [m.sconv for n, m in model.named_children() if n.startswith('layer') and getattr(m, 'sconv', None) is not None
           and m.sconv.mconv.upsample]
    assert len(ups) == 4 and set(succ) == {id(u) for u in ups}
    for name, idx in (('layer3', 2), ('layer5', 4), ('layer7', 6), ('layer9', 8)):
        nxt, lat = succ[id(getattr(model, name).sconv)]
        assert nxt is getattr(model, 'layer%d' % (int(name[5:]) + 1)).sconv and lat == idx
    # a bag carrying the hand-over key in front of an upsampling layer, or of a layer that runs module by module
    z = torch.from_numpy(g['z'])
    with torch.no_grad():
        bag = model.bag_from_z(z)
        for n, m in model.named_children():
            if n == 'layer5':
                break
            if n != 'bag_in':
                bag = m(bag)
        bad = models.DataBag(bag, prescaled=torch.ones(z.shape[0], 512))
        with pytest.raises(RuntimeError, match='pre-scaled'):
            model.layer5(bad)                            # upsampling
        from rewriting_amd.utils import nethook
        with nethook.InstrumentedModel(model) as inst:
            inst.retain_layer('layer4.sconv.mconv.adain', detach=False)
            with pytest.raises(RuntimeError, match='pre-scaled'):
                model.layer4(models.DataBag(bag, prescaled=torch.ones(z.shape[0], 512)))


@pytest.mark.parametrize('hook', [False, True])
def test_insert_on_a_two_layer_target_runs_through_autograd(emulated_hip, hook):
    """The host side of the autograd path (utils/stylegan2/grad.py: which gradients are formed how, the module-by-
    module switch of a fully covered StyledConvSeq under grad mode, the rewriter's fall-through) against the
    reference's own run of the same two-layer target; the kernels are exercised by the -m gpu twin."""
    from tests.common_checks import check_two_layer_target
    check_two_layer_target('cpu', hook=hook)


def test_styled_conv_modules_are_differentiable(emulated_hip):
    """d fmap, d weight (both terms: the demodulation factor is part of the graph, quirk Q3) and d style of
    DemodulatedConv2dF / ApplyStyle / NoiseInjectionF against torch.autograd of the oracle restatement."""
    from rewriting_amd.utils.stylegan2 import models
    from oracle import restatement as R
    torch.manual_seed(0)
    for up in (False, True):
        m = models.DemodulatedConv2dF(8, 6, 3, upsample=up)
        x = torch.randn(2, 8, 5, 7, requires_grad=True)
        st = (1 + 0.3 * torch.randn(2, 8)).requires_grad_(True)
        y = m(models.DataBag(fmap=x, style=st)).fmap
        # the graph runs through grad.py's Function (the emulated kernels are torch ops: were they left attached, the
        # adjoints under test would never be called)
        assert type(y.grad_fn).__name__.startswith('DemodConv')
        g = torch.randn_like(y)
        (y * g).sum().backward()
        x2, st2, w2 = (t.detach().clone().requires_grad_(True) for t in (x, st, m.weight))
        (R.demod_conv(x2, st2, w2, upsample=up) * g).sum().backward()
        for got, want in ((x.grad, x2.grad), (m.weight.grad, w2.grad), (st.grad, st2.grad)):
            assert (got - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
    x = torch.randn(2, 4, 3, 5, requires_grad=True)
    st = torch.randn(2, 4, requires_grad=True)
    y = models.ApplyStyle()(models.DataBag(fmap=x, style=st)).fmap
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    assert torch.allclose(x.grad, gy * st.detach()[:, :, None, None], atol=1e-6)
    assert torch.allclose(st.grad, (gy * x.detach()).sum((2, 3)), atol=1e-5)
    inj = models.NoiseInjectionF()
    inj.weight.data.fill_(0.3)
    x = torch.randn(2, 4, 3, 5, requires_grad=True)
    y = inj(models.DataBag(fmap=x)).fmap
    (y * gy).sum().backward()
    noise = models.reference_noise(2, 15, x.device).reshape(2, 1, 3, 5)
    assert torch.equal(x.grad, gy) and torch.allclose(inj.weight.grad, (gy * noise).sum().reshape(1), atol=1e-5)


def test_bounds_are_produced_only_where_the_next_layer_reads_them(emulated_hip, monkeypatch):
    """Round 5: a fused layer asks its kernel for the bound of its result (hip.new_bound + one reduction launch on the
    device) only when the layer that reads the map runs a split-operand kernel; the bound rides on the tensor, never in
    the bag; nobody measures a map (hip.absmax) whose producer already described it; a hooked model produces none."""
    from rewriting_amd import hip
    from rewriting_amd.utils import nethook
    from rewriting_amd.utils.stylegan2 import models
    g = load_golden('gen_s64_cm1')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'])
    z = torch.from_numpy(g['z'])
    made, measured = [], []
    real_new, real_abs = hip.new_bound, hip.absmax
    monkeypatch.setattr(hip, 'new_bound', lambda n, dev: (made.append(n), real_new(n, dev))[1])
    monkeypatch.setattr(hip, 'absmax', lambda x: (measured.append(tuple(x.shape)), real_abs(x))[1])
    readers = model._readers()
    names = {id(getattr(m, 'sconv', None) or getattr(m, 'conv', None)): n for n, m in model.named_children()}
    chain = [(names[k], names[id(v)]) for k, v in readers.items()]
    assert chain[0] == ('layer2', 'layer3') and chain[-1] == ('layer9', 'layer10') and len(chain) == 8
    bags = []
    hook = model.layer9.register_forward_hook(lambda m, i, o: bags.append(o))      # a torch hook: the model stays "un-hooked"
    with torch.no_grad():
        img = model(z)
    hook.remove()
    b = z.shape[0]
    # 64-model: layer 10 (64^2, F(4x4,3x3) + ToRGB) reads the bound of layer 9's result (the blur pass reports it); layer 9
    # (32^2 -> 64^2, F(2,2) split) reads layer 8's -- since round 6 the 32^2 stride-1 layer is a direct sum on the 16-bit
    # pipe (runs_small_direct16), which reports its result's bound and reads the one layer 7's blur pass leaves; the
    # layers in front ask for none; the last layer has no reader; nobody measures a map
    n32 = b * model.channels[32] * 32 * 32
    assert made == [n32, n32, b * model.channels[64] * 64 * 64], made
    assert measured == [], measured
    monkeypatch.setenv('RW_DIRECT16_SMALL', '0')      # round 5's route: layer 8 on the fp32 F(2x2,3x3) kernel, which reports nothing
    del made[:]
    with torch.no_grad():
        model(z)
    assert made == [b * model.channels[64] * 64 * 64] and measured == [(b, model.channels[32], 32, 32)], (made, measured)
    monkeypatch.delenv('RW_DIRECT16_SMALL')
    del made[:], measured[:], bags[:]
    hook = model.layer9.register_forward_hook(lambda m, i, o: bags.append(o))
    with torch.no_grad():
        img = model(z)
    hook.remove()
    assert all('amax' not in bag for bag in bags)
    fm = bags[0].fmap
    assert isinstance(getattr(fm, 'rw_amax', None), tuple) and fm.rw_amax[0].numel() >= hip.bound_floats(fm.numel())
    assert hip.bound_value(fm.rw_amax[0]) == fm.abs().max().item()
    want = torch.from_numpy(g['image'])
    assert (img - want).abs().max().item() < 1e-3
    del made[:], measured[:]
    with nethook.InstrumentedModel(model) as inst:
        inst.retain_layer('layer7', detach=False)
        with torch.no_grad():
            inst(z)
    # hooked: nobody hands a bound over; from 32^2 up the convolutions run on the 16-bit pipe there too (layer 8: direct sums,
    # layer 9: the fused upsampling kernel, layer 10: direct sums) and each measures its input itself, everything below
    # multiplies in fp32
    assert not made and measured == [(b, model.channels[32], 32, 32)] * 2 + [(b, model.channels[64], 64, 64)], measured
    monkeypatch.setenv('RW_DIRECT16_HOOKED', '0')
    del measured[:]
    with nethook.InstrumentedModel(model) as inst:
        inst.retain_layer('layer7', detach=False)
        with torch.no_grad():
            inst(z)
    assert not made and measured == [(b, model.channels[32], 32, 32)]         # layer 9 alone
    monkeypatch.delenv('RW_DIRECT16_HOOKED')
    monkeypatch.setenv('RW_MM_HOOKED', 'f32')
    del measured[:]
    with nethook.InstrumentedModel(model) as inst:
        inst.retain_layer('layer7', detach=False)
        with torch.no_grad():
            inst(z)
    assert not made and not measured
    assert not models._rgb_branch.reader and not models._rgb_branch.image_path


def test_fused_transposed_conv_and_blur_layer_stays_inside_the_image_tolerance(emulated_hip, monkeypatch):
    """DemodulatedConv2dF.fused_upsample: inside the un-hooked forward an upsampling StyledConv whose shape
    hip.tconv_blur_supported takes runs as ONE launch of hip.conv_transpose3x3s2_blur_fused (the transposed convolution at
    its own multiply count, the blur from LDS) with the layer's plain direct-16 packing; same generator, reference
    golden, image tolerance; RW_UP_FUSED2=0 and a layer above RW_UP_FUSED2_MAX_IN channels never take it; hooked models do
    unless a hook sits inside the layer."""
    from rewriting_amd import hip
    from rewriting_amd.utils import nethook
    g = load_golden('gen_s64_cm1')
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'])
    z = torch.from_numpy(g['z'])
    want = torch.from_numpy(g['image'])
    calls = []
    real = hip.conv_transpose3x3s2_blur_fused
    monkeypatch.setattr(hip, 'conv_transpose3x3s2_blur_fused', lambda *a, **k: (calls.append((a[0].shape, sorted(k))), real(*a, **k))[1])
    monkeypatch.setenv('RW_UP_FUSED2_MAX_IN', '256')
    with torch.no_grad():
        model(z)
    assert not calls                                                    # layer 9 has 512 input channels
    monkeypatch.delenv('RW_UP_FUSED2_MAX_IN')
    with torch.no_grad():
        got = model(z)                                                  # the default (models.DemodulatedConv2dF.fused_upsample)
    assert [tuple(sh[2:]) for sh, _ in calls] == [(32, 32)]             # layer 9: 32^2 -> 64^2 (h % 16 == 0, w % 32 == 0)
    assert 'y_amax' in calls[0][1] and 'post_scale' in calls[0][1]      # its reader (layer 10, F(4x4,3x3)) takes bound and style
    assert (got - want).abs().max().item() < 1e-3
    del calls[:]
    monkeypatch.setenv('RW_UP_FUSED2', '0')
    with torch.no_grad():
        base = model(z)
    assert not calls and (got - base).abs().max().item() < 1e-4
    monkeypatch.delenv('RW_UP_FUSED2')
    # a hooked model (a hook on the layer's OUTPUT, as the sweeps and the rewriters set them) takes it too, where its F(2,2)
    # transposed convolutions run in the split form; not with RW_UP_FUSED2_HOOKED=0, not with RW_MM_HOOKED=f32, and not a
    # layer with a hook INSIDE it (that one runs child by child)
    with nethook.InstrumentedModel(model) as inst:
        inst.retain_layer('layer9', detach=False)
        with torch.no_grad():
            hooked = inst(z)
        assert [tuple(sh[2:]) for sh, _ in calls] == [(32, 32)] and 'y_amax' not in calls[0][1]
        assert (hooked - got).abs().max().item() < 1e-4
        del calls[:]
        for name in ('RW_UP_FUSED2_HOOKED', 'RW_MM_HOOKED'):
            monkeypatch.setenv(name, '0' if name == 'RW_UP_FUSED2_HOOKED' else 'f32')
            with torch.no_grad():
                other = inst(z)
            monkeypatch.delenv(name)
            assert not calls and (other - got).abs().max().item() < 1e-4
    with nethook.InstrumentedModel(model) as inst:
        inst.retain_layer('layer9.sconv.mconv.blur', detach=False)
        with torch.no_grad():
            inside = inst(z)
    assert not calls and (inside - got).abs().max().item() < 1e-4
