"""SURVEY.md 8f row 2, rendering half: ``render_image`` / ``render_image_batch`` / ``render_object`` with heat-map
and mask overlays (rewrite/ganrewrite.py:596-650, called from rewrite/rewriteapp.py:136,168,190,270,447) through
this package's own ``utils/imgviz.py``, against the reference's classes executed on CPU."""
import numpy
import pytest
import torch

from oracle import reference_shim
from tests.conftest import build_stylegan, load_mask_request

needs_reference = pytest.mark.skipif(not reference_shim.available(), reason='compares with /root/reference')


@needs_reference
def test_image_visualizer_equals_the_reference():
    ref = reference_shim.load()
    theirs = ref.ganrewrite.imgviz
    from rewriting_amd.utils import imgviz as mine
    g = torch.Generator().manual_seed(0)
    img = torch.randn(1, 3, 64, 64, generator=g).clamp(-1, 1)
    act = torch.randn(16, 16, generator=g)
    for size in (64, (48, 80), 128):
        a, b = theirs.ImageVisualizer(size), mine.ImageVisualizer(size)
        for kw in (dict(level=0.3), dict(level=0.0, border_color=[255, 0, 0], thickness=3), dict(percent_level=0.9),
                   dict(level=0.5, inside_color=[0, 255, 0], outside_bright=0.25)):
            assert numpy.array_equal(numpy.asarray(a.masked_image(img, act, **kw)),
                                     numpy.asarray(b.masked_image(img, act, **kw))), (size, kw)
        shape = (size, size) if isinstance(size, int) else size
        m = torch.rand(*shape, generator=g) > 0.5
        assert numpy.array_equal(numpy.asarray(a.masked_image(img, mask=m)), numpy.asarray(b.masked_image(img, mask=m)))
        for mode in ('bilinear', 'nearest'):
            assert numpy.array_equal(numpy.asarray(a.heatmap(act, mode=mode)), numpy.asarray(b.heatmap(act, mode=mode)))
        assert numpy.array_equal(numpy.asarray(a.image(img)), numpy.asarray(b.image(img)))


@needs_reference
def test_rewriter_rendering_matches_the_reference(emulated_hip):
    """The UI's rendering calls on a 64^2 generator: plain image, key heat-map overlay at a level, mask overlay,
    object box overlay and the batch forms -- pixel for pixel (one byte level: 8-bit rounding of fp32 images)."""
    from oracle import reference_shim as shim
    ref = shim.load()
    from rewriting_amd import synthetic
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset
    nseeds, layer = 60, 8
    mine_model = build_stylegan(64, 0.5)
    their_model = ref.models.SeqStyleGAN2(64, 512, 8, truncation=0.5, mconv='seq')
    synthetic.randomize_(their_model, seed=0)
    their_model.eval()
    gm = ganrewrite.SeqStyleGanRewriter(mine_model, zdataset.z_dataset_for_model(mine_model, size=nseeds), layer)
    gt = ref.ganrewrite.SeqStyleGanRewriter(their_model, ref.zdataset.z_dataset_for_model(their_model, size=nseeds),
                                            layer, cachedir=None)
    req = load_mask_request('recorded_horse_hat.json', nseeds)

    def same(a, b):
        a, b = numpy.asarray(a).astype(int), numpy.asarray(b).astype(int)
        assert a.shape == b.shape
        # overlays threshold a heat map: a pixel exactly on the level may flip; everything else within a byte level
        assert (numpy.abs(a - b) > 1).mean() < 2e-3, (numpy.abs(a - b) > 1).mean()
    same(gm.render_image(3), gt.render_image(3))
    key = gt.query_key_from_selection(*req['key'][0])           # the same key on both sides
    same(gm.render_image(5, key=key, level=0.2), gt.render_image(5, key=key, level=0.2))
    mask = torch.zeros(64, 64, dtype=torch.bool)
    mask[10:30, 20:50] = True
    same(gm.render_image(7, mask=mask), gt.render_image(7, mask=mask))
    for x, y in zip(gm.render_image_batch([1, 2, 4, 8]), gt.render_image_batch([1, 2, 4, 8])):
        same(x, y)
    for x, y in zip(gm.render_image_batch([1, 2], key=key, level=0.1), gt.render_image_batch([1, 2], key=key, level=0.1)):
        same(x, y)
    obj_m = gm.object_from_selection(*req['object'])
    obj_t = gt.object_from_selection(*req['object'])
    same(gm.render_object(obj_m[1], box=obj_m[3]), gt.render_object(obj_t[1], box=obj_t[3]))
    same(gm.render_object(obj_m[1]), gt.render_object(obj_t[1]))
