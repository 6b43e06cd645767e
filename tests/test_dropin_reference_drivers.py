"""B1 drop-in claim: after ``rewriting_amd.install_reference_aliases()`` the reference's own
drivers run UNCHANGED on this package -- ``metrics/make_watermark_images.py`` main() (two rewriters sharing one
cache directory, collect_2nd_moment, apply_erase with low_rank_gradient, the gandissect + zero() variant, its
DataLoader / SaveImagePool image writer) and the rewriter construction and sampling loop of
``metrics/sample_edited.py`` (:38-47, :53-61) -- and write the same PNGs as they write on the reference itself.

The reference-side PNGs are committed under tests/golden/dropin/ (made by
``python tests/dropin_driver.py --mode reference --out X`` and copied from X/watermark/*-ours-*/images and
X/samples/dome2spire).  The driver code lives in /root/reference, so this test runs in the build container only
(CPU; the kernels are the torch stand-ins of tests/hip_emulation.py: what is under test is the API surface)."""
import json
import os
import subprocess
import sys

import numpy
import PIL.Image
import pytest

from oracle import reference_shim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = os.path.join(ROOT, 'tests', 'golden', 'dropin')


def _same_pngs(got_dir, want_dir, n):
    names = sorted(f for f in os.listdir(want_dir) if f.endswith('.png'))
    assert len(names) == n
    for f in names:
        a = numpy.asarray(PIL.Image.open(os.path.join(got_dir, f))).astype(int)
        b = numpy.asarray(PIL.Image.open(os.path.join(want_dir, f))).astype(int)
        assert a.shape == b.shape, f
        assert numpy.abs(a - b).max() <= 1, (f, numpy.abs(a - b).max())     # one byte level: 8-bit rounding


@pytest.mark.skipif(not reference_shim.available(), reason='the reference drivers live in /root/reference')
def test_reference_drivers_run_unchanged_and_write_the_same_images(tmp_path):
    out = str(tmp_path / 'dropin')
    env = dict(os.environ, PYTHONPATH=ROOT)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dropin_driver.py'), '--mode', 'ours',
                           '--out', out], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    with open(os.path.join(out, 'report.json')) as f:
        report = json.load(f)
    ours, gandissect = None, None
    for d in report['watermark_dirs']:
        if d.endswith('-ours-30-1'):
            ours = os.path.join(out, 'watermark', d)
        if d.endswith('-gandissect-30'):
            gandissect = os.path.join(out, 'watermark', d)
    assert ours and gandissect
    for d in (ours, gandissect):
        assert os.path.isfile(os.path.join(d, 'done.txt'))                    # pidfile protocol completed
        assert os.path.isfile(os.path.join(d, 'r2m.npz'))                     # statistics cached in the job dir (Q7)
        assert len([f for f in os.listdir(os.path.join(d, 'images')) if f.endswith('.png')]) == 20
    _same_pngs(os.path.join(ours, 'images'), os.path.join(FIXTURES, 'watermark_ours'), 20)
    # (the gandissect variant zeroes units drawn from a NaN tie class and uses the reference's randomised quantile
    # sketch: its choice is arbitrary in the reference itself, DESIGN.md section 2 -- it must run, not match)
    assert report['r2m_cache'] and 'done.txt' in report['sample_edited_files']
    _same_pngs(os.path.join(out, 'samples', 'dome2spire'), os.path.join(FIXTURES, 'sample_edited'), 4)


@pytest.mark.skipif(not reference_shim.available(), reason='needs the reference tree')
def test_aliases_keep_the_reference_ui_modules_importable():
    """rewrite/rewriteapp.py does `from utils import ... show, labwidget, paintwidget, imgviz`: those modules are
    not rebuilt here and must still come from the reference, next to this package's nethook / renormalize."""
    code = (
        "import sys, types; sys.path.insert(0, %r)\n"
        "tv = types.ModuleType('torchvision'); tr = types.ModuleType('torchvision.transforms')\n"
        "fn = types.ModuleType('torchvision.transforms.functional'); tr.functional = fn; tv.transforms = tr\n"
        "tv.models = types.ModuleType('torchvision.models')\n"
        "sys.modules.update({'torchvision': tv, 'torchvision.transforms': tr, 'torchvision.transforms.functional': fn,"
        " 'torchvision.models': tv.models})\n"
        "import rewriting_amd; rewriting_amd.install_reference_aliases(reference_root=%r)\n"
        "from utils import nethook, renormalize, pbar, tally, zdataset, labwidget, paintwidget, pidfile, imgsave\n"
        "import utils.stylegan2.models as m, rewrite.ganrewrite as g\n"
        "assert nethook.__name__ == 'rewriting_amd.utils.nethook' and g.__name__ == 'rewriting_amd.rewrite.ganrewrite'\n"
        "assert m.__name__ == 'rewriting_amd.utils.stylegan2.models'\n"
        "assert labwidget.__file__.startswith(%r) and pidfile.__file__.startswith(%r)\n"
        "print('ok')\n" % (ROOT, reference_shim.REFERENCE_ROOT, reference_shim.REFERENCE_ROOT,
                           reference_shim.REFERENCE_ROOT))
    proc = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 and proc.stdout.strip().endswith('ok'), proc.stdout + proc.stderr[-3000:]
