"""rewriting_amd -- the rule-editing hot path of davidbau/rewriting, built MI355X-first.

Layout (only what the path needs):

    csrc/                 hand-written gfx950 kernels + the C ABI (include/rewriting_hip.h)
    _lib.py, hip.py       ctypes binding and tensor-level wrappers (no fallback path)
    utils/                mirror of the reference's ``utils`` names on the path: nethook, tally,
                          runningstats, zdataset, renormalize, stylegan2 (models + op), proggan
    rewrite/ganrewrite.py mirror of rewrite/ganrewrite.py (``*Rewriter`` classes)
    parallel.py           one-process-per-GPU sharding of the sweeps over RCCL/xGMI
    synthetic.py          seeded synthetic weights (no checkpoints offline)

``install_reference_aliases()`` registers this package's ``utils`` and ``rewrite`` under the
reference's top-level module names so ``rewrite/rewriteapp.py``, the notebooks and
``metrics/*.py`` import it unchanged.
"""
import sys

__version__ = '0.1.0'


_ALIASED = ('nethook', 'tally', 'runningstats', 'zdataset', 'renormalize', 'pbar', 'sampler', 'proggan', 'imgviz',
            'stylegan2', 'stylegan2.models', 'stylegan2.op')


def install_reference_aliases(force=False, reference_root=None):
    """Makes ``from utils import nethook, zdataset``, ``from utils.stylegan2 import load_seq_stylegan`` and
    ``from rewrite import ganrewrite`` resolve to this package -- and ONLY the modules this package implements.

    The reference's ``utils`` and ``rewrite`` are namespace packages (no ``__init__.py``).  They are registered
    here as packages whose ``__path__`` lists this package's directory FIRST and, when ``reference_root`` (or
    the environment variable RW_REFERENCE_ROOT) names a checkout of the reference, that checkout's directory
    second: ``utils.show``, ``utils.labwidget``, ``utils.paintwidget``, ``utils.pidfile``,
    ``utils.imgsave``, ``utils.workerpool``, ``utils.segmenter`` and ``rewrite.rewriteapp`` -- consumers of the
    boundary that this package does not rebuild -- keep importing from the reference, on top of the kernels
    here.  The modules of the path itself are bound explicitly so that the reference's same-named files can
    never shadow them."""
    import importlib
    import os
    import types
    here = os.path.dirname(os.path.abspath(__file__))
    root = reference_root or os.environ.get('RW_REFERENCE_ROOT')
    for pkg in ('utils', 'rewrite'):
        if pkg in sys.modules and not force:
            continue
        mod = types.ModuleType(pkg)
        mod.__path__ = [os.path.join(here, pkg)]
        if root and os.path.isdir(os.path.join(root, pkg)):
            mod.__path__.append(os.path.join(root, pkg))
        mod.__package__ = pkg
        sys.modules[pkg] = mod
    names = {'utils.' + n: 'rewriting_amd.utils.' + n for n in _ALIASED}
    names['rewrite.ganrewrite'] = 'rewriting_amd.rewrite.ganrewrite'
    for alias, real in names.items():
        if alias in sys.modules and not force:
            continue
        module = importlib.import_module(real)
        sys.modules[alias] = module
        parent, _, leaf = alias.rpartition('.')
        setattr(sys.modules[parent], leaf, module)
    if root and root not in sys.path:
        sys.path.append(root)            # `metrics`, `notebooks` ... : untouched reference packages
