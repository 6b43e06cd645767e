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


def install_reference_aliases(force=False):
    """Makes ``import utils.nethook``, ``from utils.stylegan2 import load_seq_stylegan`` and
    ``from rewrite import ganrewrite`` resolve to this package."""
    import importlib
    names = {
        'utils': 'rewriting_amd.utils',
        'utils.nethook': 'rewriting_amd.utils.nethook',
        'utils.tally': 'rewriting_amd.utils.tally',
        'utils.runningstats': 'rewriting_amd.utils.runningstats',
        'utils.zdataset': 'rewriting_amd.utils.zdataset',
        'utils.renormalize': 'rewriting_amd.utils.renormalize',
        'utils.pbar': 'rewriting_amd.utils.pbar',
        'utils.sampler': 'rewriting_amd.utils.sampler',
        'utils.proggan': 'rewriting_amd.utils.proggan',
        'utils.stylegan2': 'rewriting_amd.utils.stylegan2',
        'utils.stylegan2.models': 'rewriting_amd.utils.stylegan2.models',
        'utils.stylegan2.op': 'rewriting_amd.utils.stylegan2.op',
        'rewrite': 'rewriting_amd.rewrite',
        'rewrite.ganrewrite': 'rewriting_amd.rewrite.ganrewrite',
    }
    for alias, real in names.items():
        if alias in sys.modules and not force:
            continue
        sys.modules[alias] = importlib.import_module(real)
