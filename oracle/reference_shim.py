"""TEST INFRASTRUCTURE ONLY -- loader that runs the *reference's own Python files*
from /root/reference on CPU.

This exists only in the build container (``/root/reference`` is absent on the
GPU box).  It is used by ``oracle/make_golden.py`` to generate the committed
fixtures under ``tests/golden/`` and by the CPU tests that pin the travelling
restatement (``oracle/restatement.py``) against the real reference.  Nothing in
the product package imports it.

The reference files are imported UNMODIFIED; only five things that cannot work
in this container are stubbed before import:

1. ``utils.stylegan2.op`` JIT-compiles CUDA at import
   (utils/stylegan2/op/fused_act.py:10-16, op/upfirdn2d.py:9-15).  We install a
   module whose ``upfirdn2d`` is the reference's own pure-torch spec
   ``upfirdn2d_native`` (op/upfirdn2d.py:152-186, extracted from the file with
   ``ast`` because that file forgets to import ``F``) and whose
   ``fused_leaky_relu`` states fused_bias_act_kernel.cu:29-47 (act=3, grad=0).
2. ``torchvision`` is not installed: a stub providing ``transforms.Normalize``
   and ``transforms.functional.to_tensor`` (utils/renormalize.py:7,40,94).
3. ``torch.symeig`` / ``torch.lstsq`` were removed from torch 2.x
   (rewrite/ganrewrite.py:104,822): mapped to ``torch.linalg.eigh(UPLO='U')`` and
   ``torch.linalg.lstsq(driver='gels')`` (symeig's default was upper=True; lstsq was LAPACK gels).
4. ``Tensor.cuda`` / ``Module.cuda`` -> identity for the hard-coded ``.cuda()`` calls
   (utils/stylegan2/models.py:545,652).
5. No checkpoints offline: callers construct the models directly.
"""
import ast
import contextlib
import os
import sys
import types

import numpy
import torch
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get('RW_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'rewrite', 'ganrewrite.py'))


def _extract_function(path, name, namespace):
    """Exec one top-level function of a reference file without importing the file."""
    with open(path) as f:
        tree = ast.parse(f.read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, 'exec'), namespace)
            return namespace[name]
    raise KeyError(name)


def _build_op_module():
    op = types.ModuleType('utils.stylegan2.op')
    native = _extract_function(
        os.path.join(REFERENCE_ROOT, 'utils/stylegan2/op/upfirdn2d.py'),
        'upfirdn2d_native', {'F': F, 'torch': torch})

    def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
        # Same view algebra as op/upfirdn2d.py:87-127,144-149 around the native spec.
        b, c, h, w = input.shape
        out = native(input.reshape(-1, h, w, 1), kernel, up, up, down, down,
                     pad[0], pad[1], pad[0], pad[1])
        return out.view(-1, c, out.shape[1], out.shape[2])

    def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
        # fused_bias_act_kernel.cu:29-47, case 30; bias index (xi / step_b) % size_b.
        shape = [1, -1] + [1] * (input.ndim - 2)
        return F.leaky_relu(input + bias.view(*shape), negative_slope) * scale

    class FusedLeakyReLU(torch.nn.Module):
        def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
            super().__init__()
            self.bias = torch.nn.Parameter(torch.zeros(channel))
            self.negative_slope = negative_slope
            self.scale = scale

        def forward(self, input):
            return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)

    op.upfirdn2d = upfirdn2d
    op.upfirdn2d_native = native
    op.fused_leaky_relu = fused_leaky_relu
    op.FusedLeakyReLU = FusedLeakyReLU
    return op


def _build_torchvision_stub():
    tv = types.ModuleType('torchvision')
    tr = types.ModuleType('torchvision.transforms')
    fn = types.ModuleType('torchvision.transforms.functional')
    md = types.ModuleType('torchvision.models')

    def to_tensor(pic):
        arr = numpy.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(numpy.ascontiguousarray(arr.transpose(2, 0, 1)))
        return t.float().div(255) if t.dtype == torch.uint8 else t

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

    fn.to_tensor = to_tensor
    tr.functional = fn
    tr.Normalize = Normalize
    tr.ToPILImage = object
    tv.transforms = tr
    tv.models = md
    return {'torchvision': tv, 'torchvision.transforms': tr,
            'torchvision.transforms.functional': fn, 'torchvision.models': md}


def _symeig(a, eigenvectors=False, upper=True):
    vals, vecs = torch.linalg.eigh(a, UPLO='U' if upper else 'L')
    return vals, vecs


def _lstsq(b, a):
    # torch 1.x ``torch.lstsq`` called LAPACK gels (QR, full rank assumed).  torch.linalg.lstsq defaults to gelsy,
    # which truncates singular values below eps * max(m, n) * s_max -- for the key statistics of this path
    # (cond ~2e5 in float32) that is a different, rank-truncated answer -- so the driver is pinned.
    return (torch.linalg.lstsq(a, b, driver='gels').solution, None)


_loaded = None


def load():
    """Import the reference packages; returns a namespace with the modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError('reference tree not present at %s' % REFERENCE_ROOT)
    for k in list(sys.modules):
        if k == 'utils' or k.startswith('utils.') or k == 'rewrite' or k.startswith('rewrite.'):
            raise RuntimeError('a module named %s is already imported' % k)
    sys.path.insert(0, REFERENCE_ROOT)
    sys.modules.update(_build_torchvision_stub())
    sys.modules['utils.stylegan2.op'] = _build_op_module()
    torch.symeig = _symeig
    torch.lstsq = _lstsq
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        import utils.stylegan2.models as sg_models
        from utils import nethook, tally, runningstats, zdataset, renormalize, proggan, pbar
        from rewrite import ganrewrite
    finally:
        sys.path.remove(REFERENCE_ROOT)
    ns = types.SimpleNamespace(
        models=sg_models, nethook=nethook, tally=tally, runningstats=runningstats,
        zdataset=zdataset, renormalize=renormalize, proggan=proggan, pbar=pbar,
        ganrewrite=ganrewrite, op=sys.modules['utils.stylegan2.op'])
    _loaded = ns
    return ns


@contextlib.contextmanager
def threads(n):
    old = torch.get_num_threads()
    torch.set_num_threads(n)
    try:
        yield
    finally:
        torch.set_num_threads(old)
