"""Progress/verbosity shim with the call surface of utils/pbar.py that the hot path uses:
``pbar(iterable)``, ``pbar.quiet()``, ``pbar.print``, ``pbar.descnext`` (:136-199).  The module
object itself is callable, as in the reference."""
import contextlib
import sys
import types

try:
    from tqdm import tqdm as _tqdm
except Exception:          # pragma: no cover
    _tqdm = None

_state = {'verbose': False, 'desc': None}


def descnext(desc):
    _state['desc'] = desc


def print(*args):      # noqa: A001
    if _state['verbose']:
        sys.stdout.write(' '.join(str(a) for a in args) + '\n')


@contextlib.contextmanager
def _verbosity(flag):
    old = _state['verbose']
    _state['verbose'] = flag
    try:
        yield
    finally:
        _state['verbose'] = old


def quiet():
    return _verbosity(False)


def verbose():
    return _verbosity(True)


def _wrap(iterable, desc=None, **kwargs):
    desc = desc or _state['desc']
    _state['desc'] = None
    if _state['verbose'] and _tqdm is not None:
        return _tqdm(iterable, desc=desc, **kwargs)
    return iterable


class _CallableModule(types.ModuleType):
    def __call__(self, iterable, **kwargs):
        return _wrap(iterable, **kwargs)


sys.modules[__name__].__class__ = _CallableModule
