"""Progress/verbosity shim with the call surface of utils/pbar.py that the hot path and its drivers use:
``pbar(iterable)``, ``pbar.quiet()``, ``pbar.verbose()``, ``pbar.print``, ``pbar.descnext``, ``pbar.post``,
``pbar.desc`` and ``pbar.reporthook`` (utils/pbar.py:24-199; metrics/make_watermark_images.py:63-64 drives its
solver callback through a reporthook).  The module object itself is callable, as in the reference."""
import contextlib
import sys
import types

try:
    from tqdm import tqdm as _tqdm
except Exception:          # pragma: no cover
    _tqdm = None

_state = {'verbose': False, 'desc': None, 'bar': None}


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


def post(**kwargs):
    """Status text on the innermost visible bar (utils/pbar.py:24-32); nothing to show when quiet."""
    bar = _state.get('bar')
    if bar is not None and hasattr(bar, 'set_postfix'):
        bar.set_postfix(**kwargs)


def desc(text):
    bar = _state.get('bar')
    if bar is not None and hasattr(bar, 'set_description'):
        bar.set_description(str(text))


class _ReportHook:
    """utils/pbar.py:105-133: hook(b, bsize, tsize) moves the bar to b * bsize."""

    def __init__(self, bar):
        self.t = bar

    def __call__(self, b=1, bsize=1, tsize=None):
        if self.t is None:
            return
        if tsize is not None:
            self.t.total = tsize
        self.t.update(b * bsize - self.t.n)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.t is not None:
            self.t.close()
            if _state.get('bar') is self.t:
                _state['bar'] = None


def reporthook(*args, **kwargs):
    bar = None
    if _state['verbose'] and _tqdm is not None:
        kw = dict(unit_scale=True, miniters=1)
        kw.update(kwargs)
        bar = _state['bar'] = _tqdm(None, *args, **kw)
    return _ReportHook(bar)


def _wrap(iterable, desc=None, **kwargs):
    desc = desc or _state['desc']
    _state['desc'] = None
    if _state['verbose'] and _tqdm is not None:
        bar = _state['bar'] = _tqdm(iterable, desc=desc, **kwargs)
        return bar
    return iterable


class _CallableModule(types.ModuleType):
    def __call__(self, iterable, **kwargs):
        return _wrap(iterable, **kwargs)


sys.modules[__name__].__class__ = _CallableModule
