"""Model surgery: hooking layers of a torch model and slicing nested Sequentials.

This is the hook surface the rewriter and the notebook UI are written against
(reference: utils/nethook.py -- ``InstrumentedModel`` :16-281, ``subsequence`` :322-401,
``set_requires_grad`` :404-412).  The public names, arguments and observable behaviour are
kept; the implementation is independent:

* a layer is hooked by planting a bound ``forward`` on the INSTANCE (not a torch forward
  hook), so un-hooking restores the class method and ``'forward' in layer.__dict__`` tells the
  kernels' fusion logic that someone is watching that layer;
* ``subsequence`` treats first/after as one start marker and last/upto as one stop marker
  and walks the nested Sequentials once.
"""
import copy
import inspect
import types
from collections import OrderedDict, defaultdict

import numpy
import torch


class InstrumentedModel(torch.nn.Module):
    """Wraps a model so that named layers can be retained (their outputs recorded) or edited
    (their outputs replaced/ablated) on every forward pass.

        with InstrumentedModel(model) as inst:
            inst.retain_layer('layer4')
            inst.edit_layer('layer4', ablation=0.5, replacement=feats)
            inst(z)
            acts = inst.retained_layer('layer4')
    """

    def __init__(self, model):
        super().__init__()
        self.model = model
        self._retained = OrderedDict()
        self._detach_retained = {}
        self._editargs = defaultdict(dict)
        self._editrule = {}
        self._hooked_layer = {}     # aka -> layer name
        self._old_forward = {}      # layer name -> (module, aka, previous instance forward or None)
        if isinstance(model, torch.nn.Sequential):
            self._hook_sequential()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close()

    def forward(self, *inputs, **kwargs):
        return self.model(*inputs, **kwargs)

    def layer_names(self):
        return [name for name, _ in self.model.named_modules()]

    # -- retaining ---------------------------------------------------------------------------
    @staticmethod
    def _split_aka(layername):
        if isinstance(layername, str):
            return layername, layername
        name, aka = layername
        return name, aka

    def retain_layer(self, layername, detach=True):
        self.retain_layers([layername], detach=detach)

    def retain_layers(self, layernames, detach=True):
        self.add_hooks(layernames)
        for entry in layernames:
            _, aka = self._split_aka(entry)
            if aka not in self._retained:
                self._retained[aka] = None
                self._detach_retained[aka] = detach

    def stop_retaining_layers(self, layernames):
        self.add_hooks(layernames)
        for entry in layernames:
            _, aka = self._split_aka(entry)
            if aka in self._retained:
                del self._retained[aka]
                del self._detach_retained[aka]

    def retained_features(self, clear=False):
        result = OrderedDict(self._retained)
        if clear:
            for k in result:
                self._retained[k] = None
        return result

    def retained_layer(self, aka=None, clear=False):
        if aka is None:
            aka = next(iter(self._retained))
        value = self._retained[aka]
        if clear:
            self._retained[aka] = None
        return value

    # -- editing -----------------------------------------------------------------------------
    def edit_layer(self, layername, rule=None, **kwargs):
        """output = x * (1 - ablation) + replacement * ablation by default; ``rule`` may be any
        callable (x, imodel, **kwargs)."""
        name, aka = self._split_aka(layername)
        self.add_hooks([(name, aka)])
        self._editargs[aka].update(kwargs)
        self._editrule[aka] = apply_ablation_replacement if rule is None else rule

    def remove_edits(self, layername=None):
        if layername is None:
            self._editargs.clear()
            self._editrule.clear()
            return
        _, aka = self._split_aka(layername)
        self._editargs.pop(aka, None)
        self._editrule.pop(aka, None)

    # -- hooking -----------------------------------------------------------------------------
    def add_hooks(self, layernames):
        wanted = {}
        for entry in layernames:
            name, aka = self._split_aka(entry)
            if self._hooked_layer.get(aka, None) != name:
                wanted[name] = aka
        if not wanted:
            return
        for name, layer in self.model.named_modules():
            if name in wanted:
                self._hook_layer(layer, name, wanted.pop(name))
        for name in wanted:
            raise ValueError('Layer %s not found in model' % name)

    def _hook_layer(self, layer, layername, aka):
        if aka in self._hooked_layer:
            raise ValueError('Layer %s already hooked' % aka)
        if layername in self._old_forward:
            raise ValueError('Layer %s already hooked' % layername)
        self._hooked_layer[aka] = layername
        self._old_forward[layername] = (layer, aka, layer.__dict__.get('forward', None))
        inner = layer.forward
        owner = self

        def hooked_forward(this, *inputs, **kwargs):
            return owner._postprocess_forward(inner(*inputs, **kwargs), aka)
        layer.forward = types.MethodType(hooked_forward, layer)

    def _unhook_layer(self, aka):
        if aka not in self._hooked_layer:
            return
        layername = self._hooked_layer.pop(aka)
        if aka in self._retained:
            del self._retained[aka]
            del self._detach_retained[aka]
        self.remove_edits(aka)
        layer, check, previous = self._old_forward.pop(layername)
        assert check == aka
        if previous is None:
            layer.__dict__.pop('forward', None)
        else:
            layer.forward = previous

    def _postprocess_forward(self, x, aka):
        if aka in self._retained:
            self._retained[aka] = x.detach() if self._detach_retained[aka] else x
        rule = self._editrule.get(aka, None)
        if rule is not None:
            x = invoke_with_optional_args(rule, x, self, name=aka, **(self._editargs[aka]))
        return x

    def _hook_sequential(self):
        """The wrapped Sequential accepts layer= / first_layer= / last_layer= to run a slice."""
        model = self.model
        self._hooked_layer['.'] = '.'
        self._old_forward['.'] = (model, '.', model.__dict__.get('forward', None))

        def sliced_forward(this, x, layer=None, first_layer=None, last_layer=None):
            assert layer is None or (first_layer is None and last_layer is None)
            if layer is not None:
                first_layer = last_layer = layer
            first = None if first_layer is None else str(first_layer)
            last = None if last_layer is None else str(last_layer)
            running = first is None
            for name, child in this._modules.items():
                if name == first:
                    first, running = None, True
                if running:
                    x = child(x)
                if name == last:
                    last, running = None, False
            assert first is None, '%s not found' % first
            assert last is None, '%s not found' % last
            return x
        model.forward = types.MethodType(sliced_forward, model)

    def close(self):
        for aka in list(self._hooked_layer.keys()):
            self._unhook_layer(aka)
        assert len(self._old_forward) == 0


def apply_ablation_replacement(x, imodel, **buffers):
    a = make_matching_tensor(buffers, 'ablation', x)
    if a is not None:
        x = x * (1 - a)
        v = make_matching_tensor(buffers, 'replacement', x)
        if v is not None:
            x += (v * a)
    return x


def make_matching_tensor(valuedict, name, data):
    """valuedict[name] as a tensor of data's dtype/device, unsqueezed to data's rank (a
    (C,) vector becomes (1, C, 1, 1) for NCHW data); the converted tensor is cached back."""
    v = valuedict.get(name, None)
    if v is None:
        return None
    if not isinstance(v, torch.Tensor):
        v = torch.from_numpy(numpy.array(v))
        valuedict[name] = v
    if v.device != data.device or v.dtype != data.dtype:
        assert not v.requires_grad, '%s wrong device or type' % name
        v = v.to(device=data.device, dtype=data.dtype)
        valuedict[name] = v
    if v.dim() < data.dim():
        assert not v.requires_grad, '%s wrong dimensions' % name
        v = v.view((1,) + tuple(v.shape) + (1,) * (data.dim() - v.dim() - 1))
        valuedict[name] = v
    return v


# -- slicing nested Sequentials ---------------------------------------------------------------

class _Marker:
    """A dotted layer name used as the start (first: inclusive / after: exclusive) or the stop
    (last: inclusive / upto: exclusive) of a subsequence."""

    def __init__(self, dotted, inclusive):
        self.path = dotted.split('.')
        self.inclusive = inclusive

    def at(self, depth, name):
        return self.path[depth] == name

    def deeper(self, depth):
        return len(self.path) > depth + 1

    def __str__(self):
        return '.'.join(self.path)


def subsequence(sequential, first_layer=None, last_layer=None, after_layer=None, upto_layer=None,
                single_layer=None, share_weights=False):
    """A new Sequential made of the children of ``sequential`` from first_layer to last_layer
    (inclusive) or between after_layer and upto_layer (exclusive).  Dotted names descend into
    nested Sequentials; partially covered containers are rebuilt as plain ``nn.Sequential``s of
    the covered children.  share_weights=True references the original modules (the rewriter
    relies on this: rewrite/ganrewrite.py:48-58); otherwise they are deep-copied."""
    assert single_layer is None or (first_layer is last_layer is after_layer is upto_layer is None)
    if single_layer is not None:
        first_layer = last_layer = single_layer
    assert first_layer is None or after_layer is None
    assert last_layer is None or upto_layer is None
    start = (_Marker(first_layer, True) if first_layer is not None else
             _Marker(after_layer, False) if after_layer is not None else None)
    stop = (_Marker(last_layer, True) if last_layer is not None else
            _Marker(upto_layer, False) if upto_layer is not None else None)
    return _slice(sequential, start, stop, share_weights, 0)


def hierarchical_subsequence(sequential, first, last, after, upto, share_weights=False, depth=0):
    """Compatibility entry point taking pre-split name lists (utils/nethook.py:347-401)."""
    join = (lambda p: None if p is None else '.'.join(p))
    assert depth == 0
    return subsequence(sequential, first_layer=join(first), last_layer=join(last),
                       after_layer=join(after), upto_layer=join(upto), share_weights=share_weights)


def _slice(module, start, stop, share_weights, depth):
    if start is None and stop is None:
        return module if share_weights else copy.deepcopy(module)
    if not isinstance(module, torch.nn.Sequential):
        raise AssertionError('%s not Sequential' % ('.'.join((start or stop).path[:depth]) or 'arg'))
    chosen = OrderedDict()
    running = start is None
    for name, child in module._modules.items():
        child_start = child_stop = None
        begin_after = end_after = False
        if start is not None and start.at(depth, name):
            if start.deeper(depth):
                running, child_start = True, start
            elif start.inclusive:
                running = True
            else:
                begin_after = True
            start = None
        if stop is not None and stop.at(depth, name):
            if stop.deeper(depth):
                child_stop, end_after = stop, True
            elif stop.inclusive:
                end_after = True
            else:
                running = False
            stop = None
        if running:
            part = _slice(child, child_start, child_stop, share_weights, depth + 1)
            if part is not None:
                chosen[name] = part
        if end_after:
            running = False
        if begin_after:
            running = True
    for marker in (start, stop):
        if marker is not None:
            raise ValueError('Layer %s not found' % marker)
    if not chosen and depth > 0:
        return None
    return torch.nn.Sequential(chosen)


def set_requires_grad(requires_grad, *models):
    for model in models:
        if isinstance(model, torch.nn.Module):
            for param in model.parameters():
                param.requires_grad = requires_grad
        elif isinstance(model, (torch.nn.Parameter, torch.Tensor)):
            model.requires_grad = requires_grad
        else:
            assert False, 'unknown type %r' % type(model)


def invoke_with_optional_args(fn, *args, **kwargs):
    """Calls fn with only the positional/keyword arguments its signature accepts."""
    spec = inspect.getfullargspec(fn)
    taken = 0
    if spec.varkw is None:
        taken = len([k for k in kwargs if k in spec.args])
        kwargs = {k: v for k, v in kwargs.items()
                  if k in spec.args or (spec.kwonlyargs and k in spec.kwonlyargs)}
    if spec.varargs is None:
        args = args[:len(spec.args) - taken]
    return fn(*args, **kwargs)
