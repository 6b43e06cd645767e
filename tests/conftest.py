import json
import os
import sys

import numpy
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this environment')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return numpy.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def golden_meta(g):
    return json.loads(str(g['meta']))


def load_mask_request(fname, nseeds=None):
    with open(os.path.join(GOLDEN, 'masks', fname)) as f:
        req = json.load(f)
    if nseeds is None:
        return req
    out = {}
    for k, v in req.items():
        out[k] = [[n % nseeds, m] for n, m in v] if k == 'key' else [v[0] % nseeds, v[1]]
    return out


def subsample(t, maxn=4096):
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // maxn)
    return flat[::step].cpu()


@pytest.fixture
def emulated_hip(monkeypatch):
    from tests import hip_emulation
    hip_emulation.install(monkeypatch)
    yield


def build_stylegan(size, truncation, channel_multiplier=2, seed=0, device='cpu', tails='normal'):
    from rewriting_amd.utils.stylegan2 import models
    from rewriting_amd import synthetic
    g = models.SeqStyleGAN2(size, 512, 8, channel_multiplier=channel_multiplier,
                            truncation=truncation, mconv='seq')
    synthetic.randomize_(g, seed=seed, tails=tails)
    return g.eval().to(device)


def oracle_state_dict(g):
    return {k: v.detach().cpu().clone() for k, v in g.state_dict().items()}
