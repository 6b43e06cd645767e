"""Seeded synthetic weights for generators (no checkpoints exist offline).

Every tensor of a state dict is filled from ``numpy.random.RandomState`` seeded by
(crc32(key name), seed), so the same key gets the same values whichever class
(this package's or the reference's, /root/reference/utils/stylegan2/models.py) owns
it and in whatever order parameters were created.  The distributions follow
SURVEY.md section 7.2 item 5 / section 8(d): N(0,1) for convolution weights (the
reference's own default, utils/stylegan2/models.py:303-304,381-383), ``1/lr_mul``
scaled normals for the mapping network (models.py:498-501), modulation bias 1
(models.py:285), a NON-zero noise strength and activation bias so those code paths
are exercised (their defaults are 0: models.py:538, op/fused_act.py:77), and a
non-trivial ``latent_avg`` so truncation does something (models.py:575-581).
"""
import re
import zlib

import numpy
import torch


def _rng(key, seed):
    return numpy.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7fffffff)


def _normal(key, shape, seed, std=1.0, mean=0.0):
    r = _rng(key, seed)
    a = r.standard_normal(int(numpy.prod(shape)) if len(shape) else 1).reshape(shape)
    return torch.from_numpy((a * std + mean).astype('float32'))


def stylegan2_state_dict(reference_state, seed=0):
    """Returns a dict with the keys/shapes of ``reference_state`` filled deterministically."""
    out = {}
    for key, val in reference_state.items():
        shape = tuple(val.shape)
        if key.endswith('.kernel'):            # blur / upsample FIR buffers: keep
            out[key] = val.detach().clone()
        elif key == 'latents.latent_avg':
            out[key] = _normal(key, (512,) if len(shape) == 0 else shape, seed, std=0.1)
        elif key.startswith('noises.'):
            out[key] = val.detach().clone()
        elif re.match(r'style\.\d+\.weight$', key):
            out[key] = _normal(key, shape, seed, std=100.0)     # 1 / lr_mul, lr_mul = 0.01
        elif re.match(r'style\.\d+\.bias$', key):
            out[key] = _normal(key, shape, seed, std=10.0)      # times lr_mul -> N(0, 0.1)
        elif key.endswith('modulation.bias'):
            out[key] = _normal(key, shape, seed, std=0.05, mean=1.0)
        elif key.endswith('noise.weight'):
            out[key] = torch.full(shape, 0.1)
        elif key.endswith('activate.bias') or key.endswith('rgb.bias'):
            out[key] = _normal(key, shape, seed, std=0.1)
        else:                                   # conv / modulation / constant input
            out[key] = _normal(key, shape, seed)
    return out


def proggan_state_dict(reference_state, seed=0):
    """N(0,1) conv weights and zero wscale bias (default init gives a rank-1 covariance
    and a NaN ZCA: SURVEY.md section 7.2 item 5)."""
    out = {}
    for key, val in reference_state.items():
        if key.endswith('wscale.b'):
            out[key] = torch.zeros_like(val)
        else:
            out[key] = _normal(key, tuple(val.shape), seed)
    return out


def randomize_(model, seed=0, kind='stylegan2'):
    fill = stylegan2_state_dict if kind == 'stylegan2' else proggan_state_dict
    sd = fill(model.state_dict(), seed)
    if kind == 'stylegan2' and sd['latents.latent_avg'].shape != model.state_dict()['latents.latent_avg'].shape:
        # latent_avg is registered as a 0-d placeholder (models.py:575); replace the buffer.
        model.latents.latent_avg = sd.pop('latents.latent_avg').to(model.latents.latent_avg.device)
        model.load_state_dict(sd, strict=False)
    else:
        model.load_state_dict(sd)
    return model
