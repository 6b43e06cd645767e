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


def _heavy(key, shape, seed):
    """Convolution / modulation weights as trained checkpoints have them: Student-t (nu = 3) entries -- a few per
    thousand are 5 - 20 sigma -- times per-out-channel and per-in-channel log-normal gains (sigma 0.5), scaled back to
    unit RMS.  Activations downstream are heavy-tailed too: the fixture for tolerances that scale with the output
    range (the F(4x4,3x3) error class, the f16 operand pairs of the split kernels)."""
    r = _rng(key + '/heavy', seed)
    n = int(numpy.prod(shape))
    a = r.standard_t(3, size=n).reshape(shape) / numpy.sqrt(3.0)          # variance of t(3) is 3
    if len(shape) >= 4:                                                     # (1, O, I, k, k) or (O, I, k, k)
        o_ax, i_ax = len(shape) - 4, len(shape) - 3
        go = numpy.exp(0.5 * r.standard_normal(shape[o_ax])).reshape([-1 if d == o_ax else 1 for d in range(len(shape))])
        gi = numpy.exp(0.5 * r.standard_normal(shape[i_ax])).reshape([-1 if d == i_ax else 1 for d in range(len(shape))])
        a = a * go * gi
    elif len(shape) == 2:                                                   # modulation: (in_channel, style_dim)
        a = a * numpy.exp(0.5 * r.standard_normal(shape[0]))[:, None]
    a = a / numpy.sqrt((a * a).mean())
    return torch.from_numpy(a.astype('float32'))


def stylegan2_state_dict(reference_state, seed=0, tails='normal'):
    """Returns a dict with the keys/shapes of ``reference_state`` filled deterministically.  tails='heavy': the
    convolution and modulation weights from _heavy instead of N(0, 1)."""
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
        elif tails == 'heavy' and (key.endswith('dconv.weight') or key.endswith('modulation.weight')
                                   or key.endswith('conv.weight')):
            out[key] = _heavy(key, shape, seed)
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


def randomize_(model, seed=0, kind='stylegan2', tails='normal'):
    if kind == 'stylegan2':
        sd = stylegan2_state_dict(model.state_dict(), seed, tails=tails)
    else:
        sd = proggan_state_dict(model.state_dict(), seed)
    if kind == 'stylegan2' and sd['latents.latent_avg'].shape != model.state_dict()['latents.latent_avg'].shape:
        # latent_avg is registered as a 0-d placeholder (models.py:575); replace the buffer.
        model.latents.latent_avg = sd.pop('latents.latent_avg').to(model.latents.latent_avg.device)
        model.load_state_dict(sd, strict=False)
    else:
        model.load_state_dict(sd)
    return model
