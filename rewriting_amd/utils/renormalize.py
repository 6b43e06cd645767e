"""Image <-> tensor normalisation and mask decoding (host side, PIL).

Same call surface as utils/renormalize.py (``from_url`` :45-50, ``from_image`` :35-42,
``as_image`` :15-19, ``as_url`` :22-32, ``renormalizer`` :53-82) without the torchvision
dependency.  ``from_url(mask, 'pt', size)[0]`` is how the rewriter turns a painted mask into
feature-resolution weights: base64 PNG -> RGB -> PIL *bilinear* resize -> red channel in [0,1]
(quirk Q4), so PIL stays the resampler.
"""
import base64
import io
import re

import numpy
import PIL.Image
import torch

OFFSET_SCALE = dict(
    pt=([0.0, 0.0, 0.0], [1.0, 1.0, 1.0]),
    zc=([0.5, 0.5, 0.5], [0.5, 0.5, 0.5]),
    imagenet=([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]),
    imagenet_meanonly=([0.485, 0.456, 0.406], [1.0 / 255, 1.0 / 255, 1.0 / 255]),
    places_meanonly=([0.475, 0.441, 0.408], [1.0 / 255, 1.0 / 255, 1.0 / 255]),
    byte=([0.0, 0.0, 0.0], [1.0 / 255, 1.0 / 255, 1.0 / 255]))


def _to_tensor(im):
    """PIL image -> float CHW tensor in [0,1] (what torchvision's to_tensor returns for uint8)."""
    arr = numpy.asarray(im)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(numpy.ascontiguousarray(arr.transpose(2, 0, 1)))
    return t.float().div(255) if t.dtype == torch.uint8 else t.float()


class Renormalizer:
    def __init__(self, oldoffset, oldscale, newoffset, newscale, tobyte=False):
        old_o, old_s = numpy.array(oldoffset), numpy.array(oldscale)
        new_o, new_s = numpy.array(newoffset), numpy.array(newscale)
        self.mul = torch.from_numpy(old_s / new_s)
        self.add = torch.from_numpy((old_o - new_o) / new_s)
        self.tobyte = tobyte
        self.mean, self.std = newoffset, newscale

    def __call__(self, data):
        shape = (-1, 1, 1) if data.dim() == 3 else (1, -1, 1, 1)
        mul = self.mul.to(data.device, data.dtype).view(shape)
        add = self.add.to(data.device, data.dtype).view(shape)
        out = data * mul + add
        return out.clamp(0, 255).byte() if self.tobyte else out


def renormalizer(source='zc', target='zc'):
    if isinstance(source, str):
        oldoffset, oldscale = OFFSET_SCALE[source]
    else:
        norm = find_normalizer(source)
        oldoffset, oldscale = (norm.mean, norm.std) if norm is not None else OFFSET_SCALE['pt']
    newoffset, newscale = target if isinstance(target, tuple) else OFFSET_SCALE[target]
    return Renormalizer(oldoffset, oldscale, newoffset, newscale, tobyte=(target == 'byte'))


def find_normalizer(source=None):
    if source is None:
        return None
    if isinstance(source, Renormalizer) or (hasattr(source, 'mean') and hasattr(source, 'std')):
        return source
    inner = getattr(source, 'transform', None)
    if inner is not None:
        return find_normalizer(inner)
    for t in reversed(getattr(source, 'transforms', None) or []):
        found = find_normalizer(t)
        if found is not None:
            return found
    return None


def as_tensor(data, source='zc', target='zc'):
    return renormalizer(source=source, target=target)(data)


def as_image(data, source='zc', target='byte'):
    assert data.dim() == 3
    return PIL.Image.fromarray(renormalizer(source=source, target=target)(data)
                               .permute(1, 2, 0).cpu().numpy())


def as_url(data, source='zc', size=None):
    img = data if isinstance(data, PIL.Image.Image) else as_image(data, source)
    if size is not None:
        img = img.resize(size, resample=PIL.Image.BILINEAR)
    buf = io.BytesIO()
    img.save(buf, format='png')
    return 'data:image/png;base64,%s' % base64.b64encode(buf.getvalue()).decode('utf-8')


def from_image(im, target='zc', size=None):
    if im.format != 'RGB':
        im = im.convert('RGB')
    if size is not None:
        im = im.resize(tuple(size), resample=PIL.Image.BILINEAR)
    return renormalizer(source='pt', target=target)(_to_tensor(im))


def from_url(url, target='zc', size=None):
    data = re.sub('^data:image/.+;base64,', '', url)
    im = PIL.Image.open(io.BytesIO(base64.b64decode(data)))
    if target == 'image' and size is None:
        return im
    return from_image(im, target, size=size)
