"""Heat-map / mask overlays used by the rewriter's ``render_image``, ``render_object`` and
``render_image_batch`` (rewrite/ganrewrite.py:596-640) and by the notebook UI through them
(rewrite/rewriteapp.py:136,168,190,270,447).

The part of utils/imgviz.py those calls reach: ``ImageVisualizer(size).masked_image / heatmap / image``
(:56-122,160-185,309-330) with the default centred up-sampling grid of utils/upsample.py:124-156.
It is host-side rendering of one image at a time (PIL output), outside the hot path: plain torch ops."""
import PIL.Image
import torch

from . import renormalize


def _upsample(a, size, mode='bilinear'):
    """utils/upsample.py:5-43,124-156 with the default scale/offset: feature pixel centres spread evenly over the
    target, grid_sample(align_corners=True), zeros outside."""
    dh, dw = a.shape[-2:]
    th, tw = size

    def axis(ts, ds):
        s = float(ts) / ds
        o = 0.5 * s - 0.5
        return (torch.arange(ts, dtype=torch.float32, device=a.device) - o) * (2 / (s * max(1, ds - 1))) - 1
    ty, tx = axis(th, dh), axis(tw, dw)
    grid = torch.stack((tx[None, :].expand(th, tw), ty[:, None].expand(th, tw)), 2)[None]
    return torch.nn.functional.grid_sample(a[None, None].float(), grid, mode=mode, padding_mode='zeros',
                                           align_corners=True)[0, 0]


def border_from_mask(mask, thickness=1, outside=True):
    """Pixels where the boolean mask changes between 8-neighbours, grown `thickness` times (utils/imgviz.py:309-330)."""
    a = mask
    out = torch.zeros_like(a)
    for it in range(thickness):
        h = a[:-1, :] != a[1:, :]
        v = a[:, :-1] != a[:, 1:]
        d = a[:-1, :-1] != a[1:, 1:]
        u = a[1:, :-1] != a[:-1, 1:]
        out[:-1, :-1] |= d
        out[1:, 1:] |= d
        out[1:, :-1] |= u
        out[:-1, 1:] |= u
        out[:-1, :] |= h
        out[1:, :] |= h
        out[:, :-1] |= v
        out[:, 1:] |= v
        if it > 0:
            out |= a
        a = out
    if outside:
        out &= ~mask
    return out


class ImageVisualizer:
    def __init__(self, size, renormalizer=None, level=None, percent_level=None):
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.renormalizer = renormalizer
        self.level = level
        self.percent_level = percent_level

    def pytorch_image(self, imagedata):
        if imagedata.dim() == 4:
            imagedata = imagedata[0]
        r = self.renormalizer or renormalize.renormalizer('zc', 'byte')
        return torch.nn.functional.interpolate(r(imagedata).float()[None], size=self.size)[0]

    def image(self, imagedata):
        return PIL.Image.fromarray(self.pytorch_image(imagedata).permute(1, 2, 0).byte().cpu().numpy())

    def level_for(self, activations, unit, percent_level=None):
        if unit is not None and self.level is not None:
            return self.level[unit[1] if hasattr(unit, '__len__') else unit].item()
        s, _ = activations.reshape(-1).sort()
        p = percent_level if percent_level is not None else (self.percent_level or 0.95)
        return s[int(len(s) * p)]

    def pytorch_mask(self, activations, unit, level=None, percent_level=None):
        a = activations if unit is None else activations[unit]
        if level is None:
            level = self.level_for(activations, unit, percent_level=percent_level)
        return _upsample(a, self.size) > level

    def pytorch_masked_image(self, imagedata, activations=None, unit=None, level=None, percent_level=None,
                             thickness=1, mask=None, border_color=None, outside_bright=0.5, inside_color=None):
        scaled = self.pytorch_image(imagedata).float().cpu()
        if mask is None:
            mask = self.pytorch_mask(activations, unit, level=level, percent_level=percent_level).cpu()
        border = border_from_mask(mask, thickness)
        inside = (mask & ~border).float()
        outside = (~mask & ~border).float()
        border = border.float()
        color = torch.tensor([255.0, 255.0, 0] if border_color is None else border_color,
                             dtype=torch.float32)[:, None, None]
        fill = scaled if inside_color is None else torch.tensor(inside_color, dtype=torch.float32)[:, None, None]
        return (fill * inside + color * border + outside_bright * scaled * outside).clamp(0, 255).byte()

    def masked_image(self, imagedata, activations=None, unit=None, level=None, percent_level=None, **kwargs):
        img = self.pytorch_masked_image(imagedata, activations=activations, unit=unit, level=level,
                                        percent_level=percent_level, **kwargs)
        return PIL.Image.fromarray(img.permute(1, 2, 0).cpu().numpy())

    def heatmap(self, activations, unit=None, mode='bilinear', amax=None, amin=None):
        from matplotlib import cm
        a = activations if unit is None else activations[unit]
        if amax is None or amin is None:
            amin, amax = activations.min(), activations.max()
        up = _upsample(a, self.size, mode=mode).cpu()
        return PIL.Image.fromarray((cm.hot(((up - amin) / (1e-10 + amax - amin)).numpy()) * 255).astype('uint8'))
