"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's hot-path arithmetic.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file, and only as the checker.  The product package
(``rewriting_amd``) never imports it.

Why it exists: the reference is Python and cannot travel to the GPU box
(``/root/reference`` is absent there), so the parity tests need a self-contained
statement of the same arithmetic.  It is written as plain functions over a
state dict (float32 torch CPU tensors, the same ATen CPU kernels the reference's
CPU path executes), each citing the reference lines it follows.

Pinning: the reference holds NO golden vectors or known-answer tests for this path
(SURVEY.md section 4 / 8c).  This restatement is therefore pinned against outputs of
the reference's own files executed in the build container through
``oracle/reference_shim.py``; the generating script is ``oracle/make_golden.py`` and the
fixtures are ``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` compares them;
when ``/root/reference`` is present the same test also runs the reference live).
"""
import base64
import io
import math
import re

import numpy
import torch
import torch.nn.functional as F

SQRT2 = 2 ** 0.5


# ----------------------------------------------------------------------------
# Native ops (utils/stylegan2/op/*.cu) -- torch statement; the C statement of the
# same two kernels is oracle/native_ops.c.
# ----------------------------------------------------------------------------

def fused_bias_act(x, b=None, ref=None, act=3, grad=0, alpha=0.2, scale=SQRT2):
    """fused_bias_act_kernel.cu:18-49.  ``b`` indexed by (xi / step_b) % size_b with
    step_b = prod(x.shape[2:]) (:67-71)."""
    if b is not None and b.numel():
        x = x + b.view(*([1, -1] + [1] * (x.ndim - 2)))
    code = act * 10 + grad
    if code in (10, 11):
        y = x
    elif code in (12, 32):
        y = torch.zeros_like(x)
    elif code == 30:
        y = torch.where(x > 0, x, x * alpha)
    elif code == 31:
        y = torch.where(ref > 0, x, x * alpha)
    else:
        y = x
    return y * scale


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=SQRT2):
    """op/fused_act.py:51-86 forward."""
    return fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)


def fused_leaky_relu_backward(grad_out, out, negative_slope=0.2, scale=SQRT2):
    """op/fused_act.py:19-39: grad_input through kernel case 31 with ref=out, then
    grad_bias = grad_input.sum over every dim but 1."""
    gi = fused_bias_act(grad_out, None, out, 3, 1, negative_slope, scale)
    dims = [0] + list(range(2, gi.ndim))
    return gi, gi.sum(dims)


def upfirdn2d_major(x, k, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    """Semantics of upfirdn2d_kernel.cu:52-137 on a (major, H, W, minor) tensor:
    zero-insert upsample, pad/crop, correlate with the FLIPPED kernel (:71-81),
    decimate.  out_h per :167-168.  Written from the kernel's index algebra
    (mid = out*down + up - 1 - pad; in = floor(mid/up); tap = (in+1)*up - mid - 1)."""
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h = (in_h * up_y + py0 + py1 - kh + down_y) // down_y
    out_w = (in_w * up_x + px0 + px1 - kw + down_x) // down_x
    # in the tensor's own precision, as the kernel (scalar_t accumulators, upfirdn2d_kernel.cu:108-133) and the
    # reference's torch spec (op/upfirdn2d.py:152-186) compute it
    xd = x
    kd = k.to(x.dtype)
    up = torch.zeros(major, in_h * up_y, in_w * up_x, minor, dtype=x.dtype)
    up[:, ::up_y, ::up_x, :] = xd
    ph, pw = in_h * up_y + py0 + py1, in_w * up_x + px0 + px1
    pad = torch.zeros(major, max(ph, 0), max(pw, 0), minor, dtype=x.dtype)
    # copy the overlap of the upsampled image into the padded/cropped canvas
    sy0, sx0 = max(-py0, 0), max(-px0, 0)
    dy0, dx0 = max(py0, 0), max(px0, 0)
    hh = min(in_h * up_y - sy0, ph - dy0)
    ww = min(in_w * up_x - sx0, pw - dx0)
    if hh > 0 and ww > 0:
        pad[:, dy0:dy0 + hh, dx0:dx0 + ww, :] = up[:, sy0:sy0 + hh, sx0:sx0 + ww, :]
    img = pad.permute(0, 3, 1, 2).reshape(-1, 1, ph, pw)
    w = torch.flip(kd, [0, 1]).view(1, 1, kh, kw)
    full = F.conv2d(img, w)
    full = full.reshape(major, minor, full.shape[2], full.shape[3]).permute(0, 2, 3, 1)
    out = full[:, ::down_y, ::down_x, :][:, :out_h, :out_w, :]
    return out.to(x.dtype)


def upfirdn2d(x, k, up=1, down=1, pad=(0, 0)):
    """op/upfirdn2d.py:144-149 on NCHW (view algebra :87-127)."""
    b, c, h, w = x.shape
    out = upfirdn2d_major(x.reshape(-1, h, w, 1), k, up, up, down, down,
                          pad[0], pad[1], pad[0], pad[1])
    return out.view(b, c, out.shape[1], out.shape[2])


def upfirdn2d_backward(grad_out, k, up, down, pad, in_shape):
    """op/upfirdn2d.py:17-50,100-115: the adjoint is upfirdn2d with up<->down swapped,
    the flipped kernel and g_pad."""
    b, c, in_h, in_w = in_shape
    kh, kw = k.shape
    out_h, out_w = grad_out.shape[2:]
    gx0, gy0 = kw - pad[0] - 1, kh - pad[0] - 1
    gx1 = in_w * up - out_w * down + pad[0] - up + 1
    gy1 = in_h * up - out_h * down + pad[0] - up + 1
    g = upfirdn2d_major(grad_out.reshape(-1, out_h, out_w, 1), torch.flip(k, [0, 1]),
                        down, down, up, up, gx0, gx1, gy0, gy1)
    return g.view(in_shape)


def make_kernel(k):
    """utils/stylegan2/models.py:449-454."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


# ----------------------------------------------------------------------------
# Generator forward (utils/stylegan2/models.py)
# ----------------------------------------------------------------------------

def noise_rows(batch, hw):
    """models.py:542-545: RandomState(0).randn(batch, H*W) regenerated per call (quirk Q1)."""
    return torch.from_numpy(numpy.random.RandomState(0).randn(batch, hw).astype('float32'))


def equal_linear(x, w, b, lr_mul=1.0, activation=False):
    """models.py:487-511."""
    scale = (1 / math.sqrt(w.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, w * scale), b * lr_mul)
    return F.linear(x, w * scale, bias=b * lr_mul)


def mapping(sd, z, n_mlp=8, lr_mlp=0.01):
    """PixelNormL (models.py:609-614) + n_mlp EqualLinearL with fused lrelu (:59-65)."""
    x = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(1, n_mlp + 1):
        x = equal_linear(x, sd['style.%d.weight' % i], sd['style.%d.bias' % i],
                         lr_mul=lr_mlp, activation=True)
    return x


def adjust_latent(sd, w, n_latent, truncation):
    """AdjustLatent (models.py:570-583)."""
    avg = sd['latents.latent_avg']
    if truncation != 1.0 and avg.ndim > 0:
        w = avg + truncation * (w - avg)
    return w.unsqueeze(1).repeat(1, n_latent, 1)


def demod_conv(x, style, weight, upsample):
    """DemodulatedConv2dF.forward (models.py:313-329). x is already style-multiplied."""
    _, o, i, kh, kw = weight.shape
    scale = 1 / math.sqrt(i * kh * kw)
    if upsample:
        out = F.conv_transpose2d(x, scale * weight.transpose(1, 2).squeeze(0), padding=0, stride=2)
    else:
        out = F.conv2d(x, scale * weight.squeeze(0), padding=kh // 2)
    b = x.shape[0]
    temp = scale * weight * style.view(b, 1, i, 1, 1)
    demod = torch.rsqrt(temp.pow(2).sum([2, 3, 4]) + 1e-8)
    return out * demod[:, :, None, None]


def styled_conv(sd, prefix, fmap, latent_row, upsample, taps=None):
    """StyledConvSeq = modulation -> adain -> dconv -> [blur] -> noise -> activate
    (models.py:232-289).  ``prefix`` e.g. 'layer8.sconv'.  Returns dict of stage outputs."""
    style = equal_linear(latent_row, sd[prefix + '.mconv.modulation.weight'],
                         sd[prefix + '.mconv.modulation.bias'])
    key = style[:, :, None, None] * fmap                       # ApplyStyle :616-620
    out = demod_conv(key, style, sd[prefix + '.mconv.dconv.weight'], upsample)
    stages = dict(style=style, adain=key, dconv=out)
    if upsample:
        out = upfirdn2d(out, sd[prefix + '.mconv.blur.kernel'], pad=(1, 1))  # :277-281,481-485
        stages['blur'] = out
    b, _, h, w = out.shape
    out = out + sd[prefix + '.noise.weight'] * noise_rows(b, h * w).view(b, 1, h, w)
    stages['noise'] = out
    out = fused_leaky_relu(out, sd[prefix + '.activate.bias'])
    stages['activate'] = out
    return out, stages


def to_rgb(sd, prefix, fmap, latent_row, skip):
    """ToRGBF (models.py:628-655) with ModulatedConv2d demodulate=False k=1 (:394-425)."""
    w = sd[prefix + '.rgb.conv.weight']                        # (1,3,C,1,1)
    c = w.shape[2]
    style = equal_linear(latent_row, sd[prefix + '.rgb.conv.modulation.weight'],
                         sd[prefix + '.rgb.conv.modulation.bias'])
    b = fmap.shape[0]
    wmod = (1 / math.sqrt(c)) * w * style.view(b, 1, c, 1, 1)
    out = F.conv2d(fmap.reshape(1, b * c, *fmap.shape[2:]),
                   wmod.view(b * 3, c, 1, 1), groups=b).view(b, 3, *fmap.shape[2:])
    out = out + sd[prefix + '.rgb.bias']
    if skip is not None:
        out = out + skip
    return out


def generator_forward(sd, z, size, truncation=1.0, n_mlp=8, collect=None):
    """SeqStyleGAN2.forward with mconv='seq' (models.py:92-141).  ``collect`` (a dict)
    receives every named stage output, keyed like nethook names."""
    log_size = int(math.log(size, 2))
    n_latent = log_size * 2 - 2
    w = mapping(sd, z, n_mlp)
    lat = adjust_latent(sd, w, n_latent, truncation)
    b = z.shape[0]
    fmap = sd['input.input'].repeat(b, 1, 1, 1)

    def rec(name, val):
        if collect is not None:
            collect[name] = val

    rec('style', w)
    rec('latents', lat)
    fmap, st = styled_conv(sd, 'layer2.conv', fmap, lat[:, 0], False)
    for k, v in st.items():
        rec('layer2.conv.' + k, v)
    out = to_rgb(sd, 'to_rgb1', fmap, lat[:, 1], None)
    rec('to_rgb1', out)
    lat_i = 1
    for i in range(3, log_size + 1):
        out = upfirdn2d(out, sd['up_rgb%d.kernel' % (i - 2)], up=2, pad=(2, 1))  # UpsampleO :435-447
        rec('up_rgb%d' % (i - 2), out)
        for j, ups in ((lat_i + 2, True), (lat_i + 3, False)):
            fmap, st = styled_conv(sd, 'layer%d.sconv' % j, fmap, lat[:, j - 2], ups)
            for k, v in st.items():
                rec('layer%d.sconv.%s' % (j, k), v)
        out = to_rgb(sd, 'to_rgb%d' % (i - 1), fmap, lat[:, lat_i + 2], out)
        rec('to_rgb%d' % (i - 1), out)
        lat_i += 2
    return out


def context_forward(sd, z, size, layernum, truncation=1.0, n_mlp=8):
    """What SeqStyleGanRewriter.context_model computes (rewrite/ganrewrite.py:47-50,
    662-665): everything up to and including layerN...adain.  Returns (key fmap, style,
    rgb output so far)."""
    col = {}
    log_size = int(math.log(size, 2))
    n_latent = log_size * 2 - 2
    w = mapping(sd, z, n_mlp)
    lat = adjust_latent(sd, w, n_latent, truncation)
    b = z.shape[0]
    fmap = sd['input.input'].repeat(b, 1, 1, 1)
    out = None
    lat_i = 1
    j = 2
    seq = [(2, 'layer2.conv', False)]
    for i in range(3, log_size + 1):
        seq += [(lat_i + 2, 'layer%d.sconv' % (lat_i + 2), True),
                (lat_i + 3, 'layer%d.sconv' % (lat_i + 3), False)]
        lat_i += 2
    for j, prefix, ups in seq:
        if j % 2 == 1:                                         # up_rgbK precedes odd layers
            out = upfirdn2d(out, sd['up_rgb%d.kernel' % ((j - 1) // 2)], up=2, pad=(2, 1))
        if j == layernum:
            style = equal_linear(lat[:, j - 2], sd[prefix + '.mconv.modulation.weight'],
                                 sd[prefix + '.mconv.modulation.bias'])
            return style[:, :, None, None] * fmap, style, out
        fmap, _ = styled_conv(sd, prefix, fmap, lat[:, j - 2], ups)
        if j % 2 == 0:
            out = to_rgb(sd, 'to_rgb%d' % (j // 2), fmap, lat[:, j - 1], out)
    raise ValueError('layer %d not in a size-%d generator' % (layernum, size))


def target_forward(sd, layernum, key, style, upsample=None):
    """target_model = dconv -> [blur] -> noise -> activate (ganrewrite.py:51-55,662-665)."""
    prefix = 'layer%d.sconv' % layernum
    if upsample is None:
        upsample = (layernum % 2 == 1)
    out = demod_conv(key, style, sd[prefix + '.mconv.dconv.weight'], upsample)
    if upsample:
        out = upfirdn2d(out, sd[prefix + '.mconv.blur.kernel'], pad=(1, 1))
    b, _, h, w = out.shape
    out = out + sd[prefix + '.noise.weight'] * noise_rows(b, h * w).view(b, 1, h, w)
    return fused_leaky_relu(out, sd[prefix + '.activate.bias'])


# ----------------------------------------------------------------------------
# Key statistics (utils/tally.py, utils/runningstats.py, rewrite/ganrewrite.py)
# ----------------------------------------------------------------------------

def second_moment(key_batches):
    """tally_second_moment + RunningSecondMoment (tally.py:424-443, runningstats.py:1086-1109):
    mom2 += a^T a over batches, moment = mom2 / count.  ``addbmm_`` of rank-1 outer
    products is restated as the equivalent GEMM."""
    mom2, count = None, 0
    for acts in key_batches:
        a = acts.permute(0, 2, 3, 1).reshape(-1, acts.shape[1])
        if mom2 is None:
            mom2 = torch.zeros(a.shape[1], a.shape[1], dtype=a.dtype)
        mom2 += a.t() @ a
        count += a.shape[0]
    return mom2 / count, mom2, count


def zca_from_cov(cov):
    """rewrite/ganrewrite.py:821-826 (symeig used the upper triangle)."""
    evals, evecs = torch.linalg.eigh(cov.double(), UPLO='U')
    return (evecs @ torch.diag(evals.sqrt().clamp(1e-20).reciprocal()) @ evecs.t()).to(cov.dtype)


def mask_from_url(url, size):
    """renormalize.from_url(target='pt', size)[0] (utils/renormalize.py:35-50): base64 PNG
    -> RGB -> PIL bilinear resize to the feature size -> channel R in [0,1] (quirk Q4)."""
    import PIL.Image
    data = re.sub('^data:image/.+;base64,', '', url)
    im = PIL.Image.open(io.BytesIO(base64.b64decode(data)))
    if im.format != 'RGB':
        im = im.convert('RGB')
    if size is not None:
        im = im.resize(tuple(size), resample=PIL.Image.BILINEAR)
    arr = numpy.asarray(im)[:, :, 0]
    return torch.from_numpy(arr.astype('float32') / 255.0)


def multi_key_zca(obs_list, weight_list, zca, rank):
    """multi_key_from_selection, zca branch (ganrewrite.py:339-374).
    obs_list: per example (H*W, C) key rows; weight_list: per example (H*W, 1) mask."""
    zk = torch.cat([(w * (zca @ obs.t()).t())[(w > 0).nonzero()[:, 0], :]
                    for obs, w in zip(obs_list, weight_list)])
    _, _, q = torch.linalg.svd(zk, full_matrices=False)
    q = q.t()                                                  # Tensor.svd returned V
    top = q[:, :rank]
    row_dirs = (zca @ top).t()
    just_avg = zk.sum(0)
    qq, _ = torch.linalg.qr(row_dirs.t())
    signs = (qq * just_avg[:, None]).sum(0).sign()
    return (qq * signs[None, :]).t(), zk


def positive_bounding_box(data):
    """ganrewrite.py:767-777."""
    pos = data > 0
    if pos.sum() == 0:
        return 0, 0, 0, 0
    v, h = pos.sum(0).nonzero(), pos.sum(1).nonzero()
    return h.min().item(), v.min().item(), h.max().item() + 1, v.max().item() + 1


def centered_location(data):
    t, l, b, r = positive_bounding_box(data)
    return (t + b) // 2, (l + r) // 2


def paste_clip_at_center(source, clip, center, area=None):
    """ganrewrite.py:785-794."""
    target = source.clone()
    t, l = (max(0, min(e - s, c - s // 2))
            for s, c, e in zip(clip.shape[2:], center, source.shape[2:]))
    b, r = t + clip.shape[2], l + clip.shape[3]
    if area is None:
        target[:, :, t:b, l:r] = clip
    else:
        a = area[None, None]
        target[:, :, t:b, l:r] = (1 - a) * target[:, :, t:b, l:r] + a * clip
    return target, (t, l, b, r)


def crop_clip_to_bounds(source, target, bounds):
    """ganrewrite.py:797-803."""
    t, l, b, r = bounds
    vr, hr = [ts // ss for ts, ss in zip(target.shape[2:], source.shape[2:])]
    st, sl, sb, sr = t // vr, l // hr, -(-b // vr), -(-r // hr)
    tt, tl, tb, tr = st * vr, sl * hr, sb * vr, sr * hr
    return source[:, :, st:sb, sl:sr], target[:, :, tt:tb, tl:tr], (st, sl, sb, sr), (tt, tl, tb, tr)


def projected_conv(weight, direction):
    """ganrewrite.py:806-813."""
    if weight.ndim == 5:
        cos = torch.einsum('goiyx,di->godyx', weight, direction)
        return torch.einsum('godyx,di->goiyx', cos, direction)
    cos = torch.einsum('oiyx,di->odyx', weight, direction)
    return torch.einsum('odyx,di->oiyx', cos, direction)


# ----------------------------------------------------------------------------
# The solve (rewrite/ganrewrite.py:254-298) -- explicit arithmetic of SURVEY.md
# section 10, NOT autograd, so that it is an independent statement of the update.
# ----------------------------------------------------------------------------

def insert_explicit(W0, key, style, val, bias, noise_w, context, niter, piter=10, lr=0.05,
                    low_rank_insert=True, low_rank_gradient=False, snapshots=(),
                    dtype=torch.float32, conv_noise=None):
    """Stride-1 layer solve.  W0 (1,O,I,3,3); key (1,I,h,w) = adain output crop; style (1,I);
    val (1,O,h,w); context (r,I) orthonormal.  Returns (W, losses, {it+1: W snapshot}).
    conv_noise = (amplitude, torch.Generator): every convolution result is perturbed by amplitude x its mean
    magnitude -- a sensitivity probe (how far the trajectory moves when the convolution is rounded differently),
    used by oracle/make_golden.py to record scatter; None for the restatement itself."""
    W = W0.clone().to(dtype)
    key, style, val, bias, context = [t.to(dtype) for t in (key, style, val, bias, context)]
    noise_w = float(noise_w)
    _, O, I, kh, kw = W.shape
    s = 1 / math.sqrt(I * kh * kw)
    h, w = key.shape[2:]
    n = noise_rows(1, h * w).view(1, 1, h, w).to(dtype)
    m = torch.zeros_like(W)
    v = torch.zeros_like(W)
    ortho = W - projected_conv(W, context) if (low_rank_insert or low_rank_gradient) else None
    xcol = F.unfold(key, (kh, kw), padding=kh // 2)[0]          # (I*9, P)  zero pad (quirk Q2)
    losses, snaps = [], {}
    sig2 = (style[0] ** 2).view(1, I, 1, 1)
    for it in range(niter):
        Wm = W[0]
        conv = (s * Wm.reshape(O, -1)) @ xcol                     # (O, P)
        if conv_noise is not None:
            conv = conv + conv_noise[0] * conv.abs().mean() * torch.randn(conv.shape, generator=conv_noise[1]).to(dtype)
        demod = torch.rsqrt(((s * Wm) ** 2 * sig2).sum([1, 2, 3]) + 1e-8)
        pre = conv * demod[:, None] + noise_w * n.view(1, -1) + bias[:, None]
        out = SQRT2 * torch.where(pre > 0, pre, 0.2 * pre)
        diff = out - val[0].reshape(O, -1)
        losses.append(diff.abs().mean().item())
        g_out = torch.sign(diff) / diff.numel()
        g_pre = g_out * SQRT2 * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, 0.2))
        dW = s * ((g_pre * demod[:, None]) @ xcol.t()).view(O, I, kh, kw)
        t_o = (g_pre * conv).sum(1)
        dW = dW - (s * s) * Wm * sig2 * (demod ** 3 * t_o).view(O, 1, 1, 1)
        dW = dW[None]
        if low_rank_gradient:
            dW = projected_conv(dW, context)
        t = it + 1                                              # torch.optim.Adam, single tensor
        m = m + (dW - m) * (1 - 0.9)
        v = v * 0.999 + (1 - 0.999) * dW * dW
        bc1 = 1 - 0.9 ** t
        bc2s = math.sqrt(1 - 0.999 ** t)
        W = W - (lr / bc1) * m / (v.sqrt() / bc2s + 1e-8)
        if low_rank_insert and (it % piter == 0 or it == niter - 1):
            W = ortho + projected_conv(W, context)
        if (it + 1) in snapshots:
            snaps[it + 1] = W.clone()
    return W, losses, snaps


def insert_autograd(W0, forward_fn, val, context, niter, piter=10, lr=0.05,
                    low_rank_insert=True, low_rank_gradient=False, snapshots=()):
    """The same loop through torch autograd + torch.optim.Adam, exactly as the reference
    drives it (ganrewrite.py:271-294); ``forward_fn(W)`` returns the layer output."""
    W = W0.clone().requires_grad_(True)
    opt = torch.optim.Adam([W], lr=lr)
    with torch.no_grad():
        ortho = W - projected_conv(W, context)
    losses, snaps = [], {}
    for it in range(niter):
        loss = F.l1_loss(val, forward_fn(W))
        opt.zero_grad()
        loss.backward()
        if low_rank_gradient:
            W.grad[...] = projected_conv(W.grad, context)
        opt.step()
        losses.append(loss.item())
        if low_rank_insert and (it % piter == 0 or it == niter - 1):
            with torch.no_grad():
                W[...] = ortho + projected_conv(W, context)
        if (it + 1) in snapshots:
            snaps[it + 1] = W.detach().clone()
    return W.detach(), losses, snaps


# ----------------------------------------------------------------------------
# Progressive GAN (utils/proggan.py) -- config 1, plain convs
# ----------------------------------------------------------------------------

def proggan_forward(sd, z, upto=None):
    """ProgressiveGenerator.forward (utils/proggan.py:126-129) over NormConvBlock /
    NormUpscaleConvBlock / OutputConvBlock (:160-193).  ``upto='layer6.conv'`` style names
    stop BEFORE that module and return its input."""
    x = z.view(z.shape[0], z.shape[1], 1, 1)
    names = sorted({k.split('.')[0] for k in sd}, key=lambda s: (not s.startswith('layer'),
                   int(s[5:]) if s.startswith('layer') else 0))
    for idx, name in enumerate(names):
        wt = sd[name + '.conv.weight']
        o, i, kh, _ = wt.shape
        x = x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)   # PixelNormLayer :132-137
        is_out = name.startswith('output')
        if (not is_out) and idx >= 2 and idx % 2 == 0:
            x = F.interpolate(x, scale_factor=2, mode='nearest')            # DoubleResolutionLayer
        if upto == name + '.conv':
            return x
        pad = 3 if idx == 0 else (0 if is_out else 1)
        x = F.conv2d(x, wt, padding=pad)
        gain = 1.0 if is_out else math.sqrt(2) / kh
        x = x * (gain / math.sqrt(i)) + sd[name + '.wscale.b'].view(1, -1, 1, 1)   # WScaleLayer :146-157
        if is_out:
            x = torch.clamp(x, -1, 1)                                        # nn.Hardtanh
        else:
            x = F.leaky_relu(x, 0.2)
    return x
