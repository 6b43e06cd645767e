"""Stand-alone reproducer: does kernel A (the "aggressor", on the current stream) change the result of kernel B (the "victim",
on a second stream) when the two overlap?  Both are deterministic on their own.  One JSON line per (aggressor, victim) pair:
how many of the victim's overlapped results differ from its solo result, the largest deviation, and where the bad elements
sit (lanes of a 64-lane wave, assuming 4 consecutive pixels per lane as in to_rgb_kernel)."""
import json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rewriting_amd import hip
DEV = 'cuda:0'
B = int(os.environ.get('B', '8'))
g = torch.Generator().manual_seed(0)

# ---- aggressors on the layer-17 shape
cin, cout, res = 64, 32, 512
x = torch.randn(B, cin, res, res, device=DEV)
wt = torch.randn(1, cout, cin, 3, 3, generator=g).to(DEV)
style = (1 + 0.3 * torch.randn(B, cin, generator=g)).to(DEV)
s = 1 / math.sqrt(cin * 9)
dm = hip.demod(hip.weight_sqsum(wt, s), style)
bias = torch.randn(cout, generator=g).to(DEV)
nw = torch.tensor([0.1], device=DEV)
noise = torch.randn(B, 1, 2 * res, 2 * res, device=DEV)
k1 = torch.tensor([1., 3., 3., 1.]); k4 = k1[:, None] * k1[None, :]; k4 = (k4 / k4.sum() * 4).to(DEV)
amax = hip.absmax(x)
ep = dict(style=style, demod=dm, noise=noise, noise_w=nw, bias=bias, act=True, x_amax=amax)
pk_plain = hip.pack_conv_weight_direct16(wt)
pk_phase = hip.pack_conv_transpose_blur_weight_direct16(wt, k4)
uf_phase = hip.pack_conv_transpose_blur_weight_wino4(wt, k4, split=True)
wt1 = torch.randn(1, 64, 64, 3, 3, generator=g).to(DEV)
pk_s1 = hip.pack_conv_weight_direct16(wt1)
dm1 = hip.demod(hip.weight_sqsum(wt1, s), style)
noise1 = torch.randn(B, res * res, device=DEV)
bias1 = torch.randn(64, generator=g).to(DEV)
aggressors = {
    'tconv_fused': lambda: hip.conv_transpose3x3s2_blur_fused(x, pk_plain, k4, cout, s, **ep),
    'dconv_ws_up (default layer 17)': lambda: hip.conv_transpose3x3s2_blur_direct16(x, pk_phase, cout, s, **ep),
    'conv_up_wino36h (F4x4 one-pass)': lambda: hip.conv_transpose3x3s2_blur_wino4(x, uf_phase, cout, s, **ep),
    'dconv stride-1 (layer 16 shape)': lambda: hip.conv3x3_direct16(x, pk_s1, 64, s, style=style, demod=dm1, noise=noise1, noise_w=nw, bias=bias1, act=True, x_amax=amax),
    'none': lambda: None,
}
# ---- victims
xr = torch.randn(B, 64, 512, 512, device=DEV)
wr = torch.randn(3, 64, device=DEV)
sr = (1 + 0.3 * torch.randn(B, 64, device=DEV))
br = torch.randn(3, device=DEV)
skip = torch.randn(B, 3, 512, 512, device=DEV)
ku = (k1[:, None] * k1[None, :] / 16 * 4).to(DEV)
victims = {
    'to_rgb_kernel': lambda: hip.to_rgb(xr, wr, sr, br, skip, 0.125),
    'torch fma (x * 1.5 + skip-like)': lambda: torch.addcmul(xr, xr, xr, value=0.5),
    'torch sum over channels': lambda: xr.sum(1),
}
only_a = os.environ.get('AGG'); only_v = os.environ.get('VIC')
side = torch.cuda.Stream()
for vn, vic in victims.items():
    if only_v and only_v not in vn: continue
    ref = vic(); torch.cuda.synchronize()
    for an, agg in aggressors.items():
        if only_a and only_a not in an: continue
        agg(); torch.cuda.synchronize()
        outs = []
        main = torch.cuda.current_stream()
        for rep in range(int(os.environ.get('REPS', '12'))):
            side.wait_stream(main)
            agg()
            with torch.cuda.stream(side):
                outs.append(vic())
            agg()
        torch.cuda.synchronize()
        bad, worst, lanes = 0, 0.0, set()
        for o in outs:
            d = (o - ref).abs()
            m = d.max().item()
            if m > 0:
                bad += 1; worst = max(worst, m)
                cols = (d > 0).reshape(-1, d.shape[-1]).any(0).nonzero().flatten()
                lanes |= set(((cols // 4) % 64).tolist())
        print(json.dumps(dict(victim=vn, aggressor=an, deviating=bad, of=len(outs), worst=worst,
                              lanes=sorted(lanes)[:8] + (['..', max(lanes)] if len(lanes) > 8 else []))), flush=True)
        del outs
