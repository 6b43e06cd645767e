"""What ARE the wrong values of to_rgb_kernel when rw_tconv's kernel runs beside it?  (scripts/interference_repro.py)"""
import json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rewriting_amd import hip
DEV = 'cuda:0'
B = 8
g = torch.Generator().manual_seed(0)
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
pk = hip.pack_conv_weight_direct16(wt)
agg = lambda: hip.conv_transpose3x3s2_blur_fused(x, pk, k4, cout, s, **ep)
xr = torch.randn(B, 64, 512, 512, device=DEV)
wr = torch.randn(3, 64, device=DEV)
sr = (1 + 0.3 * torch.randn(B, 64, device=DEV))
br = torch.randn(3, device=DEV)
skip = torch.randn(B, 3, 512, 512, device=DEV)
vic = lambda: hip.to_rgb(xr, wr, sr, br, skip, 0.125)
ref = vic(); torch.cuda.synchronize()
# partial sums of the reference: contribution of each channel i -> to identify "which channels are missing / doubled"
wm = 0.125 * wr[None] * sr[:, None, :]                                 # (B, 3, C)
side = torch.cuda.Stream(); main = torch.cuda.current_stream()
agg(); torch.cuda.synchronize()
side.wait_stream(main)
agg()
with torch.cuda.stream(side):
    out = vic()
agg()
torch.cuda.synchronize()
d = (out - ref)
bad = d.abs() > 0
print('bad elements', int(bad.sum()), 'of', bad.numel())
idx = bad.nonzero()
b0, c0, y0, x0 = idx[0].tolist()
print('first bad at', (b0, c0, y0, x0), 'out', out[b0, c0, y0, x0].item(), 'ref', ref[b0, c0, y0, x0].item())
# hypothesis 1: a wrong pixel was computed (addressing): does out equal ref somewhere else in the same image / colour?
row = out[b0, c0, y0]
cands = (ref[b0, c0] == row[x0]).nonzero()
print('same value elsewhere in the reference plane:', cands[:5].tolist())
# hypothesis 2: some channels' contributions are missing / stale: solve for per-channel coefficients on the bad pixels of this row
xs = idx[(idx[:, 0] == b0) & (idx[:, 1] == c0) & (idx[:, 2] == y0)][:, 3]
print('bad columns in that row:', xs.tolist()[:70])
px = xr[b0, :, y0, xs]                                                  # (C, n)
dd = d[b0, c0, y0, xs]                                                  # (n,)
contrib = wm[b0, c0][:, None] * px                                      # (C, n)
# is d a sum of -contrib over some set of channels?  least squares for coefficients a_i: d = sum a_i contrib_i
A = contrib.t().double().cpu(); y = dd.double().cpu()
coef = torch.linalg.lstsq(A, y[:, None]).solution.flatten() if A.shape[0] >= A.shape[1] else None
if coef is not None:
    print('lstsq coefficients (a_i = -1: channel i missing, +1: counted twice), residual', (A @ coef - y).abs().max().item())
    print([round(v, 3) for v in coef.tolist()])
# hypothesis 3: values of a different row / image: compare against ref of neighbouring rows
for dy in (-2, -1, 1, 2):
    yy = y0 + dy
    if 0 <= yy < 512:
        print('equals ref row', yy, ':', bool((out[b0, c0, y0, xs] == ref[b0, c0, yy, xs]).all().item()))
for db in range(B):
    if db != b0:
        print('equals ref image', db, ':', bool((out[b0, c0, y0, xs] == ref[db, c0, y0, xs]).all().item()))
