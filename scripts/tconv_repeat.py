"""Debug: back-to-back launches of the fused kernel (no host sync in between) must all produce the same map."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rewriting_amd import hip
DEV = 'cuda:0'
batch, cin, cout, res = int(os.environ.get('B', '16')), 64, 32, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(batch, cin, res, res, device=DEV)
wt = torch.randn(1, cout, cin, 3, 3, generator=g).to(DEV)
style = (1 + 0.3 * torch.randn(batch, cin, generator=g)).to(DEV)
s = 1 / math.sqrt(cin * 9)
dm = hip.demod(hip.weight_sqsum(wt, s), style)
bias = torch.randn(cout, generator=g).to(DEV)
nw = torch.tensor([0.1], device=DEV)
noise = torch.randn(batch, 1, 2 * res, 2 * res, device=DEV)
k1 = torch.tensor([1., 3., 3., 1.]); k4 = k1[:, None] * k1[None, :]; k4 = (k4 / k4.sum() * 4).to(DEV)
post = (1 + 0.3 * torch.randn(batch, cout, generator=g)).to(DEV)
pk = hip.pack_conv_weight_direct16(wt)
amax = hip.absmax(x)
args = dict(style=style, demod=dm, noise=noise, noise_w=nw, bias=bias, act=True, post_scale=post)
ref = hip.conv_transpose3x3s2_blur_fused(x, pk, k4, cout, s, x_amax=amax, **args)
torch.cuda.synchronize()
for mode in ('same-bound', 'fresh-bound', 'with-ybound', 'other-kernel-between'):
    outs = []
    for it in range(12):
        a = amax if mode == 'same-bound' else hip.absmax(x)
        yb = hip.new_bound(batch * cout * 4 * res * res, DEV) if mode == 'with-ybound' else None
        if mode == 'other-kernel-between':
            junk = torch.randn(batch, cin, res, res, device=DEV)       # something else touching memory / the CUs
        outs.append(hip.conv_transpose3x3s2_blur_fused(x, pk, k4, cout, s, x_amax=a, y_amax=yb, **args))
    torch.cuda.synchronize()
    print(mode, [round((o - ref).abs().max().item(), 6) for o in outs], flush=True)
