"""Debug: inside the model forward, compare hip.conv_transpose3x3s2_blur_fused with the phase-kernel one-pass route on the SAME arguments."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rewriting_amd import hip, synthetic
from rewriting_amd.utils.stylegan2 import models

dev = 'cuda:0'
g = models.SeqStyleGAN2(1024, 512, 8, truncation=0.5, mconv='seq')
synthetic.randomize_(g, seed=0)
g = g.eval().to(dev)
z = torch.randn(int(os.environ.get('B', '2')), 512, generator=torch.Generator().manual_seed(1)).to(dev)
real = hip.conv_transpose3x3s2_blur_fused

def spy(x, wp, k4, out_ch, w_scale, **kw):
    got = real(x, wp, k4, out_ch, w_scale, **kw)
    wt = spy.weight[(x.shape[1], out_ch)]
    pk1 = hip.pack_conv_transpose_blur_weight_direct16(wt, k4)
    kw2 = dict(kw); kw2.pop('y_amax', None)
    ref = hip.conv_transpose3x3s2_blur_direct16(x, pk1, out_ch, w_scale, **kw2)
    kw3 = dict(kw2); kw3.pop('x_amax', None)
    ref2 = hip.conv_transpose3x3s2_blur_direct16(x, pk1, out_ch, w_scale, **kw3)
    again = real(x, wp, k4, out_ch, w_scale, **kw3)
    print(json.dumps(dict(shape=list(x.shape), out_ch=out_ch, linf=(got - ref).abs().max().item(), range=ref.abs().max().item(),
                          ref_vs_measured_bound=(ref - ref2).abs().max().item(), fused_measured_bound=(again - ref2).abs().max().item(),
                          bound=hip.bound_value(kw['x_amax']) if kw.get('x_amax') is not None else None, xmax=x.abs().max().item(),
                          keys=sorted(kw), finite=bool(torch.isfinite(got).all()))), flush=True)
    return got
spy.weight = {}
for name, m in g.named_modules():
    if isinstance(m, models.DemodulatedConv2dF) and m.upsample:
        spy.weight[(m.in_channel, m.out_channel)] = m.weight.detach()
if not os.environ.get('NOSPY'):
    hip.conv_transpose3x3s2_blur_fused = spy
os.environ['RW_UP_FUSED2'] = '1'
os.environ['RW_UP_FUSED2_MAX_IN'] = os.environ.get('MAXIN', '128')
with torch.no_grad():
    a = g(z)
os.environ['RW_UP_FUSED2'] = '0'
with torch.no_grad():
    b = g(z)
os.environ['RW_UP_FUSED2'] = '1'
print('image diff', (a - b).abs().max().item(), 'per image', [round(v, 6) for v in (a - b).abs().flatten(1).max(1).values.tolist()][:16])
if os.environ.get('NOSPY'):
    with torch.no_grad():
        c = g(z)
        torch.cuda.synchronize()
        c2 = g(z)
    print('fused again vs fused', (a - c).abs().max().item(), (c - c2).abs().max().item(), 'vs base', (c - b).abs().max().item())
