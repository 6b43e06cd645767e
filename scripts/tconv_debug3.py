"""Debug: in the failing (overlapping) configuration, which tensor is wrong -- layer 17's output or the RGB skip?"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rewriting_amd import hip, synthetic
from rewriting_amd.utils.stylegan2 import models

dev = 'cuda:0'
g = models.SeqStyleGAN2(1024, 512, 8, truncation=0.5, mconv='seq')
synthetic.randomize_(g, seed=0)
g = g.eval().to(dev)
z = torch.randn(8, 512, generator=torch.Generator().manual_seed(1)).to(dev)
real, real_rgb = hip.conv_transpose3x3s2_blur_fused, hip.conv3x3_wino4_to_rgb
log = {}

def spy(x, *a, **kw):
    out = real(x, *a, **kw)
    log['l17_in'] = x.clone(); log['l17_out'] = out.clone()        # AFTER the launch: nothing is delayed in front of it
    if kw.get('y_amax') is not None:
        log['l17_ybound'] = kw['y_amax'][:64].clone()
    return out

def spy_rgb(x, uf, out_ch, w_scale, rgb_weight, rgb_style, rgb_bias, rgb_skip, rgb_scale, **kw):
    log['skip'] = rgb_skip.clone()
    log['l18_in'] = x.clone()
    return real_rgb(x, uf, out_ch, w_scale, rgb_weight, rgb_style, rgb_bias, rgb_skip, rgb_scale, **kw)
hip.conv_transpose3x3s2_blur_fused, hip.conv3x3_wino4_to_rgb = spy, spy_rgb
os.environ['RW_UP_FUSED2'] = '1'
os.environ['RW_UP_FUSED2_MAX_IN'] = '64'
os.environ['RW_UP_FUSED2_JOIN'] = '1'
with torch.no_grad():
    ref_img = g(z)
torch.cuda.synchronize()
ref = dict(log)
os.environ.pop('RW_UP_FUSED2_JOIN')
for rep in range(4):
    log.clear()
    with torch.no_grad():
        img = g(z)
    torch.cuda.synchronize()
    print(json.dumps(dict(rep=rep, image=(img - ref_img).abs().max().item(),
                          **{k: (log[k] - ref[k]).abs().max().item() for k in sorted(ref) if k in log})), flush=True)
    if (log['skip'] - ref['skip']).abs().max().item() > 1e-3:
        d = (log['skip'] - ref['skip']).abs()
        bad = (d > 1e-4)
        idx = bad.nonzero()
        rows = idx[:, 2]; cols = idx[:, 3]
        print('  skip: bad', int(bad.sum()), 'of', d.numel(), 'images', sorted(set(idx[:, 0].tolist())), 'channels', sorted(set(idx[:, 1].tolist())),
              'rows', int(rows.min()), int(rows.max()), 'cols', int(cols.min()), int(cols.max()))
        im = idx[0, 0].item()
        sub = bad[im, 0]
        rr = sub.any(1).nonzero().flatten(); cc = sub.any(0).nonzero().flatten()
        def runs(v):
            v = v.tolist(); out = []; start = v[0]; prev = v[0]
            for x in v[1:]:
                if x != prev + 1: out.append((start, prev)); start = x
                prev = x
            out.append((start, prev)); return out
        print('  image', im, 'n bad rows', len(rr), 'row runs', runs(rr)[:10], 'n bad cols', len(cc), 'col runs', runs(cc)[:10])
    if (log['l17_out'] - ref['l17_out']).abs().max().item() > 1e-3:
        d = (log['l17_out'] - ref['l17_out']).abs()
        bad = (d > 1e-3)
        idx = bad.nonzero()
        print('  bad elements', int(bad.sum()), 'of', d.numel(), 'first', idx[0].tolist(), 'last', idx[-1].tolist(),
              'images', sorted(set(idx[:, 0].tolist())), 'channels', sorted(set(idx[:, 1].tolist()))[:40],
              'rows', int(idx[:, 2].min()), int(idx[:, 2].max()), 'cols', int(idx[:, 3].min()), int(idx[:, 3].max()))
