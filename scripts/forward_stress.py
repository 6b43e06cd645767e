"""Stress: N forwards of the 1024 generator issued back to back (no host sync), each compared with the synced reference.
Usage: python scripts/forward_stress.py <batch> <reps> [ENV=value ...]; prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
batch, reps = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split('=', 1)
    os.environ[k] = v
import torch
from rewriting_amd import synthetic
from rewriting_amd.utils.stylegan2 import models
size = int(os.environ.get('SIZE', '1024'))
g = models.SeqStyleGAN2(size, 512, 8, truncation=0.5, mconv='seq')
synthetic.randomize_(g, seed=0)
g = g.eval().to('cuda:0')
z = torch.randn(batch, 512, generator=torch.Generator().manual_seed(1)).to('cuda:0')
with torch.no_grad():
    g(z); torch.cuda.synchronize()
    ref = g(z); torch.cuda.synchronize()
    bad, worst = 0, 0.0
    chunk = max(1, min(reps, int(os.environ.get('CHUNK', '8'))))
    done = 0
    while done < reps:
        outs = [g(z) for _ in range(min(chunk, reps - done))]
        torch.cuda.synchronize()
        for o in outs:
            d = (o - ref).abs().max().item()
            bad += d > 0
            worst = max(worst, d)
        done += len(outs)
        del outs
print(json.dumps(dict(size=size, batch=batch, reps=reps, env=sys.argv[3:], deviating=bad, worst=worst)))
