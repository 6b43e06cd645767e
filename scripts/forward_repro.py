"""Round-4 open issue: a forward of the generator that is IMMEDIATELY followed by another forward (no host sync in between)
occasionally comes out different (0.03 - 0.6 on an image of range +-8) from the same forward followed by a sync.  Runs the
sequence of tests/test_gpu_model.py::test_premultiplied_style_and_one_pass_layers_on_the_image_path several times, with and
without syncs, and prints which output deviated from its synced twin.  Usage: python scripts/forward_repro.py [size] [reps] [ENV=value ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import build_stylegan  # noqa: E402

DEV = 'cuda:0'
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
model = build_stylegan(size, 0.7, device=DEV)
z = torch.randn(4, 512, generator=torch.Generator().manual_seed(9)).to(DEV)
CONFIGS = [('default', {}), ('no_prescale', {'RW_PRESCALE': '0'}),
           ('two_pass_last_layers', {'RW_PRESCALE': '0', 'RW_UP_FUSED': '0', 'RW_RGB_F4': '0'}),
           ('f22_everywhere', {'RW_PRESCALE': '0', 'RW_UP_FUSED': '0', 'RW_RGB_F4': '0', 'RW_CONV_ALGO': 'winograd'})]


def run(sync):
    outs = {}
    with torch.no_grad():
        for name, env in CONFIGS:
            os.environ.update(env)
            outs[name] = model(z)
            if sync:
                torch.cuda.synchronize()
            for k in env:
                del os.environ[k]
    torch.cuda.synchronize()
    return outs


for k in [a for a in sys.argv[3:] if '=' in a]:
    os.environ[k.split('=')[0]] = k.split('=')[1]
ref = run(True)
again = run(True)
print(json.dumps({'synced_twice_max_diff': {k: float((ref[k] - again[k]).abs().max()) for k in ref}}))
for r in range(reps):
    got = run(False)
    print(json.dumps({'rep': r, 'unsynced_vs_synced': {k: round(float((got[k] - ref[k]).abs().max()), 6) for k in ref}}))
print(json.dumps({'env': sys.argv[3:]}))
