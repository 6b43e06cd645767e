"""One erase variant of the watermark job (church-256 architecture, layer 6, 2 x 2001-step erase solves on whole 16 x 16
maps) -- the solver shape whose key crop does not fit the LDS.  For rocprofv3: bash scripts/gpu_erase_prof.sh <tag>."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rewriting_amd import workloads            # noqa: E402
from rewriting_amd.rewrite import hipsolve    # noqa: E402

dev = torch.device('cuda', 0)
req = workloads.fold_request(workloads.load_request(), 1000)
for rep in range(int(os.environ.get('REPS', '2'))):
    t, _, _ = workloads.run_watermark_variant(workloads.WATERMARK_VARIANTS[0], dev, req, sample_size=1000,
                                              callback=os.environ.get('CALLBACK', 'loss_only'))
    print(json.dumps(dict(edit_s=round(t['edit_s'], 4), us_per_iteration=round(t['edit_s'] / 4002 * 1e6, 1),
                          solver=dict(hipsolve.LAST))))
