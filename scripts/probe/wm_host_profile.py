"""cProfile of one watermark variant (statistics + edit + sample set): where the host time goes."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rewriting_amd import workloads  # noqa: E402

dev = torch.device('cuda')
req = workloads.fold_request(workloads.load_request(), 1000)
v = workloads.WATERMARK_VARIANTS[0]
workloads.run_watermark_variant(v, dev, req)            # warm
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
t, stats, gw = workloads.run_watermark_variant(v, dev, req)
torch.cuda.synchronize()
pr.disable()
print(t)
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
