"""Can the sliced model of a key-statistics sweep be captured in a HIP graph, and what does a replay save?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rewriting_amd.utils import nethook, zdataset  # noqa: E402
from rewriting_amd.utils.stylegan2.models import noise_batch_period  # noqa: E402

dev = torch.device('cuda')
g = bench.build_generator(1024, dev)
ctx = nethook.subsequence(g, upto_layer='layer8.sconv.mconv.dconv', share_weights=True)
z = zdataset.z_dataset_for_model(g, size=250)[:][0].to(dev)
with torch.no_grad(), noise_batch_period(10):
    for _ in range(2):
        ref = ctx(z).fmap.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        out = ctx(z).fmap
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 8
    zin = z.clone()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ctx(zin)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(graph):
        gout = ctx(zin).fmap
    graph.replay()
    torch.cuda.synchronize()
    print('max diff vs eager', (gout - ref).abs().max().item())
    t0 = time.perf_counter()
    for _ in range(8):
        graph.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 8
    print('eager %.3f ms  graph %.3f ms per 250-seed context forward' % (eager * 1e3, rep * 1e3))
