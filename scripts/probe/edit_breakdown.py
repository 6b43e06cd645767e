import sys, os, time, json, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from rewriting_amd.rewrite import ganrewrite
from rewriting_amd.utils import zdataset
dev = 'cuda'
g = bench.build_generator(256, dev)
zds = zdataset.z_dataset_for_model(g, size=1000)
req = json.load(open(os.path.join(ROOT, 'tests/golden/masks/recorded_horse_hat.json')))
def once():
    gw = ganrewrite.SeqStyleGanRewriter(g, zds, 8)
    gw.apply_edit(req, rank=1, niter=2001, piter=10, lr=0.05)
    torch.cuda.synchronize()
once()
t0 = time.perf_counter(); once(); print('total', time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); once(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:7000])
