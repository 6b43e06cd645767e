"""Wall time of each of the first forwards of the 1024^2 generator after construction (allocator / packing warm-up)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rewriting_amd import synthetic
from rewriting_amd.utils.stylegan2 import models
g = models.SeqStyleGAN2(1024, 512, 8, truncation=1.0, mconv='seq')
synthetic.randomize_(g, seed=0)
g = g.eval().to('cuda')
z = torch.randn(64, 512, device='cuda')
with torch.no_grad():
    for i in range(10):
        torch.cuda.synchronize(); t = time.time()
        img = g(z)
        torch.cuda.synchronize()
        print('forward %d: %.1f ms  reserved %.1f GB  allocs %d' % (i, (time.time() - t) * 1e3, torch.cuda.memory_reserved() / 2**30,
              torch.cuda.memory_stats()['num_device_alloc']), flush=True)
        del img
