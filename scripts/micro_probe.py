"""Step time of the 1024^2 generator forward (GPU only) as a function of RW_MICRO_BATCH (SeqStyleGAN2._forward_micro):
the high-resolution steps run on slices of the batch so that hand-offs between kernels stay in the memory-side
cache.  Writes gpurun_out/micro_probe.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rewriting_amd import synthetic                       # noqa: E402
from rewriting_amd.utils.stylegan2 import models          # noqa: E402


def main(batch=int(os.environ.get('RW_BATCH', '64')), size=int(os.environ.get('RW_SIZE', '1024')), iters=4):
    dev = 'cuda'
    g = models.SeqStyleGAN2(size, 512, 8, truncation=1.0, mconv='seq')
    synthetic.randomize_(g, seed=0)
    g = g.eval().to(dev)
    z = torch.randn(batch, 512, device=dev)
    specs = os.environ.get('RW_SPECS', '0,8:256,4:256,2:256,4:512,2:512,1:512,2:1024,1:1024,4:128,8:128').split(',')
    rows = []
    ref = None
    for spec in specs:
        os.environ['RW_MICRO_BATCH'] = spec
        with torch.no_grad():
            img = g(z)
            torch.cuda.synchronize()
            if ref is None:
                ref = img[::9, :, ::16, ::16].clone()
            diff = (img[::9, :, ::16, ::16] - ref).abs().max().item()
            del img
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                g(z)
            e.record()
            torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        rows.append(dict(spec=spec, ms_per_step=round(ms, 3), images_per_s=round(batch / ms * 1e3, 1),
                         max_diff_vs_one_launch=diff, peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 2)))
        torch.cuda.reset_peak_memory_stats()
        print(rows[-1], flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', os.environ.get('RW_OUT', 'micro_probe.json')), 'w') as f:
        json.dump(dict(batch=batch, size=size, rows=rows), f, indent=1)


if __name__ == '__main__':
    main()
