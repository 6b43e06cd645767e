"""Divergence study of the solve (GPU only): HIP solver vs the oracle's explicit arithmetic in
fp32 (CPU) and fp64 (CPU) on the golden horse->hat problem, at several horizons, for both the
un-projected (it % 10 == 9) and projected (it % 10 == 0) states.  Writes gpurun_out/diverge.json.
"""
import json
import os
import sys
import time

import numpy
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R                      # noqa: E402
from rewriting_amd.rewrite import hipsolve               # noqa: E402
from rewriting_amd import synthetic                      # noqa: E402
from rewriting_amd.utils.stylegan2 import models         # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def main(niter=int(os.environ.get('RW_DIVERGE_N', '301'))):
    g = numpy.load(os.path.join(ROOT, 'tests/golden/rw_s64_l8_horsehat.npz'))
    m = models.SeqStyleGAN2(64, 512, 8, truncation=0.5, mconv='seq')
    synthetic.randomize_(m, seed=0)
    sd = m.state_dict()
    W0 = sd['layer8.sconv.mconv.dconv.weight'].clone()
    bias = sd['layer8.sconv.activate.bias'].clone()
    nw = sd['layer8.sconv.noise.weight'].clone()
    key, style, val = (torch.from_numpy(g[k]) for k in ('goal_in_fmap', 'goal_in_style', 'goal_out_fmap'))
    ctx = torch.from_numpy(g['mkey'])
    marks = sorted(set([1, 10, 11, 50, 51, 100, 101, 200, 201, 300, 301, 1000, 1001, 2000, 2001]) & set(range(1, niter + 1)))
    t0 = time.time()
    _, l32, s32 = R.insert_explicit(W0, key, style, val, bias, nw, ctx, niter=niter, snapshots=marks)
    t32 = time.time() - t0
    t0 = time.time()
    _, l64, s64 = R.insert_explicit(W0, key, style, val, bias, nw, ctx, niter=niter, snapshots=marks,
                                    dtype=torch.float64)
    t64 = time.time() - t0
    dev = 'cuda'
    Wd = W0.to(dev).clone()
    snaps = {}

    def cb(it, loss):
        pass
    solver = hipsolve.Solver(Wd, key.to(dev), style.to(dev), val.to(dev), bias.to(dev), nw.to(dev), ctx.to(dev),
                             niter, 10, 0.05, True, False)
    for it in range(niter):
        solver.step(it)                 # step + projection where due (post-projection snapshots)
        if it + 1 in marks:
            snaps[it + 1] = Wd.detach().cpu().clone()
    rows = []
    for n in marks:
        d64 = s64[n].float() - W0
        rows.append(dict(steps=n, gpu_vs_cpu32=rel(snaps[n] - W0, s32[n] - W0), gpu_vs_fp64=rel(snaps[n] - W0, d64),
                         cpu32_vs_fp64=rel(s32[n] - W0, d64),
                         loss_gpu=float(solver.losses[n - 1]), loss_cpu32=l32[n - 1], loss_fp64=l64[n - 1]))
        print(rows[-1])
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'diverge.json'), 'w') as f:
        json.dump(dict(rows=rows, cpu32_seconds=t32, fp64_seconds=t64, threads=torch.get_num_threads()), f, indent=1)


if __name__ == '__main__':
    main()
