"""Times the 2001-iteration solve of a 512 x 512-channel stride-1 layer on a 5 x 8 key crop (the horse->hat edit's
shape) in the one-launch kernel and in the step kernels (HIP graph), and reports their agreement."""
import json
import os
import sys
import time

import numpy
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rewriting_amd.rewrite import hipsolve  # noqa: E402

DEV = 'cuda'


def main():
    out = {}
    for (O, I, h, w) in [(512, 512, 5, 8), (512, 512, 6, 8), (256, 256, 8, 8), (128, 256, 8, 12)]:
        rs = numpy.random.RandomState(0)
        W0 = torch.from_numpy(rs.randn(1, O, I, 3, 3).astype('float32')).to(DEV)
        key = torch.from_numpy(rs.randn(1, I, h, w).astype('float32')).to(DEV)
        style = torch.from_numpy((1 + 0.3 * rs.randn(1, I)).astype('float32')).to(DEV)
        val = torch.from_numpy(rs.randn(1, O, h, w).astype('float32')).to(DEV)
        bias = torch.from_numpy((0.1 * rs.randn(O)).astype('float32')).to(DEV)
        nw = torch.tensor([0.1], device=DEV)
        ctx = torch.linalg.qr(torch.from_numpy(rs.randn(I, 1).astype('float32')))[0].t().contiguous().to(DEV)
        row = {}
        res = {}
        for mode in ('one_launch', 'step'):
            os.environ['RW_SOLVE_ONE_LAUNCH'] = '1' if mode == 'one_launch' else '0'
            best = 1e9
            for rep in range(3):
                Wd = W0.clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s = hipsolve.run(Wd, key, style, val, bias, nw, ctx, niter=2001, piter=10, lr=0.05)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            row[mode + '_ms'] = best * 1e3
            row[mode + '_us_per_iter'] = best * 1e6 / 2001
            res[mode] = (Wd, s.losses.clone())
        d = (res['one_launch'][0] - res['step'][0]).norm() / (res['step'][0] - W0).norm()
        row['rel_delta_2001'] = d.item()
        row['loss_last'] = [res['one_launch'][1][-1].item(), res['step'][1][-1].item()]
        out['%dx%d_%dx%d' % (O, I, h, w)] = row
        print(O, I, h, w, json.dumps(row), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/solve_bench.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
