"""Where the time of one rank-1 edit goes (GPU only): statistics sweep, ZCA, goal construction, key, solve."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from rewriting_amd.rewrite import ganrewrite            # noqa: E402
from rewriting_amd.utils import zdataset                # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    g = bench.build_generator(256, dev)
    zds = zdataset.z_dataset_for_model(g, size=1000)
    with open(os.path.join(ROOT, 'tests', 'golden', 'masks', 'recorded_horse_hat.json')) as f:
        req = json.load(f)

    def tick():
        torch.cuda.synchronize()
        return time.perf_counter()
    for rep in range(5):
        t = [tick()]
        gw = ganrewrite.SeqStyleGanRewriter(g, zds, 8)
        t.append(tick())
        obj_acts, _, obj_area, _ = gw.object_from_selection(*req['object'])
        t.append(tick())
        goal_in, goal_out, _, _ = gw.paste_from_selection(req['paste'][0], req['paste'][1], obj_acts, obj_area)
        t.append(tick())
        mkey = gw.multi_key_from_selection(req['key'], rank=1)
        t.append(tick())
        gw.insert(goal_in, goal_out, mkey, niter=2001, piter=10, lr=0.05)
        t.append(tick())
        names = ['rewriter init (sweep + zca)', 'object_from_selection', 'paste_from_selection',
                 'multi_key_from_selection', 'insert (2001 steps)']
        t0 = tick(); ganrewrite.zca_from_cov(gw.c_matrix); t1 = tick()
        print('  of which zca_from_cov %.1f ms' % ((t1 - t0) * 1e3))
        print(' | '.join('%s %.1f ms' % (n, (b - a) * 1e3) for n, a, b in zip(names, t, t[1:])))


if __name__ == '__main__':
    main()
