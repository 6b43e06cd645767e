"""Experiment: one batch-32 forward vs two batch-16 forwards issued on two HIP streams."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
from rewriting_amd.utils import zdataset               # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    g = bench.build_generator(1024, dev)
    B = int(os.environ.get('RW_BATCH', '32'))
    z = zdataset.standard_z_sample(B, 512, seed=1).to(dev)
    parts = int(os.environ.get('RW_PARTS', '2'))
    zs = list(z.chunk(parts))
    streams = [torch.cuda.Stream() for _ in range(parts)]

    def whole():
        with torch.no_grad():
            g(z)

    def split():
        main_s = torch.cuda.current_stream()
        for s, zz in zip(streams, zs):
            s.wait_stream(main_s)
            with torch.cuda.stream(s), torch.no_grad():
                g(zz)
        for s in streams:
            main_s.wait_stream(s)

    for name, fn in (('whole', whole), ('split%d' % parts, split)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        print(name, 'ms/step %.2f' % (dt * 1e3), 'img/s %.1f' % (B / dt))


if __name__ == '__main__':
    main()
