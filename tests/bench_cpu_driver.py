"""TEST INFRASTRUCTURE ONLY -- runs bench.py's main() with CPU tensors: the kernel wrappers replaced by the torch
stand-ins of tests/hip_emulation.py and the two places bench.py touches the HIP runtime (device selection,
synchronize) by host equivalents, so that `python bench.py --gpus N` -- self_launch -> torch.distributed.run -> N
ranks -> init_from_env -> sharded sweep / replica jobs -> one JSON line from rank 0 -> barrier -> exit codes -- can be
driven by the CPU test-suite (gloo).  bench.py reaches this file only through RW_BENCH_ENTRY, which
tests/test_bench_launcher.py sets; it is not a way to produce a benchmark number (the line it prints says
data='emulated kernels on CPU')."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402
from rewriting_amd import parallel  # noqa: E402
from tests import hip_emulation  # noqa: E402


def main():
    torch.set_num_threads(2)
    hip_emulation.install(pytest.MonkeyPatch())
    bench.bench_device = lambda local: torch.device('cpu')
    bench.device_sync = lambda: None
    init = parallel.init_from_env
    parallel.init_from_env = lambda backend=None: init(backend='gloo')
    fail_rank = os.environ.get('RW_TEST_FAIL_RANK')
    if fail_rank is not None and int(os.environ.get('RANK', '0')) == int(fail_rank):
        measure = bench.measure_sweep

        def failing(*a, **k):
            raise RuntimeError('injected failure on rank %s' % fail_rank)
        bench.measure_sweep = failing
    dumps = bench.json.dumps
    bench.json = type(sys)('json_proxy')
    bench.json.dumps = lambda obj, **k: dumps(dict(obj, data='emulated kernels on CPU') if isinstance(obj, dict) and
                                               'metric' in obj else obj, **k)
    bench.json.load = __import__('json').load
    bench.main()


if __name__ == '__main__':
    main()
