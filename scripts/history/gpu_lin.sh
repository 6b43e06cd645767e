#!/bin/bash
OUT=gpurun_out/lin; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "mapping or golden or hook" 2>&1 | tail -3 | tee $OUT/test.log
python - <<'PY'
import torch, math, os, sys
sys.path.insert(0, '.')
from rewriting_amd import hip
for batch in (64, 250, 1000):
    z = torch.randn(batch, 512, device='cuda'); w = torch.randn(512, 512, device='cuda'); b = torch.randn(512, device='cuda')
    for impl in ('0', '1'):
        os.environ['RW_LINEAR_IMPL'] = impl
        for _ in range(3): hip.equal_linear(z, w, b, 0.04, 1.0)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): hip.equal_linear(z, w, b, 0.04, 1.0)
        e.record(); torch.cuda.synchronize()
        print('batch %d impl %s: %.1f us' % (batch, impl, s.elapsed_time(e) / 50 * 1e3))
PY
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json
d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step']); e=d['extra']; print(e['sweep_ffhq1024_layer8']['seeds_per_s'], e['edit_horse256_layer8']['seconds_per_edit'], e['edit_horse256_layer8']['key_collect_s'], e['forward_ffhq256_b64']['images_per_s'])"
