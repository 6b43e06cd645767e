timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "heavy and direct16" -p no:cacheprovider --tb=short 2>&1 | tail -8
python - <<'PY'
import os, time, torch, json
import bench
from rewriting_amd.utils import zdataset
dev = 'cuda:0'
g = bench.build_generator(1024, dev)
z = zdataset.standard_z_sample(64, 512, seed=1).to(dev)
def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.no_grad():
        for _ in range(n): img = g(z)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n, img
out = {}
ref = None
for mm, d16 in (('split', '0'), ('split', '1'), ('f32', '0')):
    os.environ['RW_MM'] = mm; os.environ['RW_MM_DIRECT16'] = d16
    run(2); dt, img = run(5)
    key = mm + ('+direct16' if d16 == '1' else '')
    out[key] = round(64 / dt, 1)
    if ref is None: ref = img.clone()
    else: out[key + '_linf_vs_split'] = float((img - ref).abs().max())
print(json.dumps(out))
PY
