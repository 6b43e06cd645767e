"""Refreshes profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic / step.hbm_bytes_pmc) from one tagged counter run:
    python scripts/pmc_refresh.py <tag> gpurun_out/<tag>/pmc_summary.json gpurun_out/<tag>/bench_detail.json kernel [kernel ...]
Entries of the named kernels (workload ffhq1024, batch 64) are replaced, the step figure is replaced (the old one moves to
steps_earlier_runs).  Raw FETCH_SIZE + WRITE_SIZE bytes (not doubled: the window gathers are 140 - 160-byte row segments, not wide
streaming reads; the doubled figure is kept beside it as the upper bound)."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, summary, detail = sys.argv[1:4]
kernels = sys.argv[4:]
S = json.load(open(summary))
D = json.load(open(detail))
per = D.get('roofline', {}).get('per_kernel', {})
path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
T = json.load(open(path))
dst = 'profiles/%s_pmc_summary_b64.json' % tag
shutil.copy(summary, os.path.join(ROOT, dst))
src = ('%s (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace, one pass per counter, bench.py --steps 2 --warmup 1 --no-extra '
       '--no-cpu-baseline, the default kernel selection: scripts/gpu_r05_measure.sh %s "prof pmc"); raw FETCH_SIZE + WRITE_SIZE bytes averaged over '
       'the launches of the kernel name (not doubled; the doubled figure is the upper bound)' % (dst, tag))
for k in kernels:
    r = S[k]
    alg = None
    if k in per and per[k].get('launches'):
        alg = round(per[k]['hbm_gbs'] * 1e9 * per[k]['ms'] * 1e-3 / per[k]['launches'])
    e = dict(kernel=k, workload='ffhq1024', batch=64, hbm_bytes_per_launch=round(r['hbm_bytes_per_launch_raw']),
             hbm_bytes_per_launch_upper_bound_fetch_x2=round(r['hbm_bytes_per_launch_fetch_x2']),
             fetch_kib_per_launch=r['FETCH_SIZE_KiB_per_launch'], write_kib_per_launch=r['WRITE_SIZE_KiB_per_launch'],
             launches_sampled=r['FETCH_SIZE_launches'], algorithmic_bytes_per_launch=alg, source=src,
             note='average over the launches of this name in a forward')
    T['entries'] = [x for x in T['entries'] if not (x['kernel'] == k and x['workload'] == 'ffhq1024' and x['batch'] == 64)] + [e]
step = S.get('__step__')
if step:
    old = [x for x in T['steps'] if x['workload'] == 'ffhq1024' and x['batch'] == 64]
    T.setdefault('steps_earlier_runs', []).extend(old)
    T['steps'] = [x for x in T['steps'] if x not in old] + [dict(
        workload='ffhq1024', batch=64, hbm_bytes_per_step=round(step['hbm_bytes_per_step_raw']), forwards_sampled=step['forwards'],
        algorithmic_bytes_per_step=D.get('step', {}).get('hbm_bytes_algorithmic'),
        matrix_mode='split, direct sums on layers 10-17 with the ToRGB channel sums in their epilogue; upsampling layers 9 - 17 in the fused '
                    'kernels of rw_tconv.hip (round 6 forms)', source=src + '; sum over EVERY kernel of the forward, divided by the forwards in the trace')]
json.dump(T, open(path, 'w'), indent=1)
print('refreshed', [k for k in kernels], 'step', step)
