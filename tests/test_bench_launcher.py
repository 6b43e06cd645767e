"""bench.py's own multi-rank entry, end to end on CPU (world size 2, gloo, kernel stand-ins): `python bench.py --gpus 2`
re-executes itself under torch.distributed.run (self_launch), every rank goes through main() -> init_from_env ->
the workload -> the collective -> rank 0's JSON line -> barrier.  What is asserted: the launcher starts exactly
--gpus ranks and the process group says so, the sharded sweep and the replica job run through bench.py's code (not
through direct calls of tally / workloads as the other distributed tests do), the timing is the max over ranks, and
a failure on a rank OTHER than 0 reaches the exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, 'tests', 'bench_cpu_driver.py')


LAST = {}


def _strict(text):
    """json.loads that refuses NaN / Infinity (what a strict consumer does)."""
    def bad(c):
        raise ValueError('non-finite constant %s in the bench line' % c)
    return json.loads(text, parse_constant=bad)


def _run(args, env_extra=None, timeout=900):
    """-> (exit code, THE line = the LAST line of stdout parsed strictly or None, stderr).  The long form goes to the
    file RW_BENCH_DETAIL names (LAST['detail'] after the call)."""
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix='rw_bench_'), 'bench_detail.json')
    env = dict(os.environ, RW_BENCH_ENTRY=DRIVER, OMP_NUM_THREADS='2', RW_BENCH_DETAIL=detail)
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = p.stdout.strip().splitlines()
    LAST.clear()
    out = None
    if lines and lines[-1].startswith('{'):
        # the contract the driver reads: the LAST line of stdout is one compact strict-JSON object
        assert len(lines[-1]) < 4096, len(lines[-1])
        out = _strict(lines[-1])
        assert sum(l.startswith('{') for l in lines) == 1, 'more than one JSON line on stdout'
        assert sum(len(l) for l in lines) < 8192, 'stdout of a bench run must stay small'
        for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                    'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
            assert key in out, key
        assert len(out['dtype']) <= 80 and isinstance(out['config'].get('workload'), str)
        assert all(not isinstance(v, (dict, list)) for v in out['config'].values())
        if os.path.isfile(detail):
            with open(detail) as f:
                LAST['detail'] = _strict(f.read())
    return p.returncode, out, p.stderr


def test_bench_self_launch_runs_the_sharded_sweep_on_two_ranks():
    rc, out, err = _run(['--gpus', '2', '--workload', 'sweep', '--size', '32', '--layer', '6', '--seeds', '40',
                         '--steps', '1', '--warmup', '0'])
    assert rc == 0, err[-2000:]
    assert out['n_gpus'] == 2 and (out['rccl']['backend'], out['rccl']['world_size']) == ('gloo', 2)
    # the collective is timed on its own and the launch counts are in the line (a SCALE line can be audited)
    assert out['config']['allreduce_ms'] > 0 and out['config']['launches'] == 2 and out['config']['launches_this_rank'] == 1
    assert out['unit'] == 'seeds/sec' and out['value'] > 0 and out['scaling'] == 'strong'
    assert out['steps'] == 1 and out['warmup'] == 0 and out['ms_per_step'] > 0
    assert out['data'] == 'emulated kernels on CPU'               # the driver marks its line: never a bench number
    # 40 seeds, launches of 20 (>= one launch per rank): two launches, one per rank
    assert 'launches of 20 seeds' in out['config']['workload'], out['config']['workload']


def test_bench_self_launch_runs_the_watermark_replicas_on_two_ranks():
    rc, out, err = _run(['--gpus', '2', '--workload', 'watermark', '--seeds', '20', '--wm-size', '32', '--niters', '3',
                         '--steps', '1', '--warmup', '0'])
    assert rc == 0, err[-2000:]
    assert out['n_gpus'] == 2 and out['rccl']['world_size'] == 2 and out['scaling'] == 'strong'
    assert 'variants' not in out['config']                      # tables live in the long form, not in THE line
    long_form = LAST['detail']
    assert long_form['value'] == out['value'] and long_form['rccl']['world_size'] == 2
    variants = long_form['config']['variants']
    assert sorted(variants) == ['gandissect-30', 'gandissect-60', 'none', 'ours-30-2', 'ours-60-2']   # both ranks' shares
    assert all(v['images'] == 20 for v in variants.values())
    assert set(long_form['config']['frechet_vs_unedited_pooled_rgb']) == set(variants)
    assert long_form['config']['frechet_vs_unedited_pooled_rgb']['none'] < 1e-9


def test_a_failure_on_rank_1_reaches_the_exit_code():
    rc, out, err = _run(['--gpus', '2', '--workload', 'sweep', '--size', '32', '--layer', '6', '--seeds', '40',
                         '--steps', '1', '--warmup', '0'], env_extra=dict(RW_TEST_FAIL_RANK='1'), timeout=600)
    assert rc != 0
    assert out is None                                         # rank 0 never printed a result line
    assert 'injected failure on rank 1' in err


def test_launcher_refuses_a_world_size_that_is_not_gpus():
    env = dict(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    rc, out, err = _run(['--gpus', '2', '--workload', 'sweep', '--size', '32', '--layer', '6', '--seeds', '20'],
                        env_extra=dict(env, RW_BENCH_ENTRY=''), timeout=120)
    assert rc != 0 and out is None and '--gpus 2 but the launcher started 1 ranks' in err


def test_the_headline_line_of_a_recorded_run_is_compact():
    """bench.compact_line over the long form of a recorded N=1 run of the default workload (round 5's, 21 KB as it was
    printed then): THE line carries the contract keys, `roofline` and `cpu_baseline` with numbers, an `extra` of numbers
    only, and fits in 4 KB."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, 'profiles', 'r05hd_bench.json')) as f:
        long_form = _strict(f.read().strip().splitlines()[-1])
    line = json.dumps(bench.compact_line(long_form), allow_nan=False, separators=(',', ':'))
    assert len(line) < 4096, len(line)
    out = _strict(line)
    assert out['metric'] == 'images/sec StyleGANv2-1024 fwd' and out['value'] == long_form['value']
    roof = out['roofline']
    assert set(roof) >= {'bound', 'kernel', 'launches', 'avg_launch_us', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert roof['bound'] in ('hbm', 'mfma') and abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    assert set(out['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind'} and out['cpu_baseline']['kind'] in ('port', 'reference')
    assert set(out['step']) >= {'hbm_frac', 'hbm_bytes_algorithmic', 'hbm_bytes_pmc'}
    assert out['parity']['ok'] is True and out['parity']['linf'] < 1e-3
    assert all(v is None or isinstance(v, (int, float, bool)) for v in out['extra'].values()), out['extra']
    assert len(out['dtype']) <= 80
    assert max(len(v) for v in out['config'].values() if isinstance(v, str)) <= 64
