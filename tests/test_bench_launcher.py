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


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ, RW_BENCH_ENTRY=DRIVER, OMP_NUM_THREADS='2')
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


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
    variants = out['config']['variants']
    assert sorted(variants) == ['gandissect-30', 'gandissect-60', 'none', 'ours-30-2', 'ours-60-2']   # both ranks' shares
    assert all(v['images'] == 20 for v in variants.values())
    assert set(out['config']['frechet_vs_unedited_pooled_rgb']) == set(variants)
    assert out['config']['frechet_vs_unedited_pooled_rgb']['none'] < 1e-9


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
