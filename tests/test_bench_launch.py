"""`python bench.py --gpus N` must start its own ranks (reference src/train.py:46-52 spawns one process per GPU): the launch
path of bench.py on CPU -- bench.self_launch with two gloo ranks of tests/bench_sim.py (the operator simulator; bench.py itself
holds no simulator code), and the refusals: fewer GPUs than asked for, a launcher's WORLD_SIZE that contradicts --gpus."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _run(args, env_drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'), timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env['OMP_NUM_THREADS'] = '2'
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, env=env,
                          cwd=ROOT, timeout=timeout)


def test_gpus_2_launches_its_own_ranks():
    """bench.self_launch (what `python bench.py --gpus 2` calls when no launcher set WORLD_SIZE) with the CPU stand-in for the
    per-rank body: two gloo ranks, ONE JSON line back."""
    code = ('import sys; sys.path.insert(0, %r); import bench; '
            'sys.exit(bench.self_launch(2, ["--gpus", "2", "--steps", "2", "--warmup", "1"], script=%r, need_gpus=False))'
            % (ROOT, os.path.join(ROOT, 'tests', 'bench_sim.py')))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '2'
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1
    assert d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 4
    assert len(d['per_rank_ms_per_step']) == 2
    assert d['ms_per_step'] >= max(d['per_rank_ms_per_step']) - 1e-3        # the line carries the MAX over ranks
    assert d['params_equal_across_ranks'] is True                          # the ranks really exchanged gradients


def test_refuses_more_gpus_than_the_node_has():
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip('node has 64 devices')
    r = _run(['--gpus', '64', '--steps', '1', '--warmup', '0'])
    assert r.returncode == 2
    assert 'needs 64 GPUs' in r.stderr and 'usage' not in r.stderr.lower()


def test_launcher_world_size_mismatch_is_reported():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', OMP_NUM_THREADS='2')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr
