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


# ---- the supervisor (round 5: the driver's bench of round 4 died of a GPU memory-access fault with nothing on stdout) -------------
FAULT = ('/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n[bench    1.79s] timed region: 20 steps\n'
         'Memory access fault by GPU node-2 (Agent handle: 0x6333996a6d10) on address 0x7000000ff000. Reason: Unknown.\n'
         'Fatal Python error: Aborted\n\nCurrent thread 0x00007 (most recent call first):\n  File "bench.py", line 445 in worker\n')


def _bench_module():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_fault_address_is_mapped_onto_the_workers_buffers(tmp_path):
    bench = _bench_module()
    mp = tmp_path / 'map.json'
    mp.write_text(json.dumps({'buffers': [['act S3[76x76x256 act]', 0x700000000000, 0x100000], ['wpart', 0x700000200000, 0x1000]],
                              'segments': [[0x700000000000, 0x200000]]}))
    d = bench.diagnose_fault(FAULT, str(mp))
    assert d['address'] == '0x7000000ff000' and d['reason'] == 'Unknown.'
    assert d['buffer'].startswith('act S3[76x76x256 act] + %d of' % 0xff000) and d['in_allocator_segment'] is True
    assert d['last_stage'] == 'timed region: 20 steps' and any('line 445' in ln for ln in d['python_stack'])
    d = bench.diagnose_fault(FAULT.replace('0x7000000ff000', '0x700000100800'), str(mp))       # between two buffers
    assert d['buffer'] is None and 'ends 2048 bytes before' in d['nearest_below'] and 'wpart starts' in d['nearest_above']
    d = bench.diagnose_fault('Traceback ...\nRuntimeError: boom\n', str(mp))                    # not a GPU fault
    assert 'address' not in d and 'boom' in d['stderr_tail']


@pytest.mark.parametrize('outcomes,want_rc,want_retries', [
    ([(-6, None, None), (0, 'full', 'part')], 0, 1),          # fault, then a clean run
    ([(-6, None, 'part'), (-6, None, None)], 1, 1),           # dies twice, once after the timed region
    ([(0, 'full', 'part')], 0, 0),                            # the normal case
    ([(-6, 'full', 'part')], 0, 0)])                          # measured, died at teardown
def test_supervisor_survives_a_dying_worker_and_always_prints_one_line(monkeypatch, capsys, outcomes, want_rc, want_retries):
    bench = _bench_module()
    full = {'metric': 'm', 'value': 870.0, 'unit': 'images/s', 'ms_per_step': 18.4, 'roofline': {'frac': 0.17}, 'cpu_baseline': None}
    part = {'partial': True, 'value': 869.0, 'ms_per_step': 18.41}
    calls = []

    def fake(argv, map_file, timeout=0):
        rc, f, p_ = outcomes[len(calls)]
        calls.append(argv)
        return rc, (dict(full) if f else None), (dict(part) if p_ else None), (FAULT if rc else '')
    monkeypatch.setattr(bench, 'run_worker', fake)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    argv = ['--steps', '20', '--warmup', '5', '--no-extra', '--no-cpu-baseline']
    rc = bench.supervise(bench.parse_args(argv), argv)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('{')]
    assert rc == want_rc and len(lines) == 1 and len(calls) == len(outcomes)
    d = json.loads(lines[0])
    assert d['fault_retries'] == want_retries
    if want_rc == 0:
        assert d['value'] == 870.0 and 'error' not in d
        assert ('faults' in d) == any(o[0] != 0 for o in outcomes)
    else:
        assert d['value'] == 869.0 and d['ms_per_step'] == 18.41 and 'AFTER the timed region' in d['error']      # the completed timed region survives
        assert len(d['faults']) == 2 and d['faults'][0]['address'] == '0x7000000ff000'


def test_rank0_supervisor_leaves_a_line_when_the_launcher_terminates_it(tmp_path):
    """Under a launcher a dying rank makes the launcher SIGTERM the others; rank 0's supervisor must still print ONE line with `error`
    ("under self_launch a dead rank must still produce a line")."""
    import signal
    code = ('import sys, time; sys.path.insert(0, %r); import bench\n'
            'bench.run_worker = lambda argv, map_file, timeout=0: (sys.stderr.write("READY\\n"), sys.stderr.flush(), time.sleep(60), (0, None, None, ""))[3]\n'
            'argv = ["--gpus", "2", "--steps", "2", "--warmup", "1"]\n'
            'sys.exit(bench.supervise(bench.parse_args(argv), argv))\n' % ROOT)
    env = dict(os.environ, RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', OMP_NUM_THREADS='1')
    p = subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    assert 'READY' in p.stderr.readline()           # (the supervisor has installed its handler and is waiting for its worker)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert p.returncode == 1 and len(lines) == 1, (p.returncode, out, err[-500:])
    d = json.loads(lines[0])
    assert d['value'] is None and d['n_gpus'] == 2 and 'terminated by the launcher' in d['error']


FAKE_WORKER = """import json, os, sys, time
mode = sys.argv[-1]
for i in range(200):
    sys.stderr.write("[bench %7.2fs] stage %d\\n" % (i, i))
print(json.dumps({"partial": True, "value": 1.0}), flush=True)
if mode == "hang":
    time.sleep(60)
if mode == "die":
    sys.stderr.write("Memory access fault by GPU node-2 (Agent handle: 0x1) on address 0x1000. Reason: Unknown.\\n")
    sys.stderr.flush()
    os.abort()
print(json.dumps({"value": 2.0}), flush=True)
"""


def test_run_worker_keeps_both_pipes_apart(tmp_path):
    """bench.run_worker against a stand-in worker script: every stderr line is forwarded AND kept for the diagnosis, the partial and
    the full line are told apart, a non-zero exit status comes back, a worker that hangs is killed at the timeout."""
    bench = _bench_module()
    fake = tmp_path / 'fake_bench.py'
    fake.write_text(FAKE_WORKER)
    rc, full, part, err = bench.run_worker(['ok'], str(tmp_path / 'map.json'), script=str(fake))
    assert rc == 0 and full == {'value': 2.0} and part['value'] == 1.0 and err.count('stage') == 200
    rc, full, part, err = bench.run_worker(['die'], str(tmp_path / 'map.json'), script=str(fake))
    assert rc != 0 and full is None and part['value'] == 1.0 and 'Memory access fault' in err and err.count('stage') == 200
    assert bench.diagnose_fault(err, str(tmp_path / 'nomap.json'))['address'] == '0x1000'
    rc, full, part, err = bench.run_worker(['hang'], str(tmp_path / 'map.json'), timeout=3, script=str(fake))
    assert rc != 0 and full is None and part is not None and 'killed after 3 s' in err
