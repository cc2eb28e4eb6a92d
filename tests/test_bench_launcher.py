"""bench.py starts its own ranks: `python bench.py --gpus N` with N > 1 and no
launcher around it (the form the driver uses for the 1-GPU line) spawns N child
processes with the torchrun environment before anything touches the GPU,
forwards rank 0's JSON line and fails as a whole when a rank fails.  CPU only:
--launch-only children print their view of the launch and exit."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _env():
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def test_bench_launches_its_own_ranks():
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '4', '--launch-only'],
                         capture_output=True, text=True, timeout=120, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    # stdout: rank 0's line and nothing else (the driver parses it)
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, lines
    r0 = json.loads(lines[0])
    assert r0['rank'] == 0 and r0['world'] == 4 and r0['gpus'] == 4
    assert r0['master'].startswith('127.0.0.1:')
    others = sorted(json.loads(line.split('] ', 1)[1])['rank']
                    for line in out.stderr.splitlines() if line.startswith('[rank '))
    assert others == [1, 2, 3]
    ranks = [json.loads(line.split('] ', 1)[1]) for line in out.stderr.splitlines()
             if line.startswith('[rank ')]
    assert all(r['local_rank'] == r['rank'] and r['master'] == r0['master'] for r in ranks)


def test_bench_under_a_launcher_does_not_launch_again():
    env = dict(_env(), RANK='1', LOCAL_RANK='1', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
               MASTER_PORT='29999')
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--launch-only'],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip())['rank'] == 1      # this process IS rank 1


def test_a_failing_rank_ends_the_run_with_its_exit_code():
    """No GPU here: every rank stops at its device check (before
    init_process_group) with a message; the parent reports the failure, stops
    the other ranks and returns non-zero -- no hang."""
    t0 = time.time()
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '1',
                          '--warmup', '0'], capture_output=True, text=True, timeout=300, env=_env())
    import torch
    if torch.cuda.device_count() >= 2:      # (a multi-GPU box: the run is real)
        return
    assert out.returncode != 0
    assert 'needs GPU' in out.stderr and '2-rank run failed' in out.stderr, out.stderr[-2000:]
    assert out.stdout.strip() == ''
    assert time.time() - t0 < 240
