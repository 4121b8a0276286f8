"""
`python bench.py --gpus N` without a launcher must start its ranks as CHILD processes (never exec after a GPU call),
hand on rank 0's JSON line and fail when a rank fails.  Runs the real launcher code on the CPU:
--dry-run swaps the GPU work for a sleep and RCCL for gloo.
"""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra_env=None, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, universal_newlines=True, timeout=300)


def test_plain_invocation_with_two_ranks_prints_one_json_line():
    res = _run(None, '--gpus', '2', '--steps', '3', '--warmup', '0', '--dry-run')
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['dry_run'] is True and out['value'] > 0
    # the line says who took part: two ranks, two processes, each with its own time
    assert out['ranks'] == 2 and out['backend'] == 'gloo'
    assert [r['rank'] for r in out['per_rank']] == [0, 1] and len(set(r['pid'] for r in out['per_rank'])) == 2
    assert all(r['elapsed_ms'] > 0 and r['frames'] == 3 for r in out['per_rank'])


def test_a_failing_rank_fails_the_whole_run():
    res = _run({'AMT_BENCH_DRYRUN_FAIL_RANK': '1'}, '--gpus', '2', '--steps', '2', '--warmup', '0', '--dry-run')
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]


def test_single_rank_runs_in_process():
    res = _run(None, '--steps', '2', '--warmup', '0', '--dry-run')
    assert res.returncode == 0, res.stderr[-2000:]
    assert json.loads(res.stdout.strip().splitlines()[-1])['n_gpus'] == 1


def test_default_steps_are_the_same_per_rank_for_every_n():
    sys.path.insert(0, ROOT)
    import bench
    # weak scaling: the same number of frames per rank whatever N is, and at least the 32 per rank of BASELINE.json
    # configs[4] (256 frames on 8 GPUs = --gpus 8 --steps 32)
    assert bench.parse_args(['--gpus', '8']).steps == bench.parse_args([]).steps >= 32
    assert bench.parse_args(['--gpus', '8', '--steps', '32']).steps * 8 == 256
    assert bench.parse_args([]).gpus == 1
