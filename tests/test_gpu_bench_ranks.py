"""
The N-rank form of bench.py rehearsed on ONE GPU: `python bench.py --gpus N` starts its ranks itself (a child
`torch.distributed.run`, exactly what the driver's launcher does), every rank shards the synthetic sequence, the ranks agree on
a gather capacity in the warm-up, rank 0 gathers all grids inside the timed region, the time is the maximum over the ranks and
rank 0 prints the one JSON line.  RCCL refuses two ranks on one device, so the rehearsal takes the collectives through gloo
(AMT_BENCH_BACKEND=gloo) and puts every rank on cuda:0 (AMT_BENCH_ONE_GPU=1); everything else is the code of the real run.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world,steps', [(2, 8), (4, 5)])
def test_bench_with_several_ranks_on_one_gpu(world, steps):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(AMT_BENCH_BACKEND='gloo', AMT_BENCH_ONE_GPU='1')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', str(steps), '--warmup', '2',
                          '--cpu-rows', '0', '--no-variants'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == world and out['steps'] == steps and out['warmup'] == 2
    assert out['scaling'] == 'weak' and out['higher_is_better'] is True
    # whole-job value: all ranks' frames over the slowest rank's time
    npx = 4240 * 2832
    assert abs(out['value'] - world * steps * npx / 1e6 / (out['ms_per_step'] * steps * 1e-3)) < 1e-6 * out['value']
    assert out['config']['workload'].startswith('configs[')
    assert out['roofline']['frac'] > 0 and out['roofline']['bound'] == 'hbm'
    # the value is that of the median of seven timed regions (each: barrier, K steps, gather, barrier; maximum over the ranks)
    assert out['repeats'] == 7 and len(out['regions_ms']) == 7
    assert out['ms_per_step_min'] <= out['ms_per_step'] <= out['ms_per_step_max']
    assert sorted(out['regions_ms'])[3] == pytest.approx(out['ms_per_step'] * steps, rel=1e-9)
    assert out['config']['frame_loop'].startswith('library')
    assert out['config']['frames_total'] == world * steps and out['config']['single_pass_frames'] == steps
    # the line describes its ranks: N of them reported, each with its device, its own times and what it sent
    assert out['ranks'] == world and out['backend'] == 'gloo'
    per = out['per_rank']
    assert [r['rank'] for r in per] == list(range(world)) and len(set(r['pid'] for r in per)) == world
    for r in per:
        assert r['frames'] == steps and r['single_pass_frames'] == steps
        assert r['device']['name'] and ('pci_bus_id' in r['device'] or 'uuid' in r['device'])
        assert 0 < r['process_ms'] <= r['elapsed_ms'] and r['gather_ms'] >= 0 and r['kernel_us_per_frame'] > 50
        assert r['payload_bytes'] > 0 and r['gather_bytes'] >= r['payload_bytes']
        assert r['elapsed_ms'] <= out['timed_region_ms']['max_over_ranks'] + 1e-6
    # (the rehearsal puts every rank on the box's one GPU: one distinct device; the driver's run must show N)
    assert out['distinct_devices'] == 1 and out['gather_bytes_received'] == sum(r['gather_bytes'] for r in per)
    # disjoint blocks of the synthetic sequence
    firsts = sorted(r['first_frame'] for r in per)
    assert all(b - a >= steps for a, b in zip(firsts, firsts[1:]))


def test_n_rank_line_carries_the_upload_variant():
    """Without --no-variants the N-rank line also holds `variants.upload`: the same job with every rank streaming its frames'
    images from its own page-locked host buffers (what a real 256-frame run does), with the PCIe rate each rank reached."""
    world, steps = 2, 6
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(AMT_BENCH_BACKEND='gloo', AMT_BENCH_ONE_GPU='1')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', str(steps), '--warmup', '2',
                          '--cpu-rows', '0'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == world and out['config']['frames_total'] == world * steps
    up = out['variants']['upload']
    assert up['frames_per_rank'] == steps and up['single_pass_frames'] == steps
    assert len(up['pcie_GBs_per_rank']) == world and all(0.5 < v < 70 for v in up['pcie_GBs_per_rank'])
    assert abs(up['pcie_GBs_total'] - sum(up['pcie_GBs_per_rank'])) < 1e-9
    npx = 4240 * 2832
    assert abs(up['Mpixels_per_s'] - world * steps * npx / 1e6 / (up['ms_per_frame'] * steps * 1e-3)) < 1e-6 * up['Mpixels_per_s']
    # PCIe-inclusive: slower than the HBM-resident figure of the same line
    assert up['Mpixels_per_s'] < out['value']
    # through the library's frame loop, and only the rows of an image that can be binned cross the link
    assert up['frame_loop'].startswith('library') and up['repeats'] == 7
    assert 0 < up['uploaded_bytes_per_frame'] < 0.8 * up['image_bytes_per_frame']


def test_a_rank_that_fails_in_the_timed_region_ends_the_run():
    """One of two ranks' pipelines raises in the middle of a timed region (AMT_BENCH_FAIL_RANK: a frame dated beyond the IGRF table,
    refused by the library's frame loop).  The rank still takes part in the gather and in the closing fence; there every rank
    learns of it: the run ends with a non-zero code and no result line, well inside the collectives' timeout."""
    import time
    world, steps = 2, 6
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(AMT_BENCH_BACKEND='gloo', AMT_BENCH_ONE_GPU='1', AMT_BENCH_FAIL_RANK='1', AMT_DIST_TIMEOUT_S='120')
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', str(steps), '--warmup', '2',
                          '--cpu-rows', '0', '--no-variants', '--magnetic'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=300)
    # (--magnetic: the MLat / MLT grid needs the IGRF dipole of the frame's date, which amt_run_push refuses beyond the table)
    assert res.returncode != 0 and time.time() - t0 < 110, res.stderr[-3000:]
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert 'IGRF' in res.stderr and 'SequenceError' in res.stderr, res.stderr[-3000:]
