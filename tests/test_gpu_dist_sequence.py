"""
BASELINE configs[4]'s code path on MORE THAN ONE rank with real grids (missing #4): `run_sequence` — frames
sharded over the ranks (reference: the plain per-frame loop of mapping/spacecraft.py:326-332, cli/convert.py:178-185),
`SequencePipeline` on every rank, the grids gathered on rank 0 — run by 2 and 4 worker processes on the box's one GPU
(gloo: RCCL wants a GPU per rank; the sharding, the pipeline and the packing / unpacking are the very same code), and rank
0's gathered grids compared

* cell for cell with `real_sequence_iss029.npz` (made by the REAL reference from the ten consecutive real headers of its
  test resources, full size), with a full-size pole frame and a frame of empty sky inserted into the sequence;
* with the real reference's pole fixtures (`pole_frame_{north,south}_fast.npz`) between ordinary, date-line and empty
  frames, incl. `frame_coordinates()` — what the receiver makes of the date-line / pole descriptors.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(case, world, out, timeout):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_sequence_worker.py'), case, str(r), str(world), str(port), out],
                              env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors='replace'))
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)
    return np.load(out)


def lon_diff(a, b, lat=None):
    """|a - b| on the circle; with `lat` as arc length on the parallel (longitude is ill-conditioned next to a pole: the
    pole frames' cells there differ by degrees of longitude at 1e-13 deg of arc)"""
    d = np.abs(a - b)
    d = np.minimum(d, 360 - d)
    return d if lat is None else d * np.cos(np.deg2rad(lat))


def test_worker_sequences_are_well_formed():
    """CPU: the frame lists the ranks build (same on every rank) and their shards."""
    sys.path.insert(0, HERE)
    from _sequence_worker import make_frames
    from auromat_amd.sequence import shard
    frames, w, h, ppd, kinds = make_frames('small')
    assert (w, h, ppd) == (200, 160, 8) and len(frames) == 9 == len(kinds)
    assert kinds.count('pole:north') == 1 and kinds.count('pole:south') == 1 and kinds.count('empty') == 1
    assert all(f[3].shape == (h, w, 3) for f in frames)
    for world in (2, 4):
        parts = [shard(len(frames), r, world) for r in range(world)]
        assert sum(parts, []) == list(range(9)) and min(len(p) for p in parts) >= 2


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2, 4])
def test_run_sequence_on_real_consecutive_frames_over_ranks(world, tmp_path):
    """12 full-size frames (the ten real headers + a pole frame + empty sky) over `world` ranks: rank 0 holds the
    reference's grids."""
    sys.path.insert(0, HERE)
    from _sequence_worker import make_frames
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.resample import grid_coordinates
    got = run_ranks('real', world, str(tmp_path / 'gathered.npz'), timeout=500)
    z = load_golden('real_sequence_iss029.npz')
    frames, w, h, ppd, kinds = make_frames('real')
    assert len(frames) == 12 and sum(got['shard_sizes']) == 12
    empty = kinds.index('empty')
    assert got['failed'].tolist() == [empty]
    assert got['indices'].tolist() == [k for k in range(12) if k != empty]
    for k, kind in enumerate(kinds):
        if kind == 'empty':
            continue
        mean, count = got['mean_%d' % k], got['count_%d' % k]
        lat_c, lon_c = got['lat_c_%d' % k], got['lon_c_%d' % k]
        pole, disc, magnetic = got['flags_%d' % k]
        assert float(got['altitude_%d' % k]) == 110.0 and not magnetic
        if kind.startswith('real:'):
            i = int(kind[5:])
            want = z['out_data_%d' % i]
            assert mean.shape == want.shape, (k, mean.shape, want.shape)
            mask = np.isnan(mean[..., 0])
            assert np.array_equal(mask, count == 0)
            assert np.array_equal(mask, np.isnan(want[..., 0])), k
            ok = ~mask
            assert np.array_equal(mean[..., :3][ok], want[..., :3][ok]), k                  # exact integer sums / counts
            assert np.max(np.abs(mean[..., 3][ok] - want[..., 3][ok])) < 1e-9, k           # elevation (fixed-point sums)
            assert not pole and not disc
            # the receiver's cell centres are the midpoints of the reference's cell corners
            wl, wo = z['out_lat_%d' % i], z['out_lon_%d' % i]
            assert np.max(np.abs(lat_c - 0.25 * (wl[:-1, :-1] + wl[1:, :-1] + wl[:-1, 1:] + wl[1:, 1:]))) < 1e-9
            assert np.max(np.abs(lon_c - 0.25 * (wo[:-1, :-1] + wo[1:, :-1] + wo[:-1, 1:] + wo[1:, 1:]))) < 1e-9
        else:
            # the full-size pole frame: one process, two-pass plan (pinned to the reference at 200 x 160, test_pole_frames.py)
            hdr, cam, t, img = frames[k]
            pipe = FramePipeline(w, h)
            ref = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=ppd, fuse=False)
            assert pipe.last_plan == 'two-pass' and ref['contains_pole'] and pole
            assert np.array_equal(mean, ref['mean'], equal_nan=True) and np.array_equal(count, ref['count'])
            c = grid_coordinates(ref)
            assert np.max(np.abs(lat_c - c['lat_c'])) < 1e-9 and np.max(lon_diff(lon_c, c['lon_c'], lat_c)) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2, 4])
def test_run_sequence_with_pole_and_dateline_frames_over_ranks(world, tmp_path):
    """Nine small frames over `world` ranks: the reference's pole fixtures, date-line frames and a frame of empty sky
    between ordinary ones; the descriptors' flags let the receiver rebuild the reference's cell coordinates."""
    sys.path.insert(0, HERE)
    from _sequence_worker import make_frames
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.resample import grid_coordinates
    got = run_ranks('small', world, str(tmp_path / 'gathered.npz'), timeout=300)
    frames, w, h, ppd, kinds = make_frames('small')
    empty = kinds.index('empty')
    assert got['failed'].tolist() == [empty]
    assert got['indices'].tolist() == [k for k in range(len(frames)) if k != empty]
    seen = set()
    for k, kind in enumerate(kinds):
        if kind == 'empty':
            continue
        mean, count = got['mean_%d' % k], got['count_%d' % k]
        lat_c, lon_c = got['lat_c_%d' % k], got['lon_c_%d' % k]
        pole, disc, _ = got['flags_%d' % k]
        if kind.startswith('pole:'):
            zp = load_golden('pole_frame_%s_fast.npz' % kind[5:])
            want = zp['out_data']
            assert pole and mean.shape == want.shape
            mask = np.isnan(mean[..., 0])
            assert np.array_equal(mask, count == 0)
            assert np.array_equal(mask, np.isnan(want[..., 0]))
            ok = ~mask
            assert ok.sum() > 2000
            assert np.array_equal(mean[..., :3][ok], want[..., :3][ok])
            assert np.max(np.abs(mean[..., 3][ok] - want[..., 3][ok])) < 1e-9
            assert np.max(np.abs(lat_c - zp['out_lat_c'])) < 1e-9 and np.max(lon_diff(lon_c, zp['out_lon_c'], lat_c)) < 1e-9
        else:
            hdr, cam, t, img = frames[k]
            ref = FramePipeline(w, h).run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=ppd, fuse=False)
            assert bool(disc) == bool(ref['contains_discontinuity']) == (kind == 'dateline') and not pole
            assert np.array_equal(mean, ref['mean'], equal_nan=True) and np.array_equal(count, ref['count'])
            c = grid_coordinates(ref)
            assert np.max(np.abs(lat_c - c['lat_c'])) < 1e-9 and np.max(lon_diff(lon_c, c['lon_c'], lat_c)) < 1e-9
            if kind == 'dateline':
                assert lon_c.min() < -170 and lon_c.max() > 170          # true longitudes on both sides of the date line
        seen.add(kind.split(':')[0])
    assert seen == {'pole', 'dateline', 'plain'}
