"""
ONE frame with its rows spread over several ranks (the alternative for a single very large frame):
every rank georeferences its band of rows, the ranks agree on the bounding box, bin on the common grid and all-reduce
the integer accumulators (auromat_amd.sequence.resample_frame_sharded).  The result must be the one a single
process computes for the whole frame: grids cell for cell, per-pixel arrays to rounding.

The ranks are processes on the one GPU of the box over gloo (RCCL wants one GPU per rank); the collectives are the
same calls.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_row_bands_cover_the_frame_in_whole_chunks():
    from auromat_amd.sequence import row_band
    for h in (16, 340, 2832, 2833, 47):
        for world in (1, 2, 3, 8):
            if (h + 15) // 16 < world:
                continue
            bands = [row_band(h, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == h
            for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
                assert a1 == b0 and a1 % 16 == 0 and a1 > a0
            sizes = [(b - a + 15) // 16 for a, b in bands]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize('case,world,mode', [('iss030', 2, 'fast'), ('iss030', 3, 'exact'), ('dateline', 2, 'fast'),
                                             ('pole', 3, 'fast')])
def test_rows_of_one_frame_over_several_ranks(case, world, mode, tmp_path):
    sys.path.insert(0, HERE)
    from _sharded_worker import make_case
    from auromat_amd.pipeline import FramePipeline
    port = free_port()
    out = str(tmp_path / 'rank%d.npz')
    env = dict(os.environ)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_sharded_worker.py'), case, str(r), str(world),
                               str(port), out, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors='replace'))
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)

    hdr, cam, t, img = make_case(case)
    w, h = hdr['IMAGEW'], hdr['IMAGEH']
    pipe = FramePipeline(w, h)
    ref = pipe.run(hdr, 110, cam, t, img=img, fast=mode == 'fast', min_elevation=10, pxPerDeg=8, fuse=False)
    whole = pipe.host_arrays()
    if case == 'pole':
        assert ref['contains_pole']
    if case == 'dateline':
        assert ref['contains_discontinuity']
    rows = 0
    for r in range(world):
        z = np.load(out % r)
        # every rank holds the complete grids, identical to the single-process result
        assert bool(z['contains_pole']) == bool(ref['contains_pole'])
        assert bool(z['contains_discontinuity']) == bool(ref['contains_discontinuity'])
        for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
            assert z[k].shape == ref[k].shape, (k, z[k].shape, ref[k].shape)
            assert np.array_equal(z[k], ref[k], equal_nan=True), (r, k)
        # ... and its band of the per-pixel arrays (the band's rows are counted from y0: rounding-level differences)
        y0, y1 = int(z['y0']), int(z['y1'])
        rows += y1 - y0
        for k, name in (('band_lat_c', 'lat_c'), ('band_lon_c', 'lon_c'), ('band_elev', 'elev')):
            a, b = z[k], whole[name][y0:y1]
            assert np.array_equal(np.isnan(a), np.isnan(b)), (r, name)
            assert np.nanmax(np.abs(a - b), initial=0) < 1e-10, (r, name)
        a, b = z['band_lat'], whole['lat'][y0:y1 + 1]
        assert a.shape == b.shape and np.nanmax(np.abs(a - b), initial=0) < 1e-10
    assert rows == h


@pytest.mark.gpu
def test_one_band_over_rccl():
    """The same two collectives on device tensors over "nccl" (= RCCL), one rank: all_gather of the box reduction,
    all_reduce(sum) of the int64 accumulators; a pole frame, so that the rotated box is exchanged as well."""
    import torch
    import torch.distributed as dist
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.sequence import resample_frame_sharded
    from auromat_amd.synthetic import frame_image, pole_frame
    w, h = 400, 320
    hdr, cam, t = pole_frame(w, h)
    img = frame_image(w, h, seed=3)
    ref = FramePipeline(w, h).run(hdr, 110, cam, t, img=img, min_elevation=10, pxPerDeg=8, fuse=False)
    store = dist.TCPStore('127.0.0.1', free_port(), 1, True)
    dist.init_process_group('nccl', store=store, rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        res, pipe, (y0, y1) = resample_frame_sharded(hdr, 110, cam, t, img, pxPerDeg=8, min_elevation=10)
    finally:
        dist.destroy_process_group()
    assert (y0, y1) == (0, h) and res['contains_pole']
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
        assert np.array_equal(res[k], ref[k], equal_nan=True), k
