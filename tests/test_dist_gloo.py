"""
Multi-process (world_size 2, gloo, CPU tensors) test of the frame sharding and the padded gather of
per-frame grids used for sequences (auromat_amd/sequence.py).  The GPU run uses the same code with
backend "nccl" (= RCCL) and device tensors.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_result(k):
    """Deterministic stand-in for a resample_frame(keep_on_device=True) result of frame k (shape varies with k)."""
    from auromat_amd.resample import _Grid
    grid = _Grid((4, 5), 40.0 + 0.3 * k, 43.0 + 0.37 * k, -100.0 + k, -96.5 + 1.2 * k)
    rs = np.random.RandomState(k)
    mean = rs.uniform(0, 65535, (grid.ny, grid.nx, 4))
    mean[rs.rand(grid.ny, grid.nx) < 0.2] = np.nan
    count = rs.randint(0, 50, (grid.ny, grid.nx)).astype(np.float64)
    return dict(mean=torch.from_numpy(mean), count=torch.from_numpy(count), grid=grid, contains_pole=k % 4 == 3,
                contains_discontinuity=k % 4 == 1, altitude=100.0 + k, magnetic=k % 2 == 0)


def _worker(rank, world, port, n_frames, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from auromat_amd.sequence import gather_results, shard
    mine = shard(n_frames, rank, world)
    # frame 1 (when there is one) has no valid pixel: it travels as an empty descriptor
    results = [None if k == 1 and n_frames > 2 else _fake_result(k) for k in mine]
    from auromat_amd.sequence import gather_device
    g = gather_device(results, mine, torch.device('cpu'))
    got = g.unpack() if g is not None else None
    if rank == 0:
        ok = [k for k in range(n_frames) if not (k == 1 and n_frames > 2)]
        assert [f['index'] for f in got] == ok and g.failed == ([1] if n_frames > 2 else [])
        for f in got:
            ref = _fake_result(f['index'])
            np.testing.assert_array_equal(f['mean'], ref['mean'].numpy())
            np.testing.assert_array_equal(f['count'], ref['count'].numpy())
            assert f['lat0'] == ref['grid'].latCenters[0] and f['lon0'] == ref['grid'].lonCenters[0]
            assert f['dlat'] == ref['grid'].latStep and f['dlon'] == ref['grid'].lonStep
            for key in ('contains_pole', 'contains_discontinuity', 'altitude', 'magnetic'):
                assert f[key] == ref[key], key
            if f['contains_discontinuity']:
                # the grid was laid out in longitudes shifted by 180 deg: frame_coordinates() undoes it
                from auromat_amd.sequence import frame_coordinates
                from auromat_amd.mapping.mapping import wrap_at_180
                lat_c, lon_c = frame_coordinates(f)
                assert np.allclose(lat_c[:, 0], ref['grid'].latCenters, atol=1e-12, rtol=0)
                assert np.allclose(lon_c[0], wrap_at_180(np.asarray(ref['grid'].lonCenters) - 180))
        open(os.path.join(out_dir, 'ok'), 'w').write('ok')
    else:
        assert got is None
    # the same with ONE collective: a capacity agreed beforehand (here from these very results, 25 % margin) ...
    from auromat_amd.sequence import agree_capacity
    cap = agree_capacity(results, mine, torch.device('cpu'))
    assert cap[0] == max(len(shard(n_frames, r, world)) for r in range(world))
    g2 = gather_device(results, mine, torch.device('cpu'), capacity=cap)
    if rank == 0:
        again = g2.unpack()
        assert g2.n_frames == n_frames and [f['index'] for f in again] == [f['index'] for f in got] and g2.failed == g.failed
        for a, b in zip(again, got):
            np.testing.assert_array_equal(a['mean'], b['mean'])
            np.testing.assert_array_equal(a['count'], b['count'])
    # ... and a capacity that is too small: nobody hangs, the destination is told
    g3 = gather_device(results, mine, torch.device('cpu'), capacity=(cap[0], 10))
    if rank == 0 and n_frames > 1:
        with pytest.raises(ValueError, match='did not fit'):
            g3.unpack()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_frames', [5, 2, 1])
def test_gather_of_frame_grids_world2(tmp_path, n_frames):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_frames, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / 'ok').exists()


def test_shard_is_a_contiguous_partition():
    from auromat_amd.sequence import shard
    for n in (0, 1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            parts = [shard(n, r, world) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    from auromat_amd.sequence import pack_results, unpack_results
    results = [_fake_result(k) for k in (3, 9)]
    descs, payload = pack_results(results, [3, 9], torch.device('cpu'))
    out = unpack_results(descs, payload)
    assert [f['index'] for f in out] == [3, 9]
    for f, r in zip(out, results):
        np.testing.assert_array_equal(f['mean'], r['mean'].numpy())


def _shard_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from auromat_amd.sequence import RowShard
    s = RowShard(pole=False)
    inf = float('inf')
    # rank 0: a band over Europe; rank 1: a band of sky (nothing valid: the kernel's neutral elements)
    mine = [[40.0, 55.0, -5.0, 20.0, 0.5, -0.25, 1234.0, 0.0], [inf, -inf, inf, -inf, inf, -inf, 0.0, 0.0]][rank]
    red = s.box(np.array(mine))
    assert red.tolist() == [40.0, 55.0, -5.0, 20.0, 0.5, -0.25, 1234.0, 0.0], red
    acc = torch.arange(12, dtype=torch.int64).reshape(3, 4) * (rank + 1) - (7 if rank else 0)
    s.acc(acc)
    want = torch.arange(12, dtype=torch.int64).reshape(3, 4) * 3 - 7
    assert torch.equal(acc, want)
    with open(os.path.join(out_dir, 'shard_ok_%d' % rank), 'w') as fp:
        fp.write('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_row_shard_exchange_steps(tmp_path):
    """The two collectives of a frame whose rows are spread over the ranks (sequence.RowShard): box and accumulators."""
    world = 2
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(str(tmp_path / ('shard_ok_%d' % r))) for r in range(world))


def _packer_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from auromat_amd.sequence import Packer, agree_capacity, gather_device, shard
    n_frames = 7
    mine = shard(n_frames, rank, world)
    results = [None if k == 1 else _fake_result(k) for k in mine]
    for r in results:
        if r is not None and r['grid'].ny % 2 == 0:
            # like a single-pass frame: mean and count as one slice
            r['packed'] = torch.cat([r['mean'].reshape(-1), r['count'].reshape(-1)])
    cap = agree_capacity(results, mine, torch.device('cpu'))
    packer = Packer(cap, torch.device('cpu'))
    # the frames arrive launch by launch, while "later launches run"
    k = 0
    for size in (1, 3, 3):
        if k < len(results):
            packer.add(k, results[k:k + size])
            k += len(results[k:k + size])
    g = gather_device(results, mine, torch.device('cpu'), packer=packer)
    plain = gather_device(results, mine, torch.device('cpu'), capacity=cap)
    if rank == 0:
        got, want = g.unpack(), plain.unpack()
        assert [f['index'] for f in got] == [f['index'] for f in want] == [0, 2, 3, 4, 5, 6] and g.failed == [1]
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a['mean'], b['mean'])
            np.testing.assert_array_equal(a['count'], b['count'])
            for key in ('lat0', 'lon0', 'dlat', 'dlon', 'contains_pole', 'contains_discontinuity', 'altitude', 'magnetic'):
                assert a[key] == b[key], key
    # a rank whose grids do not fit says so in its buffer; nobody is left waiting
    small = Packer((cap[0], 10), torch.device('cpu'))
    small.add(0, results)
    g = gather_device(results, mine, torch.device('cpu'), packer=small)
    if rank == 0:
        with pytest.raises(ValueError, match='did not fit'):
            g.unpack()
    with open(os.path.join(out_dir, 'packer_ok_%d' % rank), 'w') as fp:
        fp.write('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_gather_from_a_buffer_packed_while_the_frames_were_computed(tmp_path):
    """sequence.Packer: the send buffer filled launch by launch == the buffer packed afterwards; overflow is reported."""
    world = 2
    mp.spawn(_packer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(str(tmp_path / ('packer_ok_%d' % r))) for r in range(world))


# ---- BASELINE configs[4] as far as it goes without eight GPUs: 256 frames over 8 ranks ------------------------------------------

def _config5_result(k):
    """What rank r's pipeline would hand over for frame k of the synthetic sequence: the frame's REAL
    0.1 deg grid — its bounding box from the oracle on the frame's own header at 1/40 of the resolution (the box of the
    masked footprint does not depend on the sampling beyond a pixel) — filled with seeded numbers."""
    from oracle import ref_numpy as O
    from auromat_amd.coordinates import transform as T
    from auromat_amd.resample import _Grid
    from auromat_amd.synthetic import sequence_frame
    hdr, cam, t, _ = sequence_frame(k, 106, 71)
    g = O.georef_frame(hdr, 110.0, cam, O.mat_j2000_to_geo(T.date2es(t)), None, fast=True)
    corner_mask, _ = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), 10)
    (lat_s, lon_w, lat_n, lon_e), disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
    assert not disc
    grid = _Grid((10, 10), lat_s, lat_n, lon_w, lon_e)
    rs = np.random.RandomState(k)
    mean = rs.uniform(0, 65535, (grid.ny, grid.nx, 4))
    mean[rs.rand(grid.ny, grid.nx) < 0.1] = np.nan
    count = rs.randint(0, 400, (grid.ny, grid.nx)).astype(np.float64)
    return dict(mean=torch.from_numpy(mean), count=torch.from_numpy(count), grid=grid, contains_pole=False,
                contains_discontinuity=False, altitude=110.0, magnetic=False)


def _config5_worker(rank, world, port, n_frames, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from auromat_amd.sequence import DESC_LEN, agree_capacity, gather_device, shard
    mine = shard(n_frames, rank, world)
    # config 5: 32 CONSECUTIVE frames per rank (a rank's box hints come from its own neighbouring frames)
    assert mine == list(range(32 * rank, 32 * rank + 32))
    results = [_config5_result(k) for k in mine]
    # the warm-up's agreement: every rank ends up with the same (frames, payload) capacity, large enough for the longest
    cap = agree_capacity(results, mine, torch.device('cpu'))
    own = sum(r['mean'].numel() + r['count'].numel() for r in results)
    caps = [None] * world
    dist.all_gather_object(caps, (cap, own))
    assert all(c[0] == cap for c in caps) and cap[0] == 32 and cap[1] >= max(c[1] for c in caps)
    assert cap[1] <= 1.25 * max(c[1] for c in caps) + 2
    # the timed gather: ONE collective of fixed size
    g = gather_device(results, mine, torch.device('cpu'), capacity=cap)
    if rank == 0:
        got = g.unpack()
        assert g.n_frames == n_frames and g.failed == []
        assert [f['index'] for f in got] == list(range(n_frames))
        shapes = set()
        for f in got[::17] + got[-1:]:
            ref = _config5_result(f['index'])
            np.testing.assert_array_equal(f['mean'], ref['mean'].numpy())
            np.testing.assert_array_equal(f['count'], ref['count'].numpy())
            assert (f['lat0'], f['lon0'], f['dlat'], f['dlon']) == (ref['grid'].lat0, ref['grid'].lon0, ref['grid'].latStep, ref['grid'].lonStep)
            shapes.add(f['mean'].shape)
        assert len(shapes) > 1                                   # the footprint moves: the grids differ frame by frame
        per_rank_bytes = (cap[0] * DESC_LEN + cap[1] + 2) * 8
        with open(os.path.join(out_dir, 'ok'), 'w') as fp:
            fp.write('%d' % (per_rank_bytes * world))
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


def test_config5_sharding_and_gather_world8(tmp_path):
    """256 frames of the synthetic sequence over 8 ranks (gloo; one process per rank as on the 8-GPU node): contiguous blocks
    of 32, the capacity agreement of the warm-up, ONE fixed-size gather of all 256 grids on rank 0 (about 2 MB per rank
    and 17 MB in all at 0.1 deg)."""
    world = 8
    mp.spawn(_config5_worker, args=(world, _free_port(), 256, str(tmp_path)), nprocs=world, join=True)
    total = int((tmp_path / 'ok').read_text())
    assert 8e6 < total < 1e8, total


# ---- a rank that fails must end the whole job --------------------------------------------------------------

@pytest.mark.parametrize('mode,culprit', [('ok', None), ('raise', 2), ('overflow', 1)])
def test_a_failing_rank_ends_every_rank(mode, culprit):
    """World 4 over gloo, one process per rank, started and watched one by one (as the launcher does): a rank whose pipeline
    raises NativeError in the middle of its frames ('raise'), or whose grids do not fit the capacity the ranks agreed on in the
    warm-up ('overflow'), still takes part in the gather and says so; every rank then raises SequenceError naming it and exits
    with a non-zero code within the timeout — nobody returns normally, nobody waits in a collective (before round 6 only
    rank 0 noticed an overflow, and a raising rank left the others in the gather until the backend's timeout)."""
    import subprocess
    import time
    world, port = 4, _free_port()
    worker = os.path.join(ROOT, 'tests', '_failing_rank_worker.py')
    env = dict(os.environ, OMP_NUM_THREADS='1')
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, worker, mode, str(r), str(world), str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, universal_newlines=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=max(5.0, 150.0 - (time.time() - t0))))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.returncode for p in procs]
    if culprit is None:
        assert codes == [0] * world, (codes, [o[1][-500:] for o in outs])
        return
    assert codes == [3] * world, (codes, [o[1][-500:] for o in outs])
    for r, (_, err) in enumerate(outs):
        assert 'SequenceError' in err and 'rank(s) [%d]' % culprit in err, (r, err[-500:])
    if mode == 'raise':
        assert all('NativeError' in err and 'injected' in err for _, err in outs)
    else:
        assert all('did not fit' in err for _, err in outs)


def test_host_thread_share_follows_the_ranks_on_the_node():
    """_native.host_threads (pure Python) and amt_host_threads (the library's pools) apply the same rule: the cores this process
    may run on divided by LOCAL_WORLD_SIZE, at least one, never more than asked for."""
    import ctypes as C
    import subprocess
    code = ("import ctypes as C, json, os, sys; sys.path.insert(0, %r); from auromat_amd import _native as N; "
            "c, r = C.c_int(), C.c_int(); n = N.lib().amt_host_threads(16, C.byref(c), C.byref(r)); "
            "print(json.dumps([n, c.value, r.value, N.host_threads(16), N.host_threads_report()]))" % ROOT)
    import json
    cores = len(os.sched_getaffinity(0))
    for ranks in (None, 2, 8, 1000):
        env = {k: v for k, v in os.environ.items() if k not in ('LOCAL_WORLD_SIZE', 'AMT_LOCAL_RANKS')}
        if ranks:
            env['LOCAL_WORLD_SIZE'] = str(ranks)
        out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, env=env, universal_newlines=True, check=True).stdout
        n, c, r, py, rep = json.loads(out.strip().splitlines()[-1])
        want = max(1, min(16, cores // (ranks or 1)))
        assert (n, c, r, py) == (want, cores, ranks or 1, want), (ranks, n, c, r, py)
        assert rep['copy_threads'] <= max(1, (rep['share'] + 1) // 2) and rep['triangulator_threads'] == want
