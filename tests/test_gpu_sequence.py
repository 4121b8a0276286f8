"""
GPU tests of the sequence loop in its PRODUCTION mode — ``SequencePipeline.process(..., keep_on_device=True)``, what
bench.py and ``run_sequence`` use: nothing synchronises between frames, results are copied to the host only after
process() has returned.  Distinct per-frame images (pageable, pinned, device resident), every plan and batch size,
with pole frames (per-frame fall-back to the two-pass plan inside the single-pass plan) and frames without any valid
pixel in between (image-buffer races, empty frames, gather descriptors).
"""
from datetime import datetime

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def pole_frame(w, h):
    """A camera 400 km above the geographic north pole looking straight down (48 deg field of view)."""
    from auromat_amd.coordinates import transform as T
    t = datetime(2012, 1, 25, 9, 26, 55)
    zen = T.mat_j2000_to_geo(T.date2es(t)).T.dot([0.0, 0.0, 1.0])
    bore = -zen
    s = 48.0 / w
    hdr = {'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN', 'LONPOLE': 180.0, 'LATPOLE': 0.0,
           'CRVAL1': np.rad2deg(np.arctan2(bore[1], bore[0])) % 360, 'CRVAL2': np.rad2deg(np.arcsin(bore[2])),
           'CRPIX1': w / 2 + 0.5, 'CRPIX2': h / 2 + 0.5, 'CD1_1': -s, 'CD1_2': 0.0, 'CD2_1': 0.0, 'CD2_2': s,
           'IMAGEW': w, 'IMAGEH': h}
    return hdr, zen * (6356.75 + 400.0), t


def sky_frame(w, h):
    """The same camera looking away from the Earth: no ray hits the shell."""
    hdr, cam, t = pole_frame(w, h)
    hdr = dict(hdr, CRVAL1=(hdr['CRVAL1'] + 180.0) % 360, CRVAL2=-hdr['CRVAL2'])
    return hdr, cam, t


def build_sequence(w, h, n, every_pole=5, empty_at=()):
    from auromat_amd.synthetic import frame_image, sequence_frame
    frames = []
    for k in range(n):
        img = frame_image(w, h, seed=100 + k)
        if k in empty_at:
            hdr, cam, t = sky_frame(w, h)
        elif every_pole and k % every_pole == every_pole - 1:
            hdr, cam, t = pole_frame(w, h)
        else:
            hdr, cam, t, _ = sequence_frame(k, w, h)
        frames.append((hdr, cam, t, img))
    return frames


def host(res):
    import torch
    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in res.items()}


KEYS = ('mean', 'count', 'img', 'mask')


@pytest.mark.parametrize('how', ['pageable', 'pinned', 'resident'])
def test_keep_on_device_sequences_equal_frame_by_frame(how):
    import torch
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    w, h, n = 1060, 708, 23
    frames = build_sequence(w, h, n)
    single = FramePipeline(w, h)
    want = [single.run(hdr, 110, cam, t, img=img, pxPerDeg=6) for hdr, cam, t, img in frames]
    n_pole = sum(1 for r in want if r['contains_pole'])
    assert n_pole == 4
    if how == 'pinned':
        feed = [(hd, c, t, torch.from_numpy(im.view(np.int16)).pin_memory()) for hd, c, t, im in frames]
    elif how == 'resident':
        feed = [(hd, c, t, torch.from_numpy(im.view(np.int16)).cuda()) for hd, c, t, im in frames]
    else:
        feed = frames
    for plan, batch, bin_stream in (('single-pass', 3, True), ('single-pass', 1, True), ('single-pass', 2, True),
                                    ('two-pass', 1, True), ('two-pass', 1, False)):
        seq = SequencePipeline(w, h, pxPerDeg=6, plan=plan, batch=batch, bin_stream=bin_stream,
                               own_image_buffers=how != 'resident')
        for rep in range(2):                # the second call re-uses buffers that still have readers in flight
            got = seq.process(feed, keep_on_device=True)
            assert all(isinstance(r['mean'], torch.Tensor) and r['mean'].is_cuda for r in got)
            if plan == 'single-pass':
                assert seq.plans == ['single-pass'] * n      # pole frames included (pole plan of the fused kernel)
            got = [host(r) for r in got]    # only now
            for k in range(n):
                for key in KEYS:
                    a, b = got[k][key], want[k][key]
                    if key == 'img':
                        a = a.view(b.dtype)
                    assert np.array_equal(a.astype(b.dtype) if key == 'mask' else a, b, equal_nan=True), \
                        (how, plan, batch, bin_stream, rep, k, key)
                assert got[k]['contains_pole'] == want[k]['contains_pole']


def test_a_frame_without_valid_pixels_does_not_stop_the_sequence():
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    from auromat_amd.sequence import frame_coordinates, run_sequence
    w, h, n = 300, 200, 9
    frames = build_sequence(w, h, n, every_pole=4, empty_at=(2, 8))
    single = FramePipeline(w, h)
    want = []
    for k, (hdr, cam, t, img) in enumerate(frames):
        if k in (2, 8):
            with pytest.raises(ValueError):
                single.run(hdr, 110, cam, t, img=img, pxPerDeg=5)
            want.append(None)
        else:
            want.append(single.run(hdr, 110, cam, t, img=img, pxPerDeg=5))
    for plan, batch in (('single-pass', 3), ('single-pass', 1), ('two-pass', 1)):
        seq = SequencePipeline(w, h, pxPerDeg=5, plan=plan, batch=batch)
        got = seq.process(frames, keep_on_device=True)
        assert [r is None for r in got] == [k in (2, 8) for k in range(n)]
        assert seq.plans[2] == seq.plans[8] == 'empty'
        for k in range(n):
            if want[k] is not None:
                g = host(got[k])
                assert np.array_equal(g['mean'], want[k]['mean'], equal_nan=True), (plan, batch, k)
                assert np.array_equal(g['count'], want[k]['count']), (plan, batch, k)
    # the gather path: empty frames are reported, pole and date-line flags travel, coordinates can be recovered
    out, failed = run_sequence(frames, w, h, pxPerDeg=5, return_failed=True)
    assert failed == [2, 8] and [f['index'] for f in out] == [0, 1, 3, 4, 5, 6, 7]
    for f in out:
        ref = want[f['index']]
        assert f['contains_pole'] == ref['contains_pole'] and f['altitude'] == 110 and not f['magnetic']
        assert f['contains_discontinuity'] == ref['contains_discontinuity']
        lat_c, lon_c = frame_coordinates(f)
        assert np.allclose(lat_c, ref['lat_c'], atol=1e-9, rtol=0), f['index']
        dl = np.abs(lon_c - ref['lon_c'])
        assert np.all(np.minimum(dl, 360 - dl) * np.cos(np.deg2rad(ref['lat_c'])) < 1e-9), f['index']


def test_gathered_dateline_frame_keeps_its_true_coordinates():
    """iss029 moved so that its footprint straddles the date line: the grid is laid out in longitudes shifted by
    180 deg (resample.py:203-218); the descriptor says so and frame_coordinates() undoes it."""
    from datetime import timedelta
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.sequence import frame_coordinates, run_sequence
    from auromat_amd.synthetic import frame_header, frame_image
    w, h = 253, 171
    hdr, cam, t = frame_header(w, h, 'iss029')
    t = t - timedelta(minutes=80)           # same inertial geometry, the Earth 20 deg further west (142..168 E before)
    img = frame_image(w, h, seed=5)
    want = FramePipeline(w, h).run(hdr, 110, cam, t, img=img, pxPerDeg=5)
    assert want['contains_discontinuity']
    out = run_sequence([(hdr, cam, t, img)], w, h, pxPerDeg=5)
    assert out[0]['contains_discontinuity'] and not out[0]['contains_pole']
    lat_c, lon_c = frame_coordinates(out[0])
    assert np.allclose(lat_c, want['lat_c'], atol=1e-12, rtol=0) and np.allclose(lon_c, want['lon_c'], atol=1e-12, rtol=0)
    assert lon_c.min() < -170 and lon_c.max() > 170


def test_grids_only_pipeline_equals_the_full_one():
    """keep_coordinates=False: the single-pass plan writes no per-pixel coordinate arrays (pole frames included).  Same
    grids, bit for bit; the arrays appear when asked for."""
    import torch
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    w, h, n = 530, 354, 11
    frames = build_sequence(w, h, n, every_pole=4)
    full = SequencePipeline(w, h, pxPerDeg=6)
    want = [host(r) for r in full.process(frames)]
    lean = SequencePipeline(w, h, pxPerDeg=6, keep_coordinates=False)
    got = [host(r) for r in lean.process(frames)]
    assert lean.plans == full.plans == ['single-pass'] * n
    for a, b in zip(got, want):
        for key in KEYS:
            assert np.array_equal(a[key], b[key], equal_nan=True), key
    # no coordinate arrays were allocated ...
    assert all(q.fd.lat is None for q in lean.pipes)
    # ... and a single frame's arrays come on demand, equal to the full pipeline's
    one = FramePipeline(w, h, alloc_coords=False)
    hdr, cam, t, img = frames[0]
    res = one.run(hdr, 110, cam, t, img=img, pxPerDeg=6, fuse=True)
    assert one.last_plan == 'single-pass' and one.fd.lat is None
    ref = FramePipeline(w, h)
    ref.run(hdr, 110, cam, t, img=img, pxPerDeg=6, fuse=True)
    a, b = one.host_arrays(), ref.host_arrays()
    for k in b:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.array_equal(res['mean'], host(want[0])['mean'], equal_nan=True)


def test_c_abi_pack_writes_the_python_wire_format():
    """amt_seq_pack (what a C host feeds to RCCL) == auromat_amd.sequence.pack_results on the same device grids."""
    import ctypes as C
    import torch
    from auromat_amd._native import SeqFrame, ptr
    from auromat_amd.pipeline import SequencePipeline
    from auromat_amd.sequence import DESC_LEN, pack_results
    w, h, n = 300, 200, 5
    frames = build_sequence(w, h, n, every_pole=3, empty_at=(3,))
    seq = SequencePipeline(w, h, pxPerDeg=5)
    results = seq.process(frames)
    idx = [10, 11, 12, 13, 14]
    descs, payload = pack_results(results, idx, seq.ctx.device)
    max_frames = 7
    fr = (SeqFrame * n)()
    for f, r, i in zip(fr, results, idx):
        f.index = i
        if r is None:
            continue
        g = r['grid']
        f.ny, f.nx, f.nc = r['mean'].shape
        f.lat0, f.lon0, f.dlat, f.dlon = g.lat0, g.lon0, g.latStep, g.lonStep
        f.contains_pole, f.contains_discontinuity, f.magnetic = int(r['contains_pole']), int(r['contains_discontinuity']), 0
        f.altitude = r['altitude']
        f.mean, f.count = r['mean'].data_ptr(), r['count'].data_ptr()
    size = C.c_int64()
    assert seq.ctx._lib.amt_seq_payload_size(fr, n, C.byref(size)) == 0 and size.value == payload.numel()
    buf = torch.full((max_frames * DESC_LEN + size.value,), -1.0, dtype=torch.float64, device=seq.ctx.device)
    seq.ctx.call('amt_seq_pack', fr, n, max_frames, ptr(buf), buf.numel())
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert np.array_equal(got[:n * DESC_LEN].reshape(n, DESC_LEN), descs.cpu().numpy())
    assert not got[n * DESC_LEN:max_frames * DESC_LEN].any()
    assert np.array_equal(got[max_frames * DESC_LEN:], payload.cpu().numpy(), equal_nan=True)


def test_fallback_frames_inside_a_single_pass_sequence_do_not_race_with_the_finalise_stream(monkeypatch):
    """Frames that take the two-pass plan in the middle of single-pass batches, production mode (results stay on the device
    until process() has returned), 8 repetitions.  History: sequence 35 of `tools/fuzz_sequence.py 80 1` had two frames
    whose superset grid could not be laid out; their accumulators are torch temporaries of the main stream, and the next
    frames' outputs used to be carved from the same memory while the driver's finalise kernel — on its own stream — wrote
    them at once (fixed: amt_pipe_finalize_stream, outputs are allocated for that stream).  Those two frames take the
    single-pass plan now, so three frames are MADE to fall back here (their driver result is withheld).  The scenario is
    the same; whether the old allocation would still trip over it is a matter of timing (it was 4 of 5 fuzz runs then)."""
    import torch
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    from auromat_amd.synthetic import random_sequence
    w, h = 250, 168
    rng = np.random.RandomState(1)
    for s in range(36):
        frames = random_sequence(rng, w, h)
        rng.randint(2), rng.choice([4, 8, 10])              # what the fuzzer draws per sequence
    assert len(frames) == 24
    one = FramePipeline(w, h)
    want = [one.run(hd, 110, cam, t, img=img, pxPerDeg=4) for hd, cam, t, img in frames]
    fall_back = {7, 8, 16}
    calls = [0]
    ready = FramePipeline.fused_ready

    def withheld(self, pxPerDeg, magnetic):
        k = calls[0]                                        # frames are finished in order, one call each
        calls[0] += 1
        res = ready(self, pxPerDeg, magnetic)
        if k in fall_back and res is not None:
            res.status = 1                                  # "the driver cannot finalise this frame": the general path
            res.edge_pixels = 1 << 20                       # (as if for more on-edge pixels than records: no second launch)
            return None
        return res
    monkeypatch.setattr(FramePipeline, 'fused_ready', withheld)
    for rep in range(8):
        calls[0] = 0
        seq = SequencePipeline(w, h, pxPerDeg=4, plan='single-pass', batch=3)
        got = seq.process(frames, keep_on_device=True)
        assert [k for k, plan in enumerate(seq.plans) if plan == 'two-pass'] == sorted(fall_back), seq.plans
        got = [host(r) for r in got]
        for k, (a, b) in enumerate(zip(got, want)):
            for key in KEYS:
                x = a[key].view(b[key].dtype) if key == 'img' else a[key].astype(b[key].dtype)
                assert np.array_equal(x, b[key], equal_nan=True), (rep, k, key, seq.plans[k])


@pytest.mark.parametrize('arcsec', [None, 600])
def test_an_iterator_of_frames_is_consumed_as_the_sequence_advances(arcsec):
    """process() used to turn its iterable into a list before it decided on a loop — a convert run's read-ahead
    generator of decoded host images (36-72 MB each) then held the whole sequence in memory before the first launch.  A
    generator is pulled a few batches ahead of the frames that are finished, never to its end; results as for a list."""
    from auromat_amd.pipeline import SequencePipeline
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h, n = 250, 168, 30
    frames = []
    for k in range(n):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        frames.append((hdr, cam, t, frame_image(w, h, seed=seed)))
    kw = dict(arcsecPerPx=arcsec) if arcsec else dict(pxPerDeg=8)
    want = SequencePipeline(w, h, **kw).process(frames, keep_on_device=False)
    pulled, seen = [0], []

    def feed():
        for f in frames:
            pulled[0] += 1
            yield f

    def on_batch(k0, results):
        seen.append((k0 + len(results), pulled[0]))

    seq = SequencePipeline(w, h, **kw)
    got = seq.process(feed(), keep_on_device=False, on_batch=on_batch)
    assert len(got) == n and seen[-1][0] == n
    ahead = max(p - done for done, p in seen[:-3])
    assert ahead <= 3 * seq.batch + 1 and seen[0][1] < n // 2, seen
    pulled[0] = 0
    again = seq.process(feed(), keep_on_device=False)          # without a hook: the same lazy loop for an iterator
    for a, b, c in zip(got, want, again):
        for key in ('mean', 'count', 'img', 'mask'):
            assert np.array_equal(a[key], b[key], equal_nan=True) and np.array_equal(c[key], b[key], equal_nan=True), key


def test_box_hints_are_extrapolated_at_the_cadence_of_real_sequences():
    """One frame every 3 s (the ISS sequences of the reference's resources): a frame is prepared 20 s of orbit ahead of
    the latest finished one, too far for that frame's box but not for the extrapolation of the two latest boxes."""
    from auromat_amd.pipeline import FramePipeline, SequencePipeline
    from auromat_amd.synthetic import frame_image, sequence_frame
    w, h, n = 530, 354, 40
    frames = []
    for k in range(n):
        hdr, cam, t, seed = sequence_frame(3 * k, w, h)
        frames.append((hdr, cam, t, frame_image(w, h, seed=seed)))
    one = FramePipeline(w, h)
    want = [one.run(hd, 110, cam, t, img=img, pxPerDeg=6) for hd, cam, t, img in frames]
    for batch in (3, 1):
        seq = SequencePipeline(w, h, pxPerDeg=6, batch=batch)
        got = [host(r) for r in seq.process(frames)]
        assert seq.plans == ['single-pass'] * n, seq.plans
        assert seq.hinted >= n - 3 * 2 * batch - 2, seq.hinted          # all but the frames prepared before two were finished
        for a, b in zip(got, want):
            for key in KEYS:
                x = a[key].view(b[key].dtype) if key == 'img' else a[key].astype(b[key].dtype)
                assert np.array_equal(x, b[key], equal_nan=True), key


def test_two_launch_streams_give_the_same_grids():
    """launch_streams=2 (batches alternate between two streams; a buffer set stays on the stream of its batch parity)."""
    from auromat_amd.pipeline import SequencePipeline
    w, h, n = 530, 354, 20
    frames = build_sequence(w, h, n, every_pole=6, empty_at=(7,))
    want = [None if r is None else host(r) for r in SequencePipeline(w, h, pxPerDeg=6).process(frames)]
    for batch in (3, 2):
        seq = SequencePipeline(w, h, pxPerDeg=6, batch=batch, launch_streams=2)
        for rep in range(2):
            got = [None if r is None else host(r) for r in seq.process(frames)]
            for a, b in zip(got, want):
                assert (a is None) == (b is None)
                if a is not None:
                    for key in KEYS:
                        assert np.array_equal(a[key], b[key], equal_nan=True), (batch, rep, key)


def test_send_buffer_packed_while_the_sequence_runs():
    """sequence.Packer fed by SequencePipeline.process(on_batch=...) on the finalise stream: the buffer equals the one
    packed after the call (single-pass frames, pole frames, a frame of empty sky)."""
    import torch
    from auromat_amd.pipeline import SequencePipeline
    from auromat_amd.sequence import DESC_LEN, Packer, pack_results
    w, h = 300, 200
    frames = build_sequence(w, h, 13, every_pole=5, empty_at=(7,))
    seq = SequencePipeline(w, h, pxPerDeg=6)
    first = seq.process(frames)
    assert any(r is None for r in first) and any(r is not None and r['contains_pole'] for r in first)
    idx = list(range(100, 100 + len(frames)))
    descs, payload = pack_results(first, idx, seq.ctx.device)
    cap = (len(frames) + 2, payload.numel() + 100)
    for rep in range(3):
        packer = Packer(cap, seq.ctx.device, seq.finalize_stream())
        res = seq.process(frames, on_batch=packer.add)
        buf = packer.finish(res, idx)
        # no device-wide synchronisation: finish() orders the current stream behind the packer's own, and
        # the copy to the host below runs on the current stream
        got = buf.cpu().numpy()
        assert got[-2] == len(frames) and got[-1] == payload.numel()
        assert np.array_equal(got[:descs.numel()].reshape(descs.shape), descs.cpu().numpy())
        assert np.array_equal(got[cap[0] * DESC_LEN:cap[0] * DESC_LEN + payload.numel()], payload.cpu().numpy(), equal_nan=True)


def test_plain_c_host_runs_a_sequence(tmp_path):
    """examples/c_sequence_demo.c — C99, no Python / torch / HIP headers — runs a whole sequence through the native runner
    (amt_run_create / amt_run_process): the grids it writes are the Python host's, bit for bit, pole frame and frame of
    empty sky included."""
    import ctypes as C
    import os
    import subprocess
    from conftest import ROOT
    from auromat_amd._build import LIB_DIR
    from auromat_amd._native import RunFrame, RunResult
    from auromat_amd.mapping.astrometry import run_frame
    from auromat_amd.pipeline import SequencePipeline
    exe = str(tmp_path / 'c_sequence_demo')
    build = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-I' + os.path.join(ROOT, 'include'),
                            os.path.join(ROOT, 'examples', 'c_sequence_demo.c'), '-L' + LIB_DIR, '-lauromat_hip',
                            '-Wl,-rpath,' + LIB_DIR, '-Wl,-rpath,/opt/rocm/lib', '-lm', '-o', exe],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert build.returncode == 0 and build.stdout.strip() == '', build.stdout
    w, h, n = 300, 200, 11
    frames = build_sequence(w, h, n, every_pole=5, empty_at=(7,))
    arr = (RunFrame * n)()
    for k, (hdr, cam, t, img) in enumerate(frames):
        run_frame(hdr, cam, t, 0.0, None, out=arr[k])
    (tmp_path / 'frames.bin').write_bytes(bytes(C.string_at(C.byref(arr), C.sizeof(arr))))
    (tmp_path / 'images.bin').write_bytes(b''.join(img.tobytes() for _, _, _, img in frames))
    for resolution, kw, where in (('6', dict(pxPerDeg=6), []), ('-600', dict(arcsecPerPx=600), []),
                                  ('6', dict(pxPerDeg=6), ['host']), ('-600', dict(arcsecPerPx=600), ['host'])):
        # (a negative "px per degree" is arcsec per pixel: the reference's own call form, every frame at the px/deg of its box;
        # `host`: the images stay in page-locked host memory and the runner sends the rows of each that can be binned)
        run = subprocess.run([exe, str(tmp_path / 'frames.bin'), str(tmp_path / 'images.bin'), str(n), str(w), str(h), resolution,
                              str(tmp_path / 'out.bin')] + where, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             universal_newlines=True, timeout=120)
        assert run.returncode == 0 and run.stdout.startswith('ok %d frames' % n), run.stdout
        raw = (tmp_path / 'out.bin').read_bytes()
        rec = np.frombuffer(raw[:n * C.sizeof(RunResult)], dtype=np.dtype(RunResult))
        if where:
            assert 0 < rec['uploaded_bytes'].sum() < 0.8 * n * w * h * 6 and (rec['uploaded_bytes'][[7]] == 0).all(), rec['uploaded_bytes']
        else:
            assert not rec['uploaded_bytes'].any()
        grids = np.frombuffer(raw[n * C.sizeof(RunResult):], dtype=np.float64)
        seq = SequencePipeline(w, h, **kw)
        want = seq.process(frames, keep_on_device=False)
        n_pole = 2
        if 'arcsecPerPx' in kw:
            # a pole in view: no longitude resolution (status 4 from the C host, None here)
            assert seq.plans.count('pole-without-resolution') == n_pole and sum(r is None for r in want) == 1 + n_pole
        else:
            assert sum(r is None for r in want) == 1 and sum(1 for r in want if r is not None and r['contains_pole']) == n_pole
        for k, r in enumerate(want):
            if r is None:
                assert rec['status'][k] == (4 if seq.plans[k] == 'pole-without-resolution' else 2)
                continue
            assert rec['status'][k] == 0 and (rec['ny'][k], rec['nx'][k]) == r['count'].shape
            assert bool(rec['contains_pole'][k]) == bool(r['contains_pole'])
            if 'arcsecPerPx' in kw:
                assert (rec['lat_px_per_deg'][k], rec['lon_px_per_deg'][k]) == r['pxPerDeg']
            o, cells = int(rec['grid_offset'][k]), int(rec['ny'][k]) * int(rec['nx'][k])
            assert np.array_equal(grids[o:o + 4 * cells].reshape(r['mean'].shape), r['mean'], equal_nan=True), (resolution, k)
            assert np.array_equal(grids[o + 4 * cells:o + 5 * cells].reshape(r['count'].shape), r['count']), (resolution, k)


def test_native_runner_reports_every_frame_with_its_own_shell_and_parameters():
    """17 frames through the library's frame loop, every one on a shell of its own (and a pole frame now and then, whose cell
    coordinates depend on the shell): the result of frame k carries frame k's altitude, its amt_frame_params and its hint flag —
    also when the slot of the frame is taken by a later frame before the frame is finished (slots = 2 x frames per launch)."""
    import torch
    from auromat_amd._native import RunResult
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.pipeline import FramePipeline, NativeResults, SequencePipeline
    from auromat_amd.resample import grid_coordinates
    w, h, n = 530, 354, 17
    frames = build_sequence(w, h, n, every_pole=4)
    feed = [(hd, c, t, torch.from_numpy(im.view(np.int16)).cuda(), 95.0 + 2.5 * k) for k, (hd, c, t, im) in enumerate(frames)]
    seq = SequencePipeline(w, h, pxPerDeg=6, own_image_buffers=False)
    got = seq.process(feed, keep_on_device=True)
    assert isinstance(got, NativeResults) and seq.plans == ['single-pass'] * n
    table = np.frombuffer(got._rec, dtype=np.dtype(RunResult))
    single = FramePipeline(w, h)
    for k, (hd, c, t, im) in enumerate(frames):
        alt = 95.0 + 2.5 * k
        assert got[k]['altitude'] == alt, (k, got[k]['altitude'])
        p = frame_params(hd, alt, c, t, True)
        assert np.array_equal(np.array(table['params']['cam'][k]), np.array(p.cam)) and table['params']['a'][k] == p.a, k
        want = single.run(hd, alt, c, t, img=im, pxPerDeg=6)
        r = host(got[k])
        assert np.array_equal(r['mean'], want['mean'], equal_nan=True) and np.array_equal(r['count'], want['count']), k
        cg, cw = grid_coordinates(r), grid_coordinates(want)
        assert np.array_equal(cg['lat_c'], cw['lat_c']) and np.array_equal(cg['lon_c'], cw['lon_c']), k
