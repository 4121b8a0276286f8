"""
Strip-padded rows (include/auromat_hip.h amt_georef_out.row_layout, round 6): the layout in which a pipeline's own buffers are
written.  The bar: NOTHING a consumer sees changes — every array that comes out of a padded pipeline (compacted by
amt_unpad_rows) and every grid is bit for bit what the contiguous layout gives, for every kernel variant (fast / exact centres,
MLat / MLT, nine arrays, direction arrays, fused binning, pole plan), at awkward sizes and at BASELINE.json's full size; the
contiguous layout itself is what the other GPU test files pin to the oracle and the reference's fixtures.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COORDS = ('lat', 'lon', 'lat_c', 'lon_c', 'elev')
MAG = ('mlat', 'mlt', 'mlat_c', 'mlt_c')


def same_bits(a, b, what=''):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), '%s: %d elements differ' % (
        what, int((a.view(np.uint64) != b.view(np.uint64)).sum()) if a.dtype == np.float64 else -1)


@pytest.mark.parametrize('width', [1, 5, 62, 63, 64, 125, 126, 127, 189, 4240])
def test_unpad_rows_vs_numpy(width):
    import torch
    from auromat_amd._native import Context, ptr
    ctx = Context.current()
    pitch = int(ctx._lib.amt_padded_pitch(width))
    strips = (width + 1 + 62) // 63
    assert pitch == 64 * strips
    rows = 7
    rng = np.random.RandomState(width)
    src = rng.standard_normal((rows, pitch))
    src[rng.random_sample(src.shape) < 0.1] = np.nan
    dev = torch.from_numpy(src).to(ctx.device)
    for cols in (width, width + 1):
        x = np.arange(cols)
        want = src[:, 64 * (x // 63) + x % 63]
        out = torch.full((rows, cols), 7.0, dtype=torch.float64, device=ctx.device)
        ctx.call('amt_unpad_rows', ptr(dev), rows, cols, width, ptr(out))
        same_bits(out.cpu().numpy(), want, 'width %d cols %d' % (width, cols))
    # error behaviour of the entry point
    out = torch.zeros((rows, width + 2), dtype=torch.float64, device=ctx.device)
    assert ctx._lib.amt_unpad_rows(ctx.handle, ptr(dev), rows, width + 2, width, ptr(out)) != 0
    assert ctx._lib.amt_unpad_rows(ctx.handle, None, rows, width, width, ptr(out)) != 0


def test_padded_pitch_of_the_bench_frame():
    from auromat_amd._native import lib
    assert lib().amt_padded_pitch(4240) == 4352
    assert lib().amt_padded_pitch(0) == 0


@pytest.mark.parametrize('width,height,pointing,fast,mag', [
    (253, 171, 'iss030', True, True),
    (61, 35, 'iss029', True, False),       # narrower than one strip
    (130, 97, 'iss030', False, True),      # exact centres
    (126, 40, 'iss030', True, False),      # a width that is a multiple of 63: the last strip holds one corner column
    (189, 33, 'iss029', False, False),
])
def test_padded_equals_contiguous_small(width, height, pointing, fast, mag):
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image
    hdr, cam, t = frame_header(width, height, pointing)
    img = frame_image(width, height, seed=11)
    names = COORDS + (MAG if mag else ())
    for fuse in (False, True):
        a = FramePipeline(width, height, with_mag=mag)
        b = FramePipeline(width, height, with_mag=mag, padded=True)
        ra = a.run(hdr, 110, cam, t, img=img, fast=fast, min_elevation=10, pxPerDeg=7, fuse=fuse)
        rb = b.run(hdr, 110, cam, t, img=img, fast=fast, min_elevation=10, pxPerDeg=7, fuse=fuse)
        assert a.last_plan == b.last_plan
        ha, hb = a.host_arrays(), b.host_arrays()
        for k in names:
            same_bits(ha[k], hb[k], '%s (fuse=%s)' % (k, fuse))
        for k in ('mean', 'count', 'img', 'mask'):
            same_bits(ra[k], rb[k], 'grid %s (fuse=%s)' % (k, fuse))
        ba, bb = a.bounding_box(), b.bounding_box()
        assert (ba.latSouth, ba.lonWest, ba.latNorth, ba.lonEast) == (bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast)


def test_padded_equals_contiguous_full_size_all_variants():
    """BASELINE.json's frame: geodetic (fast, exact), MLat / MLT only, nine arrays — arrays and grids of the single-pass plan."""
    import torch
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import sequence_frame, frame_image
    W, H = 4240, 2832
    hdr, cam, t, seed = sequence_frame(3, W, H)
    img = frame_image(W, H, seed=seed)
    for fast, mag, geo in ((True, False, True), (False, False, True), (True, True, False), (True, True, True)):
        res, arrays, variants = [], [], []
        for padded in (False, True):
            q = FramePipeline(W, H, with_mag=mag, with_geo=geo, padded=padded)
            r = q.run(hdr, 110, cam, t, img=img, fast=fast, min_elevation=10, pxPerDeg=10, magnetic=mag, fuse=True)
            assert q.last_plan == 'single-pass'
            variants.append(q.ctx.last_variant())
            res.append(r)
            arrays.append(q.host_arrays(kept_only=True))
            del q
            torch.cuda.empty_cache()
        assert variants[0] == variants[1]
        assert set(arrays[0]) == set(arrays[1])
        for k in arrays[0]:
            same_bits(arrays[0][k], arrays[1][k], '%s fast=%s mag=%s geo=%s' % (k, fast, mag, geo))
        for k in ('mean', 'count', 'img', 'mask'):
            same_bits(res[0][k], res[1][k], 'grid %s fast=%s mag=%s geo=%s' % (k, fast, mag, geo))
        del res, arrays


def test_padded_direction_arrays_and_pole_plan():
    """k_georef_rows<DIRS_IN> and the pole plan (SECOND = 2) write padded rows like the other variants."""
    from auromat_amd.coordinates.wcs import pix2world
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.synthetic import frame_header, frame_image, pole_frame
    W, H = 380, 211
    hdr, cam, t = frame_header(W, H, 'iss030')
    img = frame_image(W, H, seed=2)
    got = []
    for padded in (False, True):
        q = FramePipeline(W, H, padded=padded)
        dirs = pix2world(hdr, W, H, corner=True, ascartesian=True, device=q.ctx.device)
        p = frame_params(hdr, 110, cam, t, True, magnetic=False)
        r = q.run(None, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=9, params=p, fuse=True, dirs=dirs,
                  containsPole=False)
        assert q.last_plan == 'single-pass'
        got.append((r, q.host_arrays()))
    for k in COORDS:
        same_bits(got[0][1][k], got[1][1][k], 'dirs-in ' + k)
    for k in ('mean', 'count', 'img', 'mask'):
        same_bits(got[0][0][k], got[1][0][k], 'dirs-in grid ' + k)
    # pole in view
    W, H = 640, 420
    hdr, cam, t = pole_frame(W, H)
    img = frame_image(W, H, seed=4)
    got = []
    for padded in (False, True):
        q = FramePipeline(W, H, padded=padded)
        r = q.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=4, fuse=True)
        got.append((r, q.host_arrays(), q.last_plan))
    assert got[0][2] == got[1][2]
    assert got[0][0]['contains_pole']
    for k in COORDS:
        same_bits(got[0][1][k], got[1][1][k], 'pole ' + k)
    for k in ('mean', 'count', 'img', 'mask'):
        same_bits(got[0][0][k], got[1][0][k], 'pole grid ' + k)


def test_sequence_padded_equals_contiguous_incl_fallback_frames():
    """SequencePipeline (library frame loop and the Python loop) with padded buffers: grids of every frame and the arrays left in
    the buffers equal those of contiguous buffers; frames that leave the single-pass plan (pole, empty sky) included."""
    import torch
    from auromat_amd.pipeline import SequencePipeline
    from auromat_amd.synthetic import sequence_frame, frame_image, pole_frame
    W, H = 530, 354
    frames = []
    dev_imgs = []
    for k in range(7):
        hdr, cam, t, seed = sequence_frame(k, W, H)
        frames.append([hdr, cam, t, frame_image(W, H, seed=seed), None])
    p_hdr, p_cam, p_t = pole_frame(W, H)
    frames.insert(3, [p_hdr, p_cam, p_t, frame_image(W, H, seed=77), None])
    sky = dict(frames[0][0], CRVAL2=frames[0][0]['CRVAL2'] + 70.0)        # looks off the limb: no valid pixel
    frames.insert(5, [sky, frames[0][1], frames[0][2], frame_image(W, H, seed=78), None])
    for resident in (True, False):
        out = []
        for padded in (False, True):
            seq = SequencePipeline(W, H, altitude=110, fast=True, min_elevation=10, pxPerDeg=8, padded=padded)
            assert seq.padded == padded
            fr = frames
            if resident:
                fr = [f[:3] + [torch.from_numpy(f[3].view(np.int16)).to(seq.ctx.device), None] for f in frames]
            res = seq.process(fr, keep_on_device=False)
            plans = list(seq.plans)
            last = seq.pipes[(len(fr) - 1) % len(seq.pipes)]
            arrays = {k: last.fd.host(k) for k in COORDS}
            out.append((list(res), plans, arrays))
        assert out[0][1] == out[1][1], (out[0][1], out[1][1])
        assert 'single-pass' in out[0][1]
        for k, (ra, rb) in enumerate(zip(out[0][0], out[1][0])):
            assert (ra is None) == (rb is None), k
            if ra is None:
                continue
            for name in ('mean', 'count', 'img', 'mask'):
                same_bits(ra[name], rb[name], 'frame %d %s (resident=%s)' % (k, name, resident))
        for name in COORDS:
            same_bits(out[0][2][name], out[1][2][name], 'arrays left in the last buffer: ' + name)


def test_c_abi_row_layout_errors():
    from auromat_amd._native import Context, GeorefOut, ptr
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.synthetic import frame_header
    import torch
    ctx = Context.current()
    W, H = 70, 20
    hdr, cam, t = frame_header(W, H, 'iss030')
    p = frame_params(hdr, 110, cam, t, True, magnetic=False)
    out = GeorefOut()
    lat = torch.empty((H + 1) * int(ctx._lib.amt_padded_pitch(W)), dtype=torch.float64, device=ctx.device)
    out.lat = lat.data_ptr()
    out.bbox_min_elevation = -np.inf
    out.row_layout = 7
    assert ctx._lib.amt_georef_frame(ctx.handle, C.byref(p), C.byref(out)) != 0
    assert b'row_layout' in ctx._lib.amt_last_error(ctx.handle)
    out.row_layout = 1
    assert ctx._lib.amt_georef_frame(ctx.handle, C.byref(p), C.byref(out)) == 0
    torch.cuda.synchronize()
