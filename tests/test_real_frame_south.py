"""
The rest of the reference's own test data at full size:

* test/resources/ISS029-E-8492.jpg + .wcs — the frame of `testSpacecraftMappingSouth` (test/mapping_test.py:36-42), southern
  hemisphere; on the (MLat, SM longitude) grid its box straddles +-180 deg of SM longitude, the natural date-line case of
  resampleMLatMLT (resample.py:203-218);
* the header sequences test/resources/seq2/ (4 frames) and seq3/ (3 frames).

Copies of the data files: tests/golden/resources/{south,seq2,seq3}/.  Expected values from the REAL reference (oracle/make_golden.py:
real_frame_south, real_sequences_more): fast / exact centres -> maskedByElevation(10) -> _resample(pxPerDeg=10, 'mean').

CPU: the oracle equals the reference cell for cell.  GPU: the reference's call sequence through the classes, both plans of
the frame pipeline, the box-first plan for `resample(m, arcsecPerPx=100)` (the call of the reference's test), the sequence
pipeline in its library loop and its Python loop.
"""
import glob
import os

import numpy as np
import numpy.ma as ma
import pytest

from conftest import GOLDEN, load_golden, oracle_frame

JPG = os.path.join(GOLDEN, 'resources', 'south', 'ISS029-E-8492.jpg')
WCS = os.path.join(GOLDEN, 'resources', 'south', 'ISS029-E-8492.wcs')


def check(res_img, res_mask, mean, z):
    want = z['out_data']
    assert res_mask.shape == want.shape[:2], (res_mask.shape, want.shape)
    assert np.array_equal(res_mask, np.isnan(want[..., 0]))
    ok = ~res_mask
    assert ok.sum() > 2000
    if mean is not None:
        assert np.array_equal(mean[..., :3][ok], want[..., :3][ok])                 # exact integer sums / counts
        assert np.max(np.abs(mean[..., 3][ok] - want[..., 3][ok])) < 1e-9           # elevation, fixed point
    assert np.array_equal(res_img[ok], z['out_img'][ok])


def south_inputs():
    from auromat_amd.fits import getSpacecraftPosition, readHeader
    from auromat_amd.util.image import loadImage
    hdr = readHeader(WCS)
    img = loadImage(JPG)
    cam, t = getSpacecraftPosition(hdr)
    return hdr, img, cam, t


def test_oracle_equals_the_reference_on_the_south_frame():
    """Geographic grid (fast centres) and the (MLat, SM longitude) grid across +-180 deg, from ONE run of the oracle's frame."""
    from oracle import ref_numpy as O
    z = load_golden('real_frame_iss029.npz')
    hdr, img, cam, t = south_inputs()
    assert img.shape == (2832, 4256, 3) and img.dtype == np.uint8
    assert np.array_equal(np.asarray(cam, dtype=np.float64), z['cam'])
    g = oracle_frame(hdr, 110.0, z['cam'], z['m_geo'], z['m_sm'], fast=True)
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), 10)
    assert int((~center_mask).sum()) == int(z['n_valid'])
    data = np.dstack((img.astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
    assert np.array_equal(bbox, z['bbox']) and disc == bool(z['contains_discontinuity'])
    assert z['bbox'][2] < 0                                                          # southern hemisphere
    res = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), 110.0,
                          data, None, bbox, (10, 10), disc, False)
    for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c'), ('data', 'out_data')):
        assert np.array_equal(res[a], z[b], equal_nan=True), a
    # resampleMLatMLT's _resample call: (MLat, mltToSmLon(MLT)); the box runs from +154 over +-180 to -178 deg
    zs = load_golden('real_frame_iss029_sm.npz')
    sm_lon, sm_lon_c = O.mlt_to_sm_lon(g['mlt']), O.mlt_to_sm_lon(g['mlt_c'])
    bbox, disc = O.bbox_of_corners(g['mlat'], sm_lon, corner_mask)
    assert np.array_equal(bbox, zs['bbox']) and disc and bool(zs['contains_discontinuity'])
    assert bbox[1] > 150 and bbox[3] < -170
    outline = np.transpose([g['mlat'][~corner_mask], sm_lon[~corner_mask]])      # (its extremes are what _resample takes from it)
    res = O.resample_mean(np.where(center_mask, np.nan, g['mlat_c']), np.where(center_mask, np.nan, sm_lon_c), 110.0,
                          data, outline, bbox, (10, 10), True, False)
    for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c'), ('data', 'out_data')):
        assert np.array_equal(res[a], zs[b], equal_nan=True), a


@pytest.mark.gpu
def test_the_references_south_test_through_the_classes():
    """test/mapping_test.py:36-42 testSpacecraftMappingSouth: getMapping -> checkGuarantees -> maskedByElevation(10) ->
    resample; here with pxPerDeg=10 against the reference's grid (single-pass plan), then the reference's own call,
    arcsecPerPx=100, box-first plan against the array route."""
    import auromat_amd.resample as R
    from auromat_amd.mapping.spacecraft import getMapping
    for fast, name in ((True, 'real_frame_iss029.npz'), (False, 'real_frame_iss029_exact.npz')):
        z = load_golden(name)
        mm = getMapping(JPG, WCS, altitude=110, fastCenterCalculation=fast).maskedByElevation(10)
        assert mm.identifier == 'ISS029-E-8492' and mm._frame is None
        r = R.resample(mm, pxPerDeg=10)
        assert R.last_plan == 'single-pass' and mm._frame is None
        r.checkGuarantees()
        assert np.array_equal(r.lats.data, z['out_lat']) and np.array_equal(r.lonsCenter.data, z['out_lon_c'])
        check(r.img.data, ma.getmaskarray(r.img)[..., 0], None, z)
        assert r.img.dtype == np.uint8
    # the arrays, on first use, carry the mask; the box equals the reference's
    z = load_golden('real_frame_iss029.npz')
    assert int((~ma.getmaskarray(mm.latsCenter)).sum()) > 0
    m = getMapping(JPG, WCS, fastCenterCalculation=True)
    m.checkGuarantees()
    m2 = m.maskedByElevation(10)
    m2.checkGuarantees()
    assert int((~ma.getmaskarray(m2.latsCenter)).sum()) == int(z['n_valid'])
    assert np.any(~(ma.getmaskarray(m.latsCenter) == ma.getmaskarray(m2.latsCenter)))
    bb = m2.boundingBox
    np.testing.assert_allclose([bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast], z['bbox'], rtol=0, atol=1e-9)
    # the reference's own call form
    want = R.resample(m2, arcsecPerPx=100, method='mean')           # arrays exist: the array route
    assert R.last_plan != 'single-pass'
    want.checkGuarantees()
    got = R.resample(getMapping(JPG, WCS, fastCenterCalculation=True).maskedByElevation(10), arcsecPerPx=100, method='mean')
    assert R.last_plan == 'single-pass'
    got.checkGuarantees()
    for name in ('lats', 'lons', 'latsCenter', 'lonsCenter', 'img', 'elevation'):
        a, b = getattr(got, name), getattr(want, name)
        assert np.array_equal(ma.getmaskarray(a), ma.getmaskarray(b)), name
        if name == 'elevation':
            assert np.max(np.abs(a.compressed() - b.compressed())) < 1e-9
        else:
            assert np.array_equal(a.filled(0), b.filled(0)), name


@pytest.mark.gpu
def test_both_plans_on_the_south_frame():
    from auromat_amd.pipeline import FramePipeline
    hdr, img, cam, t = south_inputs()
    for fast, name in ((True, 'real_frame_iss029.npz'), (False, 'real_frame_iss029_exact.npz')):
        z = load_golden(name)
        for fuse in (True, False):
            pipe = FramePipeline(4256, 2832, img_dtype=np.uint8)
            res = pipe.run(hdr, 110, cam, t, img=img, fast=fast, min_elevation=10, pxPerDeg=10, fuse=fuse)
            assert pipe.last_plan == ('single-pass' if fuse else 'two-pass')
            assert np.array_equal(res['lat'], z['out_lat']) and np.array_equal(res['lon_c'], z['out_lon_c'])
            check(res['img'], res['mask'], res['mean'], z)
    # box-first plan in the frame pipeline: the exact box first, the resolution from it, then the single-pass launch
    pipe = FramePipeline(4256, 2832, img_dtype=np.uint8, alloc_coords=False)
    a = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, arcsecPerPx=100, fuse=True)
    assert pipe.last_plan == 'single-pass' and a['pxPerDeg'][0] == 36.0
    b = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=a['pxPerDeg'], fuse=False)
    assert pipe.last_plan == 'two-pass'
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon_c'):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.gpu
def test_south_frame_on_the_mlat_mlt_grid_across_the_date_line():
    """resampleMLatMLT on the frame whose SM-longitude box wraps at +-180 deg (reference resample.py:203-218: the grid is laid
    out for longitudes shifted by 180 deg and shifted back): the class route, both plans of the frame pipeline, the MLat /
    MLT-only mode of the fused kernel — the reference's grid cell for cell, its coordinates bit for bit."""
    import auromat_amd.resample as R
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.pipeline import FramePipeline
    z = load_golden('real_frame_iss029_sm.npz')
    assert bool(z['contains_discontinuity'])
    hdr, img, cam, t = south_inputs()
    for fuse, geo in ((True, True), (False, True), (True, False)):
        pipe = FramePipeline(4256, 2832, img_dtype=np.uint8, with_mag=True, with_geo=geo)
        res = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, fuse=fuse, magnetic=True)
        assert pipe.last_plan == ('single-pass' if fuse else 'two-pass')
        assert not fuse or pipe.ctx.last_variant()[0] == (1 if geo else 4)
        assert res['contains_discontinuity']
        for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c')):
            assert np.array_equal(res[a], z[b]), (a, fuse, geo)
        check(res['img'], res['mask'], res['mean'], z)
    mm = getMapping(JPG, WCS, fastCenterCalculation=True).maskedByElevation(10)
    r = R.resampleMLatMLT(mm, pxPerDeg=10)
    assert R.last_plan == 'single-pass'
    check(r.img.data, ma.getmaskarray(r.img)[..., 0], None, z)
    # the reference's call form on this grid: box-first plan against the array route
    got = R.resampleMLatMLT(getMapping(JPG, WCS, fastCenterCalculation=True).maskedByElevation(10), arcsecPerPx=100)
    assert R.last_plan == 'single-pass'
    ref = getMapping(JPG, WCS, fastCenterCalculation=True).maskedByElevation(10)
    ref.latsCenter
    want = R.resampleMLatMLT(ref, arcsecPerPx=100)
    assert R.last_plan != 'single-pass'
    assert got.img.shape == want.img.shape and got.img.shape[0] > 200
    for name in ('lats', 'lons', 'latsCenter', 'lonsCenter', 'img', 'elevation'):
        a, b = getattr(got, name), getattr(want, name)
        assert np.array_equal(ma.getmaskarray(a), ma.getmaskarray(b)), name
        if name == 'elevation':
            assert np.max(np.abs(a.compressed() - b.compressed())) < 1e-9
        else:
            assert np.array_equal(a.filled(0), b.filled(0)), name


# ---- the reference's other two header sequences ---------------------------------------------------------------------------

def sequence_frames(seq):
    from auromat_amd.fits import getSpacecraftPosition, readHeader
    from auromat_amd.synthetic import frame_image
    frames = []
    for k, path in enumerate(sorted(glob.glob(os.path.join(GOLDEN, 'resources', seq, '*.wcs')))):
        hdr = readHeader(path)
        cam, t = getSpacecraftPosition(hdr)
        frames.append((hdr, cam, t, frame_image(4256, 2832, seed=k)))
    return frames


@pytest.mark.parametrize('seq,first,n', [('seq2', 229356, 4), ('seq3', 102170, 3)])
def test_sequence_headers_are_the_references(seq, first, n):
    z = load_golden('real_sequence_%s.npz' % seq)
    frames = sequence_frames(seq)
    assert len(frames) == n == len(z['names'])
    assert [str(v) for v in z['names']] == ['ISS030-E-%d.wcs' % (first + i) for i in range(n)]
    assert all(f[0]['IMAGEW'] == 4256 and f[0]['IMAGEH'] == 2832 for f in frames)
    times = [f[2] for f in frames]
    assert all(0.5 < (b - a).total_seconds() < 3.5 for a, b in zip(times, times[1:]))


def test_oracle_equals_the_reference_on_a_frame_of_each_sequence():
    from oracle import ref_numpy as O
    from auromat_amd.coordinates import transform as T
    for seq, k in (('seq2', 3), ('seq3', 1)):
        z = load_golden('real_sequence_%s.npz' % seq)
        hdr, cam, t, img = sequence_frames(seq)[k]
        et = T.date2es(t)
        g = O.georef_frame(hdr, 110.0, cam, O.mat_j2000_to_geo(et), None, fast=True)
        corner_mask, center_mask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), 10)
        bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
        assert np.array_equal(bbox, z['bbox_%d' % k]) and disc == bool(z['disc_%d' % k])
        data = np.dstack((img.astype(np.float64), g['elev']))
        data[center_mask] = np.nan
        outline = np.transpose([g['lat'][~corner_mask], g['lon'][~corner_mask]])
        res = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), 110.0,
                              data, outline, bbox, (10, 10), disc, False)
        assert np.array_equal(res['data'], z['out_data_%d' % k], equal_nan=True), seq
        assert np.array_equal(res['lat'], z['out_lat_%d' % k]) and np.array_equal(res['lon'], z['out_lon_%d' % k]), seq


@pytest.mark.gpu
@pytest.mark.parametrize('seq', ['seq2', 'seq3'])
def test_sequence_pipeline_on_the_references_other_sequences(seq):
    """SequencePipeline on seq2 / seq3 at full size: host images (the Python loop with uploads) and device-resident images
    (the frame loop in the library), with and without the per-pixel coordinate arrays — every frame's grid equals the
    reference's, cell for cell; then `arcsecPerPx=100` (box-first plan) against the classes' array route."""
    import torch
    import auromat_amd.resample as R
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.pipeline import SequencePipeline
    from auromat_amd.resample import grid_coordinates
    z = load_golden('real_sequence_%s.npz' % seq)
    frames = sequence_frames(seq)
    dev = [(h, c, t, torch.from_numpy(img.view(np.int16)).cuda()) for h, c, t, img in frames]
    for keep_coordinates, resident in ((True, False), (False, True), (True, True)):
        sp = SequencePipeline(4256, 2832, pxPerDeg=10, keep_coordinates=keep_coordinates)
        got = sp.process(dev if resident else frames, keep_on_device=True)
        assert sp.plans == ['single-pass'] * len(frames)
        for k, r in enumerate(got):
            want = z['out_data_%d' % k]
            mean, mask = r['mean'].cpu().numpy(), r['mask'].cpu().numpy().astype(bool)
            assert mean.shape == want.shape, k
            assert np.array_equal(mask, np.isnan(want[..., 0])), k
            ok = ~mask
            assert np.array_equal(mean[..., :3][ok], want[..., :3][ok]), k
            assert np.max(np.abs(mean[..., 3][ok] - want[..., 3][ok])) < 1e-9, k
            c = grid_coordinates(r)
            assert np.array_equal(c['lat'], z['out_lat_%d' % k]) and np.array_equal(c['lon'], z['out_lon_%d' % k]), k
    sp = SequencePipeline(4256, 2832, arcsecPerPx=100, keep_coordinates=False)
    got = sp.process(dev, keep_on_device=True)
    assert sp.plans == ['single-pass'] * len(frames)
    k = len(frames) - 1
    hdr, cam, t, img = frames[k]
    m = ArraySpacecraftMapping(hdr, 110, img, cam, t, 'last', fastCenterCalculation=True).maskedByElevation(10)
    m.latsCenter                                            # materialise: the array route
    want = R.resample(m, arcsecPerPx=100)
    assert R.last_plan != 'single-pass'
    r = got[k]
    assert np.array_equal(r['img'].cpu().numpy().view(np.uint16), want.img.filled(0))
    assert np.array_equal(r['mask'].cpu().numpy().astype(bool), ma.getmaskarray(want.img)[..., 0])
    c = grid_coordinates(r)
    assert np.array_equal(c['lat_c'], want.latsCenter.data) and np.array_equal(c['lon'], want.lons.data)
