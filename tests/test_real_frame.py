"""
The reference's own test frame at full size: ISS030-E-102170_dc.jpg + .wcs (4256 x 2832; `draw_test.py` maps it), copies
of the two data files under tests/golden/resources/.  Expected values from the REAL reference (oracle/make_golden.py:
real_frame): fast centres -> maskedByElevation(10) -> _resample(pxPerDeg=10, 'mean'), the complete output grid.

CPU: the oracle equals the reference cell for cell.  GPU: what a user of the reference would write —
`resample(getMapping(imagePath, wcsPath, ...).maskedByElevation(10), pxPerDeg=10)` — and the frame pipeline with both plans.
"""
import os

import numpy as np
import numpy.ma as ma
import pytest

from conftest import GOLDEN, load_golden, oracle_frame

JPG = os.path.join(GOLDEN, 'resources', 'ISS030-E-102170_dc.jpg')
WCS = os.path.join(GOLDEN, 'resources', 'ISS030-E-102170_dc.wcs')


def check(res_img, res_mask, mean, z):
    want = z['out_data']
    assert np.array_equal(res_mask, np.isnan(want[..., 0]))
    ok = ~res_mask
    assert ok.sum() > 4000
    if mean is not None:
        assert np.array_equal(mean[..., :3][ok], want[..., :3][ok])                 # exact integer sums / counts
        assert np.max(np.abs(mean[..., 3][ok] - want[..., 3][ok])) < 1e-9           # elevation, fixed point
    assert np.array_equal(res_img[ok], z['out_img'][ok])


def test_oracle_equals_the_reference_on_its_own_test_frame():
    from oracle import ref_numpy as O
    from auromat_amd.fits import readHeader
    from auromat_amd.util.image import loadImage
    z = load_golden('real_frame_iss030.npz')
    hdr = readHeader(WCS)
    img = loadImage(JPG)
    assert img.shape == (2832, 4256, 3) and img.dtype == np.uint8
    g = oracle_frame(hdr, 110.0, z['cam'], z['m_geo'], z['m_sm'], fast=True)
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), 10)
    assert int((~center_mask).sum()) == int(z['n_valid'])
    bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
    assert np.array_equal(bbox, z['bbox']) and not disc
    data = np.dstack((img.astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    res = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), 110.0,
                          data, None, bbox, (10, 10), False, False)
    for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c'),
                 ('data', 'out_data')):
        assert np.array_equal(res[a], z[b], equal_nan=True), a


@pytest.mark.gpu
def test_the_references_call_sequence_on_its_own_test_frame():
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.resample import resample
    z = load_golden('real_frame_iss030.npz')
    m = getMapping(JPG, WCS, altitude=110, fastCenterCalculation=True)
    assert m.identifier == 'ISS030-E-102170_dc' and m.img.shape == (2832, 4256, 3)
    mm = m.maskedByElevation(10)
    assert int((~ma.getmaskarray(mm.latsCenter)).sum()) == int(z['n_valid'])
    bb = mm.boundingBox
    np.testing.assert_allclose([bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast], z['bbox'], rtol=0, atol=1e-9)
    r = resample(mm, pxPerDeg=10)
    r.checkGuarantees()
    assert np.array_equal(r.lats.data, z['out_lat']) and np.array_equal(r.lonsCenter.data, z['out_lon_c'])
    check(r.img.data, ma.getmaskarray(r.img)[..., 0], None, z)
    assert r.img.dtype == np.uint8


@pytest.mark.gpu
def test_user_guide_flow_takes_the_single_pass_plan():
    """`resample(getMapping(img, wcs).maskedByElevation(10), pxPerDeg=10)` with nothing else asked of the mapping: the mask
    is only remembered and resample() runs the fused kernel (ONE launch: georeferencing, mask, box, binning) — the
    reference's grid, cell for cell; the same for resampleMLatMLT; the arrays, asked for afterwards, carry the mask."""
    import auromat_amd.resample as R
    from auromat_amd.mapping.spacecraft import getMapping
    z = load_golden('real_frame_iss030.npz')
    for fast, name in ((True, 'real_frame_iss030.npz'), (False, 'real_frame_iss030_exact.npz')):
        z = load_golden(name)
        mm = getMapping(JPG, WCS, altitude=110, fastCenterCalculation=fast).maskedByElevation(10)
        assert mm._frame is None and mm._lazy_elev == 10.0
        r = R.resample(mm, pxPerDeg=10)
        assert R.last_plan == 'single-pass' and mm._frame is None
        r.checkGuarantees()
        assert np.array_equal(r.lats.data, z['out_lat']) and np.array_equal(r.lonsCenter.data, z['out_lon_c'])
        check(r.img.data, ma.getmaskarray(r.img)[..., 0], None, z)
        assert r.img.dtype == np.uint8
    zs = load_golden('real_frame_iss030_sm.npz')
    mm = getMapping(JPG, WCS, fastCenterCalculation=True).maskedByElevation(10)
    r = R.resampleMLatMLT(mm, pxPerDeg=10)
    assert R.last_plan == 'single-pass' and mm._frame is None
    check(r.img.data, ma.getmaskarray(r.img)[..., 0], None, zs)
    # ... and the arrays of the lazily masked mapping, on first use
    z = load_golden('real_frame_iss030.npz')
    assert int((~ma.getmaskarray(mm.latsCenter)).sum()) == int(z['n_valid'])
    mm.checkGuarantees()
    assert R.resample(mm, pxPerDeg=10) is not None and R.last_plan is None          # arrays exist: the array pipeline
    # a threshold that masks everything: the reference's ValueError, at first use
    none = getMapping(JPG, WCS, fastCenterCalculation=True).maskedByElevation(89.99)
    with pytest.raises(ValueError):
        R.resample(none, pxPerDeg=10)
    with pytest.raises(ValueError):
        none.lats


@pytest.mark.gpu
def test_both_plans_on_the_references_test_frame():
    from auromat_amd.fits import getShiftedSpacecraftPosition, readHeader
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.util.image import loadImage
    z = load_golden('real_frame_iss030.npz')
    hdr = readHeader(WCS)
    img = loadImage(JPG)
    cam, t, _ = getShiftedSpacecraftPosition(hdr)
    for fuse in (True, False):
        pipe = FramePipeline(4256, 2832, img_dtype=np.uint8)
        res = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, fuse=fuse)
        assert pipe.last_plan == ('single-pass' if fuse else 'two-pass')
        assert np.array_equal(res['lat'], z['out_lat']) and np.array_equal(res['lon_c'], z['out_lon_c'])
        check(res['img'], res['mask'], res['mean'], z)


# ---- ten consecutive real headers (seq/ISS029-E-8493 ... 8502.wcs, one frame every 3 s) with synthetic images ---------

def real_sequence():
    import glob
    from auromat_amd.fits import getSpacecraftPosition, readHeader
    from auromat_amd.synthetic import frame_image
    frames = []
    for k, path in enumerate(sorted(glob.glob(os.path.join(GOLDEN, 'resources', 'seq', '*.wcs')))):
        hdr = readHeader(path)
        cam, t = getSpacecraftPosition(hdr)
        frames.append((hdr, cam, t, frame_image(4256, 2832, seed=k)))
    return frames


def test_real_sequence_headers_are_the_references():
    z = load_golden('real_sequence_iss029.npz')
    frames = real_sequence()
    assert len(frames) == 10 == len(z['names'])
    assert [str(n) for n in z['names']] == ['ISS029-E-%d.wcs' % i for i in range(8493, 8503)]
    times = [f[2] for f in frames]
    assert all(2.9 < (b - a).total_seconds() < 3.1 for a, b in zip(times, times[1:]))
    assert all(f[0]['IMAGEW'] == 4256 and f[0]['IMAGEH'] == 2832 for f in frames)
    # neighbouring real frames are neighbours for the box hints too (their separately solved CD matrices differ in the
    # sixth digit), frames 20 s apart are not
    from auromat_amd.mapping.astrometry import frame_params
    from auromat_amd.pipeline import _close
    ps = [frame_params(h, 110, cam, t, True) for h, cam, t, _ in frames]
    assert all(_close(a, b) for a, b in zip(ps, ps[1:])) and _close(ps[0], ps[3]) and not _close(ps[0], ps[7])


@pytest.mark.gpu
def test_sequence_pipeline_on_real_consecutive_frames():
    """The sequence loop (batches of three, box hints between neighbouring frames, results left on the device) on ten real
    consecutive headers at full size: every frame's grid equals the reference's, cell for cell."""
    import torch
    from auromat_amd.pipeline import SequencePipeline
    from auromat_amd.resample import grid_coordinates
    z = load_golden('real_sequence_iss029.npz')
    frames = real_sequence()
    for keep_coordinates in (True, False):
        seq = SequencePipeline(4256, 2832, pxPerDeg=10, keep_coordinates=keep_coordinates)
        got = seq.process(frames, keep_on_device=True)
        # (a frame is prepared 6-7 frames = 20 s = 150 km of orbit ahead of the latest finished one: box hints at this
        # cadence are extrapolations of the two latest finished frames' boxes, which a ten-frame sequence is too short to
        # have in time: tests/test_gpu_sequence.py::test_box_hints_are_extrapolated_at_the_cadence_of_real_sequences)
        assert seq.plans == ['single-pass'] * 10
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


@pytest.mark.gpu
@pytest.mark.parametrize('magnetic', [False, True])
def test_sequence_pipeline_with_arcsec_per_px_on_real_consecutive_frames(magnetic):
    """`auromat-convert --resample --resolution 100` as the reference runs it (cli/convert.py:176-185: `map(partial(resample_,
    arcsecPerPx=R), mappings)`) on the ten real consecutive headers at full size, with a frame of empty sky and a frame with the
    pole in view inserted: SequencePipeline(arcsecPerPx=100) — the box-first plan, a resolution per frame — gives, frame for
    frame, the grid the mapping classes' array route gives (materialised mapping: bounding box from the arrays,
    plateCarreeResolution, two-pass binning), bit for bit; images resident on the device and from host memory."""
    import torch
    import auromat_amd.resample as R
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.pipeline import SequencePipeline
    from auromat_amd.resample import grid_coordinates
    from auromat_amd.synthetic import pole_frame
    frames = real_sequence()
    up = np.asarray(frames[3][1], dtype=np.float64) / np.linalg.norm(frames[3][1])
    sky = dict(frames[3][0], CRVAL1=float(np.rad2deg(np.arctan2(up[1], up[0])) % 360), CRVAL2=float(np.rad2deg(np.arcsin(up[2]))))
    frames.insert(4, (sky, frames[3][1], frames[3][2], frames[3][3]))
    if not magnetic:
        p_hdr, p_cam, p_t = pole_frame(4256, 2832)
        frames.insert(7, (p_hdr, p_cam, p_t, frames[0][3]))
    seq = SequencePipeline(4256, 2832, arcsecPerPx=100, keep_coordinates=False, magnetic=magnetic)
    assert len(seq.pipes) == 9
    got = seq.process(frames, keep_on_device=True)
    want_plans = ['single-pass'] * len(frames)
    want_plans[4] = 'empty'
    if not magnetic:
        want_plans[7] = 'pole-without-resolution'
    assert seq.plans == want_plans
    assert seq.ctx.last_variant()[0] == (4 if magnetic else 0)
    dev = [(h, c, t, torch.from_numpy(img.view(np.int16)).cuda()) for h, c, t, img in frames]
    again = seq.process(dev, keep_on_device=True)
    fn = R.resampleMLatMLT if magnetic else R.resample
    res = set()
    for k, (r, r2, (hdr, cam, t, img)) in enumerate(zip(got, again, frames)):
        if want_plans[k] != 'single-pass':
            assert r is None and r2 is None
            continue
        for key in ('mean', 'count', 'img', 'mask'):
            assert torch.equal(r[key], r2[key]) or key == 'mean' and np.array_equal(r[key].cpu().numpy(), r2[key].cpu().numpy(), equal_nan=True), (k, key)
        assert r['pxPerDeg'] == r2['pxPerDeg'] and r['pxPerDeg'][0] == 36.0
        res.add(r['pxPerDeg'][1])
        if k % 3 != 0 and k != len(frames) - 1:
            continue                                        # (the class route takes seconds per full-size frame)
        m = ArraySpacecraftMapping(hdr, 110, img, cam, t, 'f%d' % k, fastCenterCalculation=True).maskedByElevation(10)
        m.latsCenter                                        # materialise: the array route
        want = fn(m, arcsecPerPx=100)
        assert R.last_plan != 'single-pass'
        assert np.array_equal(r['img'].cpu().numpy().view(np.uint16), want.img.filled(0)), k
        assert np.array_equal(r['mask'].cpu().numpy().astype(bool), ma.getmaskarray(want.img)[..., 0]), k
        if not magnetic:
            c = grid_coordinates(r)
            assert np.array_equal(c['lat_c'], want.latsCenter.data) and np.array_equal(c['lon'], want.lons.data), k
        el = r['mean'][..., 3].cpu().numpy()
        assert np.max(np.abs(el[~ma.getmaskarray(want.elevation)] - want.elevation.compressed())) < 1e-9, k
    assert len(res) >= 5                                    # every frame really had a resolution of its own


@pytest.mark.gpu
def test_convert_driver_on_the_references_test_frame(tmp_path):
    """`auromat-convert --data <folder with ISS030-E-102170_dc.jpg + .wcs> --resample --grid geo --px-per-deg 10`: the file
    it writes holds the reference's grid (image through Pillow, header cards parsed from the FITS file, single-pass
    pipeline, netCDF classic writer and reader)."""
    from auromat_amd.cli.convert import main
    from auromat_amd.export import _nc4
    z = load_golden('real_frame_iss030.npz')
    out = str(tmp_path / 'converted')
    main(['--data', os.path.join(GOLDEN, 'resources'), '--format', 'netcdf', '--resample', '--min-elevation', '10', '--grid', 'geo', '--px-per-deg', '10',
          '--out', out, '--without-mag'])
    assert os.listdir(out) == ['ISS030-E-102170_dc.nc']
    f = _nc4.open_file(os.path.join(out, 'ISS030-E-102170_dc.nc'))
    assert np.array_equal(f.vars['lat'].data, z['out_lat_c'][:, 0]) and np.array_equal(f.vars['lon'].data, z['out_lon_c'][0, :])
    mask = z['out_img_mask'][..., 0]
    for c, name in enumerate(('img_red', 'img_green', 'img_blue')):
        v = f.vars[name]
        got = ma.masked_equal(v.data, v.attrs['_FillValue'])
        assert np.array_equal(ma.getmaskarray(got), mask), name
        assert np.array_equal(got.filled(0)[~mask], z['out_img'][..., c][~mask]), name


# ---- the same frame with exact centres (getMapping's default), and on the MLat / MLT grid ----------------------------

@pytest.mark.gpu
def test_real_frame_with_exact_centres():
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.resample import resample
    from auromat_amd.fits import getShiftedSpacecraftPosition, readHeader
    from auromat_amd.util.image import loadImage
    z = load_golden('real_frame_iss030_exact.npz')
    mm = getMapping(JPG, WCS).maskedByElevation(10)                   # fastCenterCalculation=False is the default
    mm.checkGuarantees()
    assert int((~ma.getmaskarray(mm.latsCenter)).sum()) == int(z['n_valid'])
    r = resample(mm, pxPerDeg=10)
    assert np.array_equal(r.lats.data, z['out_lat']) and np.array_equal(r.lonsCenter.data, z['out_lon_c'])
    check(r.img.data, ma.getmaskarray(r.img)[..., 0], None, z)
    hdr, img = readHeader(WCS), loadImage(JPG)
    cam, t, _ = getShiftedSpacecraftPosition(hdr)
    for fuse in (True, False):
        pipe = FramePipeline(4256, 2832, img_dtype=np.uint8)
        res = pipe.run(hdr, 110, cam, t, img=img, fast=False, min_elevation=10, pxPerDeg=10, fuse=fuse)
        assert pipe.last_plan == ('single-pass' if fuse else 'two-pass')
        check(res['img'], res['mask'], res['mean'], z)


@pytest.mark.gpu
def test_real_frame_on_the_mlat_mlt_grid():
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.resample import resampleMLatMLT
    from auromat_amd.fits import getShiftedSpacecraftPosition, readHeader
    from auromat_amd.util.image import loadImage
    z = load_golden('real_frame_iss030_sm.npz')
    mm = getMapping(JPG, WCS, fastCenterCalculation=True).maskedByElevation(10)
    r = resampleMLatMLT(mm, pxPerDeg=10)
    assert r.img.shape[:2] == z['out_img'].shape[:2]
    check(r.img.data, ma.getmaskarray(r.img)[..., 0], None, z)
    hdr, img = readHeader(WCS), loadImage(JPG)
    cam, t, _ = getShiftedSpacecraftPosition(hdr)
    for fuse, geo in ((True, True), (False, True), (True, False)):
        # (with_geo=False: the MLat / MLT-only mode of the fused kernel — no ECEF -> geodetic step, five arrays kept)
        pipe = FramePipeline(4256, 2832, img_dtype=np.uint8, with_mag=True, with_geo=geo)
        res = pipe.run(hdr, 110, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, fuse=fuse, magnetic=True)
        assert pipe.last_plan == ('single-pass' if fuse else 'two-pass')
        assert not fuse or pipe.ctx.last_variant()[0] == (1 if geo else 4)
        assert np.array_equal(res['lat'], z['out_lat']) and np.array_equal(res['lon_c'], z['out_lon_c'])
        check(res['img'], res['mask'], res['mean'], z)


@pytest.mark.gpu
@pytest.mark.parametrize('altitude', [100, 120])
def test_real_frame_on_the_mlat_mlt_grid_other_shells(altitude):
    """BASELINE configs[3]'s other two shells (100 / 120 km) on the reference's own test frame, pinned to the REAL reference
    (real_frame_iss030_sm_{100,120}km.npz: its resampleMLatMLT chain at that altitude): both plans of the frame pipeline and
    the class route, cell for cell."""
    import auromat_amd.resample as R
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.pipeline import FramePipeline
    from auromat_amd.fits import getShiftedSpacecraftPosition, readHeader
    from auromat_amd.util.image import loadImage
    z = load_golden('real_frame_iss030_sm_%dkm.npz' % altitude)
    hdr, img = readHeader(WCS), loadImage(JPG)
    cam, t, _ = getShiftedSpacecraftPosition(hdr)
    for fuse, geo in ((True, True), (False, True), (True, False)):
        pipe = FramePipeline(4256, 2832, img_dtype=np.uint8, with_mag=True, with_geo=geo)
        res = pipe.run(hdr, altitude, cam, t, img=img, fast=True, min_elevation=10, pxPerDeg=10, fuse=fuse, magnetic=True)
        assert pipe.last_plan == ('single-pass' if fuse else 'two-pass')
        assert not fuse or pipe.ctx.last_variant()[0] == (1 if geo else 4)
        assert np.array_equal(res['lat'], z['out_lat']) and np.array_equal(res['lon_c'], z['out_lon_c'])
        check(res['img'], res['mask'], res['mean'], z)
    mm = getMapping(JPG, WCS, altitude=altitude, fastCenterCalculation=True).maskedByElevation(10)
    r = R.resampleMLatMLT(mm, pxPerDeg=10)
    assert R.last_plan == 'single-pass'
    from auromat_amd._native import Context
    assert Context.current().last_variant()[0] == 4          # the class route: MLat / MLT only (grids-only pipeline)
    check(r.img.data, ma.getmaskarray(r.img)[..., 0], None, z)


@pytest.mark.gpu
@pytest.mark.parametrize('magnetic', [False, True])
def test_arcsec_per_px_on_the_real_frame_takes_the_box_first_plan(magnetic):
    """The reference's own call form — `resample(mapping, arcsecPerPx=100)` in test/mapping_test.py:24-42 and
    `partial(resample_, arcsecPerPx=args.resolution)` in cli/convert.py:176-185 — on its own test frame at full size: on a
    mapping nobody has materialised it runs the box-first plan (box pass, plateCarreeResolution, single-pass launch) and
    gives the grid of the array route (a materialised mapping: bounding box from the arrays, two-pass binning) bit for bit,
    on the geodetic and on the MLat / MLT grid, with and without maskedByElevation."""
    import auromat_amd.resample as R
    from auromat_amd._native import Context
    from auromat_amd.mapping.spacecraft import getMapping
    fn = R.resampleMLatMLT if magnetic else R.resample
    for elev in (10, None):
        mm = getMapping(JPG, WCS, fastCenterCalculation=True)
        ref = getMapping(JPG, WCS, fastCenterCalculation=True)
        if elev is not None:
            mm, ref = mm.maskedByElevation(elev), ref.maskedByElevation(elev)
        ref.latsCenter                                   # materialise: the array route
        want = fn(ref, arcsecPerPx=100)
        assert R.last_plan != 'single-pass'
        got = fn(mm, arcsecPerPx=100)
        assert R.last_plan == 'single-pass'
        assert Context.current().last_variant()[0] == (4 if magnetic else 0)
        assert got.img.shape == want.img.shape and got.img.shape[0] > 200
        for name in ('lats', 'lons', 'latsCenter', 'lonsCenter', 'img', 'elevation'):
            a, b = getattr(got, name), getattr(want, name)
            assert np.array_equal(ma.getmaskarray(a), ma.getmaskarray(b)), name
            if name == 'elevation':
                assert np.max(np.abs(a.compressed() - b.compressed())) < 1e-9
            else:
                assert np.array_equal(a.filled(0), b.filled(0)), name


@pytest.mark.gpu
def test_folder_provider_on_the_references_resources():
    """SpacecraftMappingProvider over the folder with the test frame (image + .wcs, as the reference's ISS provider leaves
    them), wrapped like `auromat-convert` wraps its provider: masked by elevation, resampled — the reference's grid."""
    from auromat_amd.mapping.mapping import MaskByElevationProvider
    from auromat_amd.mapping.spacecraft import SpacecraftMappingProvider
    from auromat_amd.resample import resample
    z = load_golden('real_frame_iss030.npz')
    prov = SpacecraftMappingProvider(os.path.join(GOLDEN, 'resources'), fastCenterCalculation=True)
    assert len(prov) == 1 and prov.imageFileExtension == 'jpg' and prov.ids == ['ISS030-E-102170_dc']
    t0, t1 = prov.range
    assert t0 == t1 and t0.isoformat() == str(z['time_iso'])          # the shifted photo time
    masked = MaskByElevationProvider(prov, 10)
    m = masked.get(t0)
    assert m.identifier == 'ISS030-E-102170_dc'
    assert int((~ma.getmaskarray(m.latsCenter)).sum()) == int(z['n_valid'])
    r = resample(m, pxPerDeg=10)
    check(r.img.data, ma.getmaskarray(r.img)[..., 0], None, z)
    assert [q.identifier for q in masked.getSequence()] == ['ISS030-E-102170_dc']


@pytest.mark.gpu
@pytest.mark.parametrize('fmt', ['cdf', 'netcdf'])
def test_the_references_export_round_trip_on_its_test_frame(fmt, tmp_path):
    """test/export_cdf_test.py:26-49 and test/export_netcdf_test.py:28-52: the reference's test frame, unresampled, written
    with the default options, the file's variables by name, read back as a mapping, checkGuarantees(), check_equal
    (export_netcdf_test.py:72-88: arrays equal, the same bounding box, elevation to 5 decimals — it is stored as a float32
    zenith angle).  The CDF container is this package's own (export/_cdf3.py: no CDF library in the image)."""
    from numpy.testing import assert_array_almost_equal, assert_array_equal
    from auromat_amd.mapping.spacecraft import getMapping
    mapping = getMapping(JPG, WCS, fastCenterCalculation=True)
    mapping.checkGuarantees()
    path = str(tmp_path / ('frame.' + ('cdf' if fmt == 'cdf' else 'nc')))
    names = {'lat', 'lon', 'altitude', 'lat_bounds', 'lon_bounds', 'mlat', 'mlt', 'mlat_bounds', 'mlt_bounds', 'mcrs', 'img_red',
             'img_green', 'img_blue', 'zenith_angle', 'camera_pos', 'crs'}
    if fmt == 'cdf':
        from auromat_amd.export import _cdf3
        from auromat_amd.export.cdf import write
        from auromat_amd.mapping.cdf import CDFMapping as Back
        write(path, mapping)
        assert set(_cdf3.Reader(path).vars) == names | {'Epoch'}
    else:
        from auromat_amd.export import _nc4
        from auromat_amd.export.netcdf import write
        from auromat_amd.mapping.netcdf import NetCDFMapping as Back
        write(path, mapping)
        assert set(_nc4.open_file(path).vars) == names | {'time'}
    back = Back(path)
    back.checkGuarantees()
    assert_array_equal(back.img.shape, mapping.img.shape)
    assert_array_equal(back.lats.shape, mapping.lats.shape)
    for key in ('img', 'lats', 'lons', 'latsCenter', 'lonsCenter'):
        a, b = getattr(back, key), getattr(mapping, key)
        assert np.array_equal(ma.getmaskarray(a), ma.getmaskarray(b)), key
        assert np.array_equal(ma.getdata(a)[~ma.getmaskarray(a)], ma.getdata(b)[~ma.getmaskarray(b)]), key
    bb, want = back.boundingBox, mapping.boundingBox
    assert (bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast) == (want.latSouth, want.lonWest, want.latNorth, want.lonEast)
    assert_array_almost_equal(back.elevation.filled(-1), mapping.elevation.filled(-1), decimal=5)
    assert back.photoTime == mapping.photoTime
    assert np.array_equal(back.cameraPosGCRS, mapping.cameraPosGCRS)
