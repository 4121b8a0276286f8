"""
BASELINE.json configs[0]: the single 512x512 synthetic frame, known camera pose, 110 km
shell — header CRPIX = (256.5, 2832/s/2 + 0.5), CD scaled by s = 4256/512, image RandomState(1), fast and exact
centres.  The expected values come from the REAL reference (oracle/make_golden.py: config1): every 4th sample and
whole-array digests of the thirteen coordinate arrays, the masks, and the complete output of
maskedByElevation(10) -> _resample(pxPerDeg=10, 'mean').

CPU: the oracle equals the reference bit for bit.  GPU: the HIP path through the C ABI against the same fixture.
"""
from datetime import datetime

import numpy as np
import numpy.ma as ma
import pytest

from conftest import header_from, load_golden

ARRAYS = ('lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlt', 'mlat_c', 'mlt_c')


def parse(s):
    return datetime.strptime(str(s), '%Y-%m-%dT%H:%M:%S.%f')


def digest(a):
    a = np.asarray(a, dtype=np.float64)
    ok = ~np.isnan(a)
    return np.array([ok.sum(), a[ok].sum(), a[ok].min(), a[ok].max(), np.abs(a[ok]).sum()], dtype=np.float64)


def unpack(z, key, shape):
    return np.unpackbits(z[key])[:shape[0] * shape[1]].reshape(shape).astype(bool)


def config1_image():
    return np.random.RandomState(1).randint(0, 65535, (512, 512, 3)).astype(np.uint16)


def test_header_is_the_one_survey_8d_states():
    z = load_golden('config1_fast.npz')
    hdr = header_from(z)
    s = 4256 / 512
    assert (hdr['IMAGEW'], hdr['IMAGEH']) == (512, 512)
    assert hdr['CRPIX1'] == 256.5 and hdr['CRPIX2'] == 2832 / s / 2 + 0.5
    assert hdr['CD1_1'] == s * -0.00912247310646 and hdr['CD2_1'] == s * 0.00250608809647
    assert np.array_equal(z['cam'], [-4809.524217485676, 524.8117887762777, 4729.265809729493])
    assert str(z['time_iso']) == '2012-01-25T09:26:55.060000' and float(z['altitude']) == 110


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_oracle_equals_the_reference_at_config1(mode):
    from oracle import ref_numpy as O
    z = load_golden('config1_%s.npz' % mode)
    hdr = header_from(z)
    step = int(z['step'])
    g = O.georef_frame(hdr, 110.0, z['cam'], z['m_geo'], z['m_sm'], fast=mode == 'fast')
    for k in ARRAYS:
        assert np.array_equal(g[k][::step, ::step], z[k], equal_nan=True), k
        assert np.array_equal(digest(g[k]), z['digest_' + k]), k
    corner_nan, center_nan = np.isnan(g['lat']), np.isnan(g['lat_c'])
    if mode == 'exact':
        corner_nan, center_nan = O.sanitize_masks(corner_nan, center_nan)
    assert np.array_equal(corner_nan, unpack(z, 'corner_mask', (513, 513)))
    assert np.array_equal(center_nan, unpack(z, 'center_mask', (512, 512)))
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], corner_nan, 10)
    assert np.array_equal(corner_mask, unpack(z, 'e10_corner_mask', (513, 513)))
    assert np.array_equal(center_mask, unpack(z, 'e10_center_mask', (512, 512)))
    bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
    assert np.array_equal(bbox, z['bbox']) and not disc
    data = np.dstack((config1_image().astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    res = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), 110.0,
                          data, None, bbox, (10, 10), False, False)
    for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c'),
                 ('data', 'out_data')):
        assert np.array_equal(res[a], z[b], equal_nan=True), a
    # SURVEY appendix A anchors for this frame: 72.2 % of the centres valid, lat 46.613 ... 61.058
    assert abs((~center_nan).mean() - 0.722) < 1e-3
    assert abs(np.nanmin(g['lat_c']) - 46.613) < 0.05 and abs(np.nanmax(g['lat']) - 61.058) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_hip_path_equals_the_reference_at_config1(mode):
    from auromat_amd.pipeline import FramePipeline
    z = load_golden('config1_%s.npz' % mode)
    hdr = header_from(z)
    step = int(z['step'])
    img = config1_image()
    t = parse(z['time_iso'])
    for fuse in (False, True):
        pipe = FramePipeline(512, 512, with_mag=True)
        res = pipe.run(hdr, 110, z['cam'], t, img=img, fast=mode == 'fast', min_elevation=10, pxPerDeg=10, fuse=fuse)
        assert pipe.last_plan == ('single-pass' if fuse else 'two-pass')
        got = pipe.host_arrays()
        for k in ARRAYS:
            tol = 1e-6 * (24 / 360) if k.startswith('mlt') else 1e-6        # north star: 1e-6 deg
            a, b = got[k][::step, ::step], z[k]
            assert np.array_equal(np.isnan(a), np.isnan(b)), k
            assert np.nanmax(np.abs(a - b)) <= tol, (k, np.nanmax(np.abs(a - b)))
            d = digest(got[k])
            assert d[0] == z['digest_' + k][0], k                           # identical NaN count over the whole array
            assert abs(d[1] - z['digest_' + k][1]) <= tol * d[0], k
            assert abs(d[2] - z['digest_' + k][2]) <= tol and abs(d[3] - z['digest_' + k][3]) <= tol, k
        bb = pipe.bounding_box()
        np.testing.assert_allclose([bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast], z['bbox'], rtol=0, atol=1e-9)
        # the resampled grid: coordinates bit-identical, counts-derived masks and integer means exactly equal
        for a, b in (('lat', 'out_lat'), ('lon', 'out_lon'), ('lat_c', 'out_lat_c'), ('lon_c', 'out_lon_c')):
            assert np.array_equal(res[a], z[b]), a
        want = z['out_data']
        assert np.array_equal(res['mask'], np.isnan(want[..., 0]))
        assert np.array_equal(res['mask'], z['out_img_mask'][..., 0])
        ok = ~res['mask']
        assert np.array_equal(res['mean'][..., :3][ok], want[..., :3][ok])                  # exact integer sums / counts
        assert np.max(np.abs(res['mean'][..., 3][ok] - want[..., 3][ok])) < 1e-9            # elevation, fixed point
        assert np.array_equal(res['img'][ok], z['out_img'][ok])


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_mapping_classes_at_config1(mode):
    """The class API (lazy properties, masks, maskedByElevation, resample) on the same frame."""
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.resample import resample
    z = load_golden('config1_%s.npz' % mode)
    hdr = header_from(z)
    step = int(z['step'])
    m = ArraySpacecraftMapping(hdr, 110, config1_image(), z['cam'], parse(z['time_iso']), 'c1',
                               fastCenterCalculation=mode == 'fast')
    assert np.array_equal(ma.getmaskarray(m.lats), unpack(z, 'corner_mask', (513, 513)))
    assert np.array_equal(ma.getmaskarray(m.latsCenter), unpack(z, 'center_mask', (512, 512)))
    if mode == 'exact':
        assert np.array_equal(ma.getmaskarray(m.img)[..., 0], unpack(z, 'img_mask', (512, 512)))
    assert np.nanmax(np.abs(m.lats.filled(np.nan)[::step, ::step] - z['lat'])) < 1e-6
    assert np.nanmax(np.abs(m.elevation.filled(np.nan)[::step, ::step] - z['elev'])) < 1e-6
    mm = m.maskedByElevation(10)
    mm.checkGuarantees()
    assert np.array_equal(ma.getmaskarray(mm.lats), unpack(z, 'e10_corner_mask', (513, 513)))
    assert np.array_equal(ma.getmaskarray(mm.latsCenter), unpack(z, 'e10_center_mask', (512, 512)))
    r = resample(mm, pxPerDeg=10)
    r.checkGuarantees()
    assert np.array_equal(r.lats.data, z['out_lat']) and np.array_equal(r.lonsCenter.data, z['out_lon_c'])
    assert np.array_equal(ma.getmaskarray(r.img)[..., 0], z['out_img_mask'][..., 0])
    ok = ~z['out_img_mask'][..., 0]
    assert np.array_equal(r.img.data[ok], z['out_img'][ok])
