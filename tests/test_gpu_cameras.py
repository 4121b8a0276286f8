"""
GPU parity tests of the other camera models that feed the same intersection + geodetic steps:
the FMI MIRACLE all-sky fisheye mapping (reference mapping/miracle.py) and the THEMIS altitude reprojection
(mapping/themis.py:224-253) — against outputs of the real reference (tests/golden/miracle_*.npz,
themis_reproject.npz) and the oracle, and through the reference's own mapping_test.py:45-51 call sequence.
"""
import ctypes as C
import os
from datetime import datetime

import numpy as np
import numpy.ma as ma
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

TOL_DEG = 1e-6          # north_star tolerance; the kernels sit at rounding level, asserted below as 1e-10


def nan_close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    err = np.max(np.abs(a[ok] - b[ok]), initial=0.0)
    assert err <= tol, 'max abs error %.3e > %.1e' % (err, tol)


def cal_from(z):
    from auromat_amd.mapping.miracle import CalibrationData
    from auromat_amd.mapping.mapping import BoundingBox
    lat, lon = float(z['cal_lat']), float(z['cal_lon'])
    bb = BoundingBox(latSouth=lat + float(z['cal_lat_minus']), lonWest=lon + float(z['cal_lon_minus']),
                     latNorth=lat + float(z['cal_lat_plus']), lonEast=lon + float(z['cal_lon_plus']))
    return CalibrationData(station=str(z['cal_station']), validFrom=None, validTo=None, lat=lat, lon=lon,
                           xc=float(z['cal_xc']), yc=float(z['cal_yc']), k=float(z['cal_k']),
                           rotation=float(z['cal_rotation']), boundingBoxSimple=bb)


def gray_image(n, seed, caption=0):
    rng = np.random.RandomState(seed)
    return rng.randint(0, 256, size=(n + caption, n)).astype(np.uint8)


@pytest.mark.parametrize('name', ['miracle_sod64.npz', 'miracle_kev96.npz'])
def test_allsky_mapping_vs_reference(name):
    from auromat_amd.mapping.miracle import MIRACLEMapping
    z = load_golden(name)
    n = int(z['size'])
    t = datetime(2012, 3, 4, 17, 19, 0)
    for prefix, off in (('', 0.5), ('np16_', 0.0)):
        m = MIRACLEMapping(cal_from(z), gray_image(n, 1), t, float(z['altitude']), center_offset=off)
        for k, got in (('lat', m.lats), ('lon', m.lons), ('lat_c', m.latsCenter), ('lon_c', m.lonsCenter),
                       ('elev', m.elevation)):
            assert not ma.getmaskarray(got).any()
            nan_close(got.data, z[prefix + k], 1e-10)
        az, el = m.calculateAzEl(center=False)
        nan_close(az, z[prefix + 'az'], 1e-10)
        nan_close(el, z[prefix + 'el_corner'], 1e-10)
        nan_close(m.azimuthCenter, z[prefix + 'az_c'], 1e-10)
        nan_close(m.cameraToPixelCornerDirection, z[prefix + 'dirs'], 1e-13)
        nan_close(m.cameraToPixelCenterDirection, z[prefix + 'dirs_c'], 1e-13)
        nan_close(m.cameraPosGEO, z['cam_geo'], 1e-9)
        nan_close(m.cameraPosGCRS, z['cam_gcrs'], 1e-9)
    # getMapping's maskedByElevation(0.1) (miracle.py:365): the reference's masks
    m = MIRACLEMapping(cal_from(z), gray_image(n, 1), t, float(z['altitude'])).maskedByElevation(0.1)
    assert np.array_equal(ma.getmaskarray(m.lats), z['corner_mask'])
    assert np.array_equal(ma.getmaskarray(m.latsCenter), z['center_mask'])
    assert np.array_equal(ma.getmaskarray(m.img)[:, :, 0], z['center_mask'])
    m.checkGuarantees()


def test_allsky_native_size_vs_reference_samples_and_oracle():
    from oracle import ref_numpy as O
    from auromat_amd.mapping.miracle import MIRACLEMapping
    z = load_golden('miracle_sod512.npz')
    m = MIRACLEMapping(cal_from(z), gray_image(512, 2), datetime(2012, 3, 4, 17, 19, 0), 110)
    step = int(z['step'])
    got = dict(lat=m.lats.data, lon=m.lons.data, lat_c=m.latsCenter.data, lon_c=m.lonsCenter.data,
               elev=m.elevation.data, az=m.azimuthCorner, az_c=m.azimuthCenter, el_corner=m.elevationCorner)
    for k, a in got.items():
        nan_close(a[::step, ::step], z[k], 1e-10)
        d = z['digest_' + k]
        assert a.size == d[0] and abs(a.sum() - d[1]) <= 1e-9 * max(1.0, abs(d[1]))
        assert abs(a.min() - d[2]) <= 1e-10 and abs(a.max() - d[3]) <= 1e-10
    cal = {k: float(z['cal_' + k]) for k in ('lat', 'lon', 'xc', 'yc', 'k', 'rotation')}
    g = O.allsky_georef(512, cal, 110.0)
    for k, a in got.items():
        nan_close(a, g[k], 1e-10)


def test_reference_miracle_mapping_test_call_sequence():
    """mapping_test.py:45-51,68-71: getMapping(image next to cal.txt) -> checkGuarantees -> maskedByElevation(10)
    -> checkGuarantees -> resample(arcsecPerPx=100, method='mean') -> checkGuarantees; the image is a synthetic
    512 x 600 grayscale frame (caption rows below the square image, as the fixture's JPEG has)."""
    from oracle import ref_numpy as O
    from auromat_amd.mapping import miracle
    from auromat_amd.resample import plateCarreeResolution, resample
    img = gray_image(512, 3, caption=88)
    path = os.path.join(GOLDEN, 'miracle', 'SOD120304_171900_557_1000.jpg')       # only its name and folder are used
    m = miracle.getMapping(path, image=img)
    assert m.identifier == 'SOD.2012.03.04.17.19.00' and m.altitude == 110
    assert m.img.shape == (512, 512, 3)
    m.checkGuarantees()
    m2 = m.maskedByElevation(10)
    m2.checkGuarantees()
    assert ma.getmaskarray(m2.latsCenter).sum() > ma.getmaskarray(m.latsCenter).sum()
    m3 = resample(m2, arcsecPerPx=100, method='mean')
    m3.checkGuarantees()
    m3.checkPlateCarree()
    # against the oracle: same grid, same counts up to edge cases of 1e-11 deg, same integer image means
    z = load_golden('miracle_sod512.npz')
    cal = {k: float(z['cal_' + k]) for k in ('lat', 'lon', 'xc', 'yc', 'k', 'rotation')}
    g = O.allsky_georef(512, cal, 110.0)
    corner_mask, center_mask = O.mask_by_elevation(g['elev'], np.isnan(g['lat']), 10)
    assert np.array_equal(ma.getmaskarray(m2.latsCenter), center_mask)
    assert np.array_equal(ma.getmaskarray(m2.lats), corner_mask)
    bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
    bb = m2.boundingBox
    np.testing.assert_allclose([bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast], bbox, atol=1e-9)
    ppd = plateCarreeResolution(bb, 100)
    rgb = np.repeat(img[:512, :, None], 3, 2)
    data = np.dstack((rgb.astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    want = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), 110,
                           data, None, bbox, ppd, disc, False)
    assert want['data'].shape[:2] == m3.latsCenter.shape
    filled = ~np.isnan(want['data'][..., 0])
    assert int((filled != ~ma.getmaskarray(m3.latsCenter)).sum()) <= 2
    both = filled & ~ma.getmaskarray(m3.latsCenter)
    expect_img, _ = O.finalize_image(want['data'][..., :3], np.uint8)
    assert int((np.asarray(m3.img.data)[both] != expect_img[both]).sum()) <= 6       # <= 2 cells x 3 channels
    assert int((np.abs(m3.elevation.data[both] - want['data'][..., 3][both]) > 1e-9).sum()) <= 2


def test_allsky_simple_mode_grid():
    """simple=True: constant lat-lon grid between the calibration's offsets, north at the top (miracle.py:198-212;
    the reference cannot run it, see oracle/make_golden.py); elevation still comes from the fisheye model."""
    from auromat_amd.mapping import miracle
    cal = miracle.getCalibrationData(os.path.join(GOLDEN, 'miracle', 'cal.txt'), 'KEV', datetime(2012, 3, 4))
    assert (cal.lat, cal.lon, cal.xc, cal.yc, cal.k, cal.rotation) == (69.76, 27.01, 249.5, 273.8, 154.59, 0.07049)
    with pytest.raises(ValueError):
        miracle.getCalibrationData(os.path.join(GOLDEN, 'miracle', 'cal.txt'), 'KEV', datetime(2013, 3, 4))
    n = 64
    m = miracle.MIRACLEMapping(cal, gray_image(n, 4), datetime(2012, 3, 4, 17, 21), 95, simple=True)
    assert m.altitude == 110
    lat, lon = m.lats.data, m.lons.data
    np.testing.assert_allclose(lat[:, 0], np.linspace(69.76 + 2.7, 69.76 - 2.7, n + 1), rtol=0, atol=1e-12)
    np.testing.assert_allclose(lon[0, :], np.linspace(27.01 - 7.9, 27.01 + 7.9, n + 1), rtol=0, atol=1e-12)
    assert np.all(lat == lat[:, :1]) and np.all(lon == lon[:1, :])
    np.testing.assert_allclose(m.latsCenter.data[:, 0], (lat[:-1, 0] + lat[1:, 0]) / 2, atol=1e-12)
    np.testing.assert_allclose(m.lonsCenter.data[0, :], (lon[0, :-1] + lon[0, 1:]) / 2, atol=1e-12)
    full = miracle.MIRACLEMapping(cal, gray_image(n, 4), datetime(2012, 3, 4, 17, 21), 110)
    assert np.array_equal(m.elevation.data, full.elevation.data)
    m.checkGuarantees()


def test_themis_reproject_vs_reference():
    from auromat_amd.mapping.themis import reproject
    z = load_golden('themis_reproject.npz')
    station = tuple(z['station'])
    for h in (90, 150):
        la, lo = reproject(station, z['lat_ref'], z['lon_ref'], float(z['height_ref']), h)
        nan_close(la, z['lat_%d' % h], 1e-10)
        nan_close(lo, z['lon_%d' % h], 1e-10)
    # masked input = missing input; higher shells spread the footprint away from the station
    la, lo = reproject(station, ma.masked_invalid(z['lat_ref']), ma.masked_invalid(z['lon_ref']), 110, 150)
    nan_close(la, z['lat_150'], 1e-10)
    ok = ~np.isnan(la)

    def arc(lat, lon):          # angle at the Earth's centre between the station and (lat, lon)
        a, b, c, d = np.deg2rad(lat), np.deg2rad(lon), np.deg2rad(station[0]), np.deg2rad(station[1])
        return np.arccos(np.clip(np.sin(a) * np.sin(c) + np.cos(a) * np.cos(c) * np.cos(b - d), -1, 1))
    assert np.all(arc(la[ok], lo[ok]) >= arc(z['lat_90'][ok], z['lon_90'][ok]) - 1e-12)


def test_themis_mapping_brightness_scaling():
    from auromat_amd.mapping.themis import ThemisMapping, bytscl, reproject
    z = load_golden('themis_reproject.npz')
    lat, lon = z['lat_150'], z['lon_150']
    n = lat.shape[0] - 1
    rng = np.random.RandomState(5)
    img = rng.randint(2000, 9000, size=(n, n)).astype(np.uint16)
    lat_c = (lat[:-1, :-1] + lat[1:, 1:]) / 2
    lon_c = (lon[:-1, :-1] + lon[1:, 1:]) / 2
    elev = np.where(np.isnan(lat_c), np.nan, 45.0)
    m = ThemisMapping(lat, lon, lat_c, lon_c, elev, 150, img, [1000.0, 2000.0, 6000.0],
                      datetime(2013, 9, 26, 5, 3), 'rank', minBrightness=2500, maxBrightness=8000)
    assert m.identifier == 'rank.2013.09.26.05.03.00'
    m.checkGuarantees()
    rgb = m.rgb_unmasked
    assert rgb.shape == (n, n, 3) and rgb.dtype == np.uint8
    expect = bytscl(img, min_=2500, max_=8000, top=255)
    assert np.array_equal(rgb[:, :, 0], expect.astype(np.uint8))
    assert bytscl(np.array([0.0, 5.0, 10.0])).tolist() == [0, 127, 255]


def test_new_entry_points_error_behaviour():
    from auromat_amd._native import AllSkyParams, Context, NativeError, ptr
    ctx = Context.current()
    p = AllSkyParams()
    out = ctx.empty((16,))
    with pytest.raises(NativeError):
        ctx.call('amt_georef_allsky', None, 1, ptr(out), None, None, None, None)
    p.size, p.k, p.a, p.b, p.a0, p.b0 = 0, 1.0, 6488.0, 6466.0, 6378.0, 6356.0
    with pytest.raises(NativeError):
        ctx.call('amt_georef_allsky', C.byref(p), 1, ptr(out), None, None, None, None)
    p.size, p.k = 3, 0.0
    with pytest.raises(NativeError):
        ctx.call('amt_georef_allsky', C.byref(p), 1, ptr(out), None, None, None, None)
    p.k = 1.0
    ctx.call('amt_georef_allsky', C.byref(p), 1, None, None, None, None, None)       # nothing asked for: no-op
    with pytest.raises(NativeError):
        ctx.call('amt_reproject_altitude', 60.0, 20.0, None, None, 4, 110.0, 150.0, 6378.137, 6356.752, None, None)
    ctx.call('amt_reproject_altitude', 60.0, 20.0, None, None, 0, 110.0, 150.0, 6378.137, 6356.752, None, None)
