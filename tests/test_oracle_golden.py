"""
Pins the CPU oracle (oracle/ref_numpy.py) to the reference:
 * outputs of the real reference run in the build container (tests/golden/*.npz,
   made by oracle/make_golden.py), and
 * the reference's own known-answer tests (tests/golden/known_answers.json:
   intersection_test.py:26-137, transform_test.py:70-129).
CPU only.
"""
import json
import os
from datetime import datetime

import numpy as np
import pytest

from conftest import GOLDEN, header_from, load_golden
from oracle import ref_numpy as O

SMALL = ['georef_small_%s_%s.npz' % (p, m) for p in ('iss030', 'iss029') for m in ('fast', 'exact')]


def same(a, b, tol=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    if tol == 0.0:
        assert np.array_equal(a[ok], b[ok])
    else:
        assert np.max(np.abs(a[ok] - b[ok]), initial=0.0) <= tol


def parse(s):
    return datetime.strptime(str(s), '%Y-%m-%dT%H:%M:%S.%f')


def test_host_scalars_bit_exact():
    z = load_golden('host_scalars.npz')
    for i, d in enumerate(z['dates']):
        et = O.date2es(parse(d))
        assert et == z['et'][i]
        for name, fn in [('m_geo', O.mat_j2000_to_geo), ('m_sm', O.mat_j2000_to_sm), ('m_geo_sm', O.mat_geo_to_sm),
                         ('m_P', O.mat_P), ('m_T1', O.mat_T1), ('m_T2', O.mat_T2), ('m_T3', O.mat_T3),
                         ('m_T4', O.mat_T4)]:
            assert np.array_equal(fn(et), z[name][i]), name
        assert O.mag_lat(et) == z['mag_lat'][i]
        assert O.mag_lon(et) == z['mag_lon'][i]
    for (ra, dec, lp), rot in zip(z['wcs_in'], z['wcs_rot']):
        hdr = {'CRVAL1': ra, 'CRVAL2': dec, 'LONPOLE': lp}
        assert np.array_equal(O.wcs_rotation(hdr), rot)


def test_igrf_raises_past_2020():
    with pytest.raises(ValueError):
        O.mag_lon(O.date2es(datetime(2020, 1, 2)))


@pytest.mark.parametrize('name', SMALL)
def test_georef_small_bit_exact(name):
    z = load_golden(name)
    out = O.georef_frame(header_from(z), float(z['altitude']), z['cam'], z['m_geo'], z['m_sm'],
                         fast=name.endswith('fast.npz'))
    for k in ('dir_corner', 'p_corner', 'dir_center', 'p_center', 'lat', 'lon', 'lat_c', 'lon_c', 'elev',
              'mlat', 'mlt', 'mlat_c', 'mlt_c'):
        same(out[k], z[k])


def test_masks_small():
    z = load_golden('georef_small_iss030_fast.npz')
    g = load_golden('masks_small.npz')
    for tag in ('fast', 'exact'):
        zz = load_golden('georef_small_iss030_%s.npz' % tag)
        corner_nan = np.isnan(zz['lat'])
        center_nan = np.isnan(zz['lat_c'])
        if tag == 'fast':      # astrometry.py:35-40: born sanitised
            cm, ce = corner_nan, center_nan
        else:                  # mapping.py:1063-1125
            cm, ce = O.sanitize_masks(corner_nan, center_nan)
        assert np.array_equal(cm, g[tag + '_corner_mask'])
        assert np.array_equal(ce, g[tag + '_center_mask'])
        if tag == 'fast':
            # reference quirk: ImageMaskAstrometryMixin and ArrayImageMixin both use `self._img`
            # (astrometry.py:230-243 vs mapping.py:1016-1021), so an ArraySpacecraftMapping's image
            # is left unmasked in fast mode until the first createMasked(); the class invariant
            # (mapping.py:299-316) says it should carry the centre mask.
            assert not g['fast_img_mask'].any()
        else:
            assert np.array_equal(ce, g[tag + '_img_mask'])
        assert np.array_equal(ce, g[tag + '_elev_mask'])
        for e in (10, 25):
            elev = np.where(ce, np.nan, zz['elev'])
            cm2, ce2 = O.mask_by_elevation(elev, cm, e)
            assert np.array_equal(cm2, g['%s_e%d_corner_mask' % (tag, e)])
            assert np.array_equal(ce2, g['%s_e%d_center_mask' % (tag, e)])
            assert np.array_equal(ce2, g['%s_e%d_img_mask' % (tag, e)])
            assert np.array_equal(ce2, g['%s_e%d_elev_mask' % (tag, e)])
    with pytest.raises(ValueError):
        O.mask_by_elevation(z['elev'], np.isnan(z['lat']), 89.9)


RESAMPLE = ['resample_geo_%s_ppd%s.npz' % (p, r) for p in ('iss030', 'iss029') for r in ('10x10', '4x7')] + \
           ['resample_sm_iss030.npz', 'resample_sm_iss029.npz',
            'resample_synth_plain.npz', 'resample_synth_disc.npz', 'resample_synth_pole.npz']


@pytest.mark.parametrize('name', RESAMPLE)
def test_resample_mean(name):
    z = load_golden(name)
    if 'data' in z.files:
        data = z['data']
    else:
        cmask = np.isnan(z['lats_c'])
        img = z['img'].astype(np.float64)
        img[cmask] = np.nan
        data = np.dstack((img, z['elev']))
    res = O.resample_mean(z['lats_c'], z['lons_c'], float(z['altitude']), data, z['outline'], tuple(z['bbox']),
                          tuple(z['ppd']), bool(z['contains_discontinuity']), bool(z['contains_pole']))
    for k in ('lat', 'lon', 'lat_c', 'lon_c'):
        same(res[k], z['out_' + k])
    same(res['data'], z['out_data'])
    if 'out_img' in z.files:
        img, mask = O.finalize_image(res['data'][..., :-1], np.uint16)
        assert np.array_equal(mask, z['out_img_mask'])
        assert np.array_equal(img[~mask], z['out_img'][~mask])
    if 'geo_lat' in z.files:     # convertSMMappingToGeo (mapping.py:1549-1559)
        la, lo = O.sm_to_latlon(res['lat'], res['lon'], z['m_geo_sm'])
        same(la, z['geo_lat'])
        same(lo, z['geo_lon'])
        la, lo = O.sm_to_latlon(res['lat_c'], res['lon_c'], z['m_geo_sm'])
        same(la, z['geo_lat_c'])
        same(lo, z['geo_lon_c'])
    if name.startswith('resample_geo'):   # the stand-in bbox equals the one the fixture was made with
        cmask = np.isnan(z['corner_lat'])
        bb, disc = O.bbox_of_corners(z['corner_lat'], z['corner_lon'], cmask)
        assert np.array_equal(np.array(bb), z['bbox']) and disc == bool(z['contains_discontinuity'])


def test_histogram_edge_semantics():
    z = load_golden('histogram_edges.npz')
    for tag in 'abc':
        hs, xe, ye = O.histogram2d(z[tag + '_x'], z[tag + '_y'], bins=tuple(int(b) for b in z[tag + '_bins']),
                                   range=z[tag + '_range'].tolist(),
                                   weights=[None, z[tag + '_w1'], z[tag + '_w2']])
        assert np.array_equal(xe, z[tag + '_xedges']) and np.array_equal(ye, z[tag + '_yedges'])
        assert np.array_equal(hs[0], z[tag + '_count'])
        assert np.array_equal(hs[1], z[tag + '_s1'])
        assert np.array_equal(hs[2], z[tag + '_s2'])
    hs, xe, ye = O.histogram2d(z['d_x'], z['d_y'], bins=[z['d_xedges'], z['d_yedges']], weights=[None, z['d_w']])
    assert np.array_equal(hs[0], z['d_count']) and np.array_equal(hs[1], z['d_s'])
    # survey probe: edges [0,1,2,3]
    x = np.array([0.0, 1.0, 3.0, np.nextafter(3, 4), np.nextafter(0, -1)])
    h, _, _ = O.histogram2d(x, np.full(5, 0.5), bins=(3, 1), range=[[0, 3], [0, 1]])
    assert h[:, 0].tolist() == [1.0, 1.0, 2.0]


def test_reference_known_answers():
    with open(os.path.join(GOLDEN, 'known_answers.json')) as fp:
        ka = json.load(fp)
    for c in ka['ellipsoid']:
        res = O.ellipsoid_line_intersection(c['a'], c['b'], c['origin'], c['dirs'], directed=c['directed'])
        np.testing.assert_array_equal(res, np.array(c['expect'], dtype=float))
        np.testing.assert_array_equal(res, np.array(c['ref'], dtype=float))
        hit = O.ellipsoid_line_intersects(c['a'], c['b'], c['origin'], c['dirs'], directed=c['directed'])
        assert hit.tolist() == c['ref_intersects']
    w = ka['wgs84_chord']
    p1, p2 = np.array(w['p1']), np.array(w['p2'])
    i1 = O.ellipsoid_line_intersection(w['a'], w['b'], p1, [p1 - p2], directed=False)
    np.testing.assert_array_almost_equal(i1, [p1], w['decimals'])
    np.testing.assert_array_equal(i1, np.array(w['ref']))
    assert w['a'] == O.WGS84_A and w['b'] == O.WGS84_B
    for c in ka['sphere']:
        res = O.sphere_line_intersection(c['r'], c['origin'], np.asarray(c['dirs'], dtype=float), c['directed'])
        np.testing.assert_array_equal(res, np.array(c['expect'], dtype=float))
    s = ka['sscweb']
    et = O.date2es(datetime.strptime(s['date'], '%Y-%m-%dT%H:%M:%S'))
    dec = s['decimals']
    aae = np.testing.assert_array_almost_equal
    aae(O.rotate_vectors(O.mat_T1(et), s['gei']), s['geo'], dec)
    aae(O.rotate_vectors(O.mat_T2(et), s['gei']), s['gse'], dec)
    aae(O.rotate_vectors(O.mat_T3(et), s['gse']), s['gsm'], dec)
    aae(O.rotate_vectors(O.mat_T4(et), s['gsm']), s['sm'], dec)
    aae(O.rotate_vectors(O.mat_T1(et).T, s['geo']), s['gei'], dec)
    aae(O.rotate_vectors(O.mat_j2000_to_geo(et), s['j2000']), s['geo'], dec)
    aae(O.rotate_vectors(O.mat_j2000_to_sm(et), s['j2000']), s['sm'], dec)
    aae(O.rotate_vectors(O.mat_geo_to_sm(et), s['geo']), s['sm'], dec)
    g = ka['geodetic_roundtrip']
    (a0, a1, astep), (b0, b1, bstep) = g['mgrid']
    lat, lon = np.mgrid[a0:a1:astep, b0:b1:bstep]
    x, y, z = O.geodetic_to_ecef_zero(np.deg2rad(lat), np.deg2rad(lon))
    la, lo = O.ecef_to_geodetic(x.ravel(), y.ravel(), z.ravel())
    aae(np.rad2deg(la).reshape(lat.shape), lat, g['decimals'])
    aae(np.rad2deg(lo).reshape(lon.shape), lon, g['decimals'])
    la_s = np.linspace(*g['lat_linspace'][:2], num=g['lat_linspace'][2])
    lo_s = np.linspace(*g['lon_linspace'][:2], num=g['lon_linspace'][2])
    lat, lon = np.meshgrid(la_s, lo_s)
    x, y, z = O.geodetic_to_ecef_zero(np.deg2rad(lat), np.deg2rad(lon))
    la, lo = O.ecef_to_geodetic(x.ravel(), y.ravel(), z.ravel())
    aae(np.rad2deg(la).reshape(lat.shape), lat, g['decimals'])
    aae(np.rad2deg(lo).reshape(lon.shape), lon, g['decimals'])


# ---- other camera models: all-sky fisheye, THEMIS altitude reprojection ----------------
def _cal(z):
    return {k: float(z['cal_' + k]) for k in ('lat', 'lon', 'xc', 'yc', 'k', 'rotation')}


@pytest.mark.parametrize('name', ['miracle_sod64.npz', 'miracle_kev96.npz'])
def test_allsky_oracle_vs_reference(name):
    """MIRACLEMapping of the real reference, with the centre offset as intended (+0.5) and as its pinned NumPy 1.6
    evaluates it (+0): az/el tables, GEO directions, geodetic coordinates."""
    z = load_golden(name)
    for prefix, off in (('', 0.5), ('np16_', 0.0)):
        g = O.allsky_georef(int(z['size']), _cal(z), float(z['altitude']), center_offset=off)
        for k in ('az', 'el_corner', 'az_c', 'elev'):
            same(g[k], z[prefix + k], 1e-12)
        for k in ('dirs', 'dirs_c'):
            same(g[k], z[prefix + k], 1e-15)
        for k in ('lat', 'lon', 'lat_c', 'lon_c'):
            same(g[k], z[prefix + k], 1e-12)
    # the rays of a ground camera always leave the shell: nothing is missing before the elevation mask
    assert not np.isnan(z['lat']).any()
    corner_mask, center_mask = O.mask_by_elevation(z['elev'], np.isnan(z['lat']), 0.1)
    assert np.array_equal(corner_mask, z['corner_mask'])
    assert np.array_equal(center_mask, z['center_mask'])


def test_allsky_oracle_native_size_samples():
    z = load_golden('miracle_sod512.npz')
    g = O.allsky_georef(512, _cal(z), 110.0)
    step = int(z['step'])
    for k in ('az', 'el_corner', 'az_c', 'elev', 'lat', 'lon', 'lat_c', 'lon_c'):
        same(g[k][::step, ::step], z[k], 1e-12)
        d = z['digest_' + k]
        a = g[k]
        assert a.size == d[0]
        assert abs(a.sum() - d[1]) <= 1e-9 * max(1.0, abs(d[1]))
        assert abs(a.min() - d[2]) <= 1e-12 and abs(a.max() - d[3]) <= 1e-12


def test_themis_reproject_oracle_vs_reference():
    z = load_golden('themis_reproject.npz')
    for h in (90, 150):
        la, lo = O.themis_reproject(tuple(z['station']), z['lat_ref'], z['lon_ref'], float(z['height_ref']), h)
        same(la, z['lat_%d' % h], 1e-12)
        same(lo, z['lon_%d' % h], 1e-12)
    # the reference height itself is nearly a fixed point (the shell a+h, b+h is not exactly the surface of
    # constant geodetic height h, hence micro-degrees and not rounding errors)
    la, lo = O.themis_reproject(tuple(z['station']), z['lat_ref'], z['lon_ref'], 110.0, 110.0)
    same(la, z['lat_ref'], 1e-5)
    same(lo, z['lon_ref'], 1e-5)


# ---- traced outline, polygon area / centroid (utils.py:97-225; outline_test.py as data) ------------------------
def outline_test_image(spec):
    n, r = int(spec['n']), float(spec['radius'])
    y, x = np.ogrid[-r: r + 1, -r: r + 1]
    im = np.zeros((n, n), bool)
    disc = x ** 2 + y ** 2 <= r ** 2
    im[:disc.shape[0], :disc.shape[1]] = disc
    im[tuple(spec['removed'])] = False
    return im


def test_outline_oracle_vs_reference_known_answers():
    """outline_test.py:109-149: the literal outline polygon of `_testIm(10)` (order and starting point included), its
    area, and the polygon centroid vector."""
    with open(os.path.join(GOLDEN, 'known_answers.json')) as fp:
        ka = json.load(fp)['outline']
    assert O.outline(outline_test_image(ka['test_image'])).tolist() == ka['polygon']
    assert O.polygon_area(ka['polygon']) == ka['area']
    assert O.polygon_area(ka['polygon'], signed=True) == ka['area']        # clockwise in image coordinates
    got = O.polygon_centroid(ka['centroid_polygon'])
    np.testing.assert_almost_equal(got, ka['centroid'], decimal=ka['centroid_decimals'])
    assert list(got) == ka['centroid_ref']                                  # the real function, bit for bit
    # two blobs: the bigger one is returned; a hole does not change the outer outline
    im = outline_test_image(ka['test_image'])
    two = np.hstack((im, np.zeros((10, 3), bool), im[:, :6]))
    assert O.outline(two).tolist() == ka['polygon']
    holed = im.copy()
    holed[4, 4] = False
    assert O.outline(holed).tolist() == ka['polygon']


# ---- resample(method='nearest') (resample.py:246-259,323-327) --------------------------------------------------
NEAREST = ['resample_nearest_iss030.npz', 'resample_nearest_iss029.npz', 'resample_nearest_synth_plain.npz',
           'resample_nearest_synth_disc.npz', 'resample_nearest_synth_pole.npz']


@pytest.mark.parametrize('name', NEAREST)
def test_resample_nearest_oracle_vs_reference(name):
    z = load_golden(name)
    if 'img' in z.files:
        data = np.dstack((z['img'].astype(np.float64), z['elev']))
        data[np.isnan(z['lats_c'])] = np.nan
    else:
        data = z['data']
    res = O.resample_nearest(z['lats_c'], z['lons_c'], float(z['altitude']), data, z['outline'], tuple(z['bbox']),
                             tuple(z['ppd']), bool(z['contains_discontinuity']), bool(z['contains_pole']))
    for k in ('lat', 'lon', 'lat_c', 'lon_c'):
        same(res[k], z['out_' + k], 0.0 if not bool(z['contains_pole']) else 1e-12)
    same(res['data'], z['out_data'])
    assert not np.isnan(res['data'][..., 0]).all()


def test_clough_tocher_restatement_equals_scipy():
    """griddata(method='cubic') (reference resample.py:323-326) is scipy's CloughTocher2DInterpolator: the restatement of
    its gradient estimator and of its element (oracle/ref_numpy.py), on scipy's own triangulation, against scipy."""
    from scipy.interpolate import CloughTocher2DInterpolator
    from scipy.spatial import Delaunay
    rng = np.random.RandomState(3)
    pts = rng.rand(70, 2)
    vals = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.1 * rng.rand(70)
    tri = Delaunay(pts)
    ref = CloughTocher2DInterpolator(tri, vals)
    indptr, indices = tri.vertex_neighbor_vertices
    grad, sweeps = O.clough_tocher_gradients(pts, indptr, indices, vals)
    assert 2 <= sweeps < 400
    assert np.max(np.abs(grad - ref.grad[:, 0, :])) < 1e-13
    q = rng.rand(150, 2)
    want = ref(q)
    simplex = tri.find_simplex(q)
    assert np.array_equal(simplex < 0, np.isnan(want))
    for p, s, w in zip(q, simplex, want):
        if s < 0:
            continue
        v = tri.simplices[s]
        T = tri.transform[s]
        c = T[:2].dot(p - T[2])
        b = (c[0], c[1], 1 - c.sum())
        cent = [None if n < 0 else pts[tri.simplices[n]].mean(axis=0) for n in tri.neighbors[s]]
        got = O.clough_tocher_value(pts[v], b, vals[v], grad[v], cent)
        assert abs(got - w) < 1e-13
