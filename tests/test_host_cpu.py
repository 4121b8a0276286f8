"""
CPU-only checks of the product's host side: the C ABI library builds/loads and exports every
symbol of include/auromat_hip.h; per-frame host scalars equal the reference's (golden fixtures);
grid layout logic; no compute entry point works without a GPU (there is no CPU fallback).
"""
import json
import os
import re
from datetime import datetime

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden


def parse(s):
    return datetime.strptime(str(s), '%Y-%m-%dT%H:%M:%S.%f')


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from auromat_amd import _native
    return _native.lib()


def test_every_declared_symbol_is_exported(lib):
    from auromat_amd import _native
    with open(os.path.join(ROOT, 'include', 'auromat_hip.h')) as fp:
        text = fp.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    declared = set(re.findall(r'\b(amt_[a-z0-9_]+)\s*\(', text))
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(lib, name), 'library does not export ' + name
    assert declared == set(_native.exported_symbols()), declared ^ set(_native.exported_symbols())
    assert lib.amt_abi_version() == _native.ABI_VERSION == int(re.search(r'#define AMT_ABI_VERSION (\d+)', text).group(1))


def test_struct_layouts_match_header():
    import ctypes as C
    from auromat_amd._native import Axis, FrameParams, GeorefOut, Grid, PipeResult
    assert C.sizeof(FrameParams) == 16 + 8 * (4 + 2 + 9 + 3 + 4 + 9 + 9)
    assert C.sizeof(GeorefOut) == 8 * 11 + 8 * 4 + 4 * 4 + 8 * 3 + 4 * 2 + 8
    assert C.sizeof(Axis) == 8 + 8 + 8 * 5
    assert C.sizeof(Grid) == 16 + 8 * 10 + 2 * C.sizeof(Axis)
    assert C.sizeof(PipeResult) == 16 + 8 * 8 + C.sizeof(Grid)


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from auromat_amd._native import NativeError
    from auromat_amd.coordinates.intersection import ellipsoidLineIntersection
    from auromat_amd.mapping.spacecraft import ArraySpacecraftMapping
    from auromat_amd.synthetic import frame_header, frame_image
    with pytest.raises(NativeError):
        ellipsoidLineIntersection(2, 2, [0, 3, 0], [[0, -1, 0]])
    hdr, cam, t = frame_header(32, 24)
    m = ArraySpacecraftMapping(hdr, 110, frame_image(32, 24), cam, t, 'x', fastCenterCalculation=True)
    with pytest.raises(NativeError):
        m.lats


def test_host_scalars_equal_reference():
    from auromat_amd.coordinates import transform as T
    from auromat_amd.coordinates.wcs import celestial_rotation
    z = load_golden('host_scalars.npz')
    for i, d in enumerate(z['dates']):
        et = T.date2es(parse(d))
        assert et == z['et'][i]
        for name, fn in [('m_geo', T.mat_j2000_to_geo), ('m_sm', T.mat_j2000_to_sm), ('m_geo_sm', T.mat_geo_to_sm),
                         ('m_P', T.mat_P), ('m_T1', T.mat_T1), ('m_T2', T.mat_T2), ('m_T3', T.mat_T3),
                         ('m_T4', T.mat_T4)]:
            assert np.array_equal(fn(et), z[name][i]), name
        assert T.mag_lat(et) == z['mag_lat'][i] and T.mag_lon(et) == z['mag_lon'][i]
    for (ra, dec, lp), rot in zip(z['wcs_in'], z['wcs_rot']):
        assert np.array_equal(celestial_rotation({'CRVAL1': ra, 'CRVAL2': dec, 'LONPOLE': lp}), rot)
    with pytest.raises(ValueError):
        T.mag_lat(T.date2es(datetime(2020, 1, 2)))          # igrf.py:55-58
    loc = T.northGeomagneticPoleLocation(datetime(2012, 1, 25))
    assert 79 < loc.lat < 81 and -73 < loc.lon < -71


def test_grid_layout_equals_reference():
    from auromat_amd.resample import _Grid, fixedGrid
    for name in ('resample_geo_iss030_ppd10x10.npz', 'resample_geo_iss029_ppd4x7.npz', 'resample_synth_plain.npz'):
        z = load_golden(name)
        lat_s, lon_w, lat_n, lon_e = z['bbox']
        g = _Grid(tuple(z['ppd']), lat_s, lat_n, lon_w, lon_e)
        assert np.array_equal(g.lat, z['out_lat']) and np.array_equal(g.lon, z['out_lon'])
        assert np.array_equal(g.lat_c, z['out_lat_c']) and np.array_equal(g.lon_c, z['out_lon_c'])
        assert (g.ny, g.nx) == z['out_data'].shape[:2]
    # survey probe: lat 60..70, lon 160..170 at 1 px/deg -> 9x9 cells centred 61..69 / 161..169
    g = _Grid((1, 1), 60, 70, 160, 170)
    assert g.latCenters.tolist() == list(range(69, 60, -1)) and g.lonCenters.tolist() == list(range(161, 170))
    assert fixedGrid((1, 1), 60, 70, 160, 170) == (11, 11, 60.0, 70.0, 160.0, 170.0)


def test_native_grid_layout_equals_python_layout(lib):
    """amt_grid_layout (C++, used by the single-pass driver) == _Grid / make_axis (NumPy, pinned to the reference)."""
    import ctypes as C
    from auromat_amd._native import Grid
    from auromat_amd.resample import _Grid
    from auromat_amd.util.histogram import make_axis
    rng = np.random.RandomState(7)
    g = Grid()
    n_checked = 0
    for ppd in [(10, 10), (4, 7), (1, 1), (20, 20), (3, 5), (10, 20), (2.5, 2.5)]:
        for _ in range(300):
            lat_c, lon_c = rng.uniform(-80, 80), rng.uniform(-150, 150)
            dlat, dlon = rng.uniform(0.7, 9), rng.uniform(0.7, 25)
            box = (lat_c - dlat, lat_c + dlat, lon_c - dlon, lon_c + dlon)
            if rng.rand() < 0.2:                 # boxes that sit exactly on grid nodes
                box = tuple(np.round(np.array(box) * ppd[0]) / ppd[0])
            rc = lib.amt_grid_layout(ppd[0], ppd[1], box[0], box[1], box[2], box[3], C.byref(g))
            py = _Grid(ppd, *box)
            assert rc == 0
            assert (g.nx, g.ny) == (py.nx, py.ny), (ppd, box)
            assert (g.lat_step, g.lon_step) == (py.latStep, py.lonStep)
            assert (g.lat_center_first, g.lat_center_last) == (py.latCenters[0], py.latCenters[-1])
            assert (g.lon_center_first, g.lon_center_last) == (py.lonCenters[0], py.lonCenters[-1])
            for ax, edges in ((g.xaxis, py.xedges), (g.yaxis, py.yedges)):
                ref, _ = make_axis(None, edges, uniform=True)
                assert ref.uniform == 1 and ax.uniform == 1 and not ax.edges
                for k in ('nbin', 'first', 'last', 'step', 'scale', 'last_rounded'):
                    assert getattr(ax, k) == getattr(ref, k), (k, ppd, box)
            n_checked += 1
    assert n_checked == 2100
    # a box inside one cell gives no output cell: the reference asserts (resample.py:225-226), the C side says EINVAL
    assert lib.amt_grid_layout(1.0, 1.0, 10.2, 10.3, 20.2, 20.3, C.byref(g)) != 0


def test_bounding_box_logic():
    from auromat_amd.mapping.mapping import BoundingBox, bounding_box_from_reduction, wrap_at_180
    bb = bounding_box_from_reduction([10, 20, -30, 40, 5, -2, 100, 0])
    assert (bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast) == (10, -30, 20, 40)
    assert not bb.containsDiscontinuity and not bb.containsPole
    bb = bounding_box_from_reduction([10, 20, -179, 178, 170, -175, 100, 0])
    assert (bb.lonWest, bb.lonEast) == (170, -175) and bb.containsDiscontinuity and not bb.containsPole
    bb = bounding_box_from_reduction([80, 89.9, -179, 178, 1, -1, 100, 3])
    assert bb.containsPole and bb.latNorth == 90 and bb.latSouth == 80
    bb = bounding_box_from_reduction([-89, -80, -179, 178, 1, -1, 100, 1])
    assert bb.containsPole and bb.latSouth == -90 and bb.latNorth == -80
    with pytest.raises(ValueError):
        bounding_box_from_reduction([np.inf, -np.inf, np.inf, -np.inf, np.inf, -np.inf, 0, 0])
    assert wrap_at_180(190.0) == -170.0 and wrap_at_180(-180.0) == -180.0 and wrap_at_180(180.0) == -180.0
    m = BoundingBox.mergedBoundingBoxes([BoundingBox(0, 170, 10, 175), BoundingBox(-5, -178, 5, -170)])
    assert (m.latSouth, m.lonWest, m.latNorth, m.lonEast) == (-5, 170, 10, -170)
    assert BoundingBox(0, 1, 2, 3) == BoundingBox(0, 1, 2, 3) and BoundingBox(0, 1, 2, 3) != BoundingBox(0, 1, 2, 4)


def test_header_helpers_and_synthetic_frames():
    from auromat_amd.mapping.spacecraft import getPhotoTime, getShiftedSpacecraftPosition, getSpacecraftPosition
    from auromat_amd.synthetic import frame_header, sequence_frame
    hdr = {'DATE-OBS': '2012-01-25T09:27:08.060000', 'POSX': 1.0, 'POSY': 2.0, 'POSZ': 3.0,
           'POSXSHIF': 4.0, 'POSYSHIF': 5.0, 'POSZSHIF': 6.0, 'DATESHIF': -13.0}
    assert getPhotoTime(hdr) == datetime(2012, 1, 25, 9, 27, 8, 60000)
    xyz, date, delta = getShiftedSpacecraftPosition(hdr)
    assert xyz.tolist() == [4.0, 5.0, 6.0] and date == datetime(2012, 1, 25, 9, 26, 55, 60000)
    assert getSpacecraftPosition(hdr)[0].tolist() == [1.0, 2.0, 3.0]
    assert getShiftedSpacecraftPosition({'DATE-OBS': '2011-09-18T11:54:56'}) == (None, None, None)
    h0, cam0, t0 = frame_header(4240, 2832)
    h5, cam5, t5, seed = sequence_frame(5)
    assert abs(np.linalg.norm(cam5) - np.linalg.norm(cam0)) < 1e-9 and (t5 - t0).total_seconds() == 5
    assert abs(np.linalg.norm(cam5 - cam0) - 5 * 7.66) < 0.01 and h5['CRVAL1'] == h0['CRVAL1'] + 0.25


def test_angular_distance_on_parallel_against_geodesic_integration():
    """
    plateCarreeResolution's geodesic arc (geographiclib's a12 in the reference, resample.py:36-61) against an
    independent computation: the geodesic equations on the WGS84 ellipsoid integrated numerically and shot from
    one end point to the other (Clairaut's relation then gives the arc on the auxiliary sphere).
    """
    import math
    from scipy.integrate import solve_ivp
    from scipy.optimize import brentq
    from auromat_amd.coordinates.geodesic import WGS84_a_m, WGS84_f, angularDistanceOnParallel
    from auromat_amd.mapping.mapping import BoundingBox
    from auromat_amd.resample import plateCarreeResolution
    f, a = WGS84_f, WGS84_a_m
    e2 = f * (2 - f)

    def shoot(lat, dlon):
        phi1, target = math.radians(lat), math.radians(dlon)

        def rhs(s, y):
            phi, lam, al = y
            w = math.sqrt(1 - e2 * math.sin(phi) ** 2)
            return [math.cos(al) * w ** 3 / (a * (1 - e2)), math.sin(al) * w / (a * math.cos(phi)),
                    math.sin(al) * math.tan(phi) * w / a]

        def back_on_parallel(s, y):
            return y[0] - phi1
        back_on_parallel.terminal = True
        back_on_parallel.direction = -1 if lat >= 0 else 1

        def miss(alpha1):
            sol = solve_ivp(rhs, [0, 3e7], [phi1, 0.0, alpha1], events=back_on_parallel, rtol=1e-12, atol=1e-12,
                            max_step=2e5)
            return sol.y_events[0][0][1] - target
        a1 = brentq(miss, 1e-6, math.pi / 2 - 1e-9, xtol=1e-14) if lat >= 0 else \
            brentq(miss, math.pi / 2 + 1e-9, math.pi - 1e-6, xtol=1e-14)
        beta = math.atan((1 - f) * math.tan(phi1))
        s1 = math.atan2(math.sin(beta), math.cos(a1) * math.cos(beta))
        return math.degrees(abs((math.pi if lat >= 0 else -math.pi) - 2 * s1))      # symmetric: sigma2 = +-pi - sigma1

    for lat, dlon in ((55.0, 25.0), (-57.5, 25.5), (80.0, 120.0), (30.0, 1.0), (10.0, 40.0)):
        assert abs(angularDistanceOnParallel(lat, dlon) / shoot(lat, dlon) - 1) < 1e-9, (lat, dlon)
    # equator, symmetry, the sphere limit as a sanity bound (< 0.3 % apart), zero
    assert angularDistanceOnParallel(0.0, 10.0) == 10.0 / (1 - f)
    assert angularDistanceOnParallel(-40.0, 33.0) == angularDistanceOnParallel(40.0, -33.0)
    assert angularDistanceOnParallel(12.0, 0.0) == 0.0
    beta = math.atan((1 - f) * math.tan(math.radians(51.0)))
    sphere = math.degrees(2 * math.asin(math.cos(beta) * math.sin(math.radians(20.6) / 2)))
    assert abs(angularDistanceOnParallel(51.0, 20.6) / sphere - 1) < 3e-3
    # plateCarreeResolution: latitude part exact, longitude part from the arc; across the dateline the same
    lat_ppd, lon_ppd = plateCarreeResolution(BoundingBox(47.9, -102.2, 54.2, -91.9), 100)
    assert lat_ppd == 36.0
    assert abs(lon_ppd - angularDistanceOnParallel(51.05, 10.3) / (100 / 3600.0) / (-91.9 + 102.2)) < 1e-9
    assert plateCarreeResolution(BoundingBox(-5, 170, 5, -170), 200) == \
        plateCarreeResolution(BoundingBox(-5, -10, 5, 10), 200)


def test_native_plate_carree_resolution_equals_the_python_restatement(lib):
    """amt_plate_carree_resolution (csrc/amt_grid.h; what resample(arcsecPerPx=...), the sequence pipeline's box-first plan
    and the native runner call) against the Python restatement of reference resample.py:36-61 on random boxes: the latitude
    part equal, the longitude part to 1e-12 for boxes of a camera frame's width (the arc of a box a fraction of a degree wide
    is ill-conditioned in both: 1e-9), and the number of global grid nodes, which is all the grid layout takes from it, the
    same; boxes across the date line; a box that goes all around (a pole in view) has no longitude resolution in either."""
    import random
    from auromat_amd.mapping.mapping import BoundingBox
    from auromat_amd.resample import plateCarreeResolution, plateCarreeResolution_py
    rnd = random.Random(5)
    for i in range(400):
        ls = rnd.uniform(-89, 85)
        ln = ls + rnd.uniform(0.01, min(40, 89.9 - ls))
        lw = rnd.uniform(-180, 180)
        width = rnd.uniform(0.05, 170) if i % 4 else rnd.uniform(5, 80)
        le = lw + width
        if le > 180:
            le -= 360
        arc = rnd.choice([100, 50, 200, 360, 13.7])
        box = BoundingBox(ls, lw, ln, le)
        got, want = plateCarreeResolution(box, arc), plateCarreeResolution_py(box, arc)
        assert got[0] == want[0] == 3600.0 / arc
        assert abs(got[1] - want[1]) <= (1e-12 if width > 5 else 1e-9) * want[1], (ls, lw, ln, le)
        assert round(got[1] * 360 + 1) == round(want[1] * 360 + 1)
    assert plateCarreeResolution(BoundingBox(-5, 170, 5, -170), 200) == plateCarreeResolution(BoundingBox(-5, -10, 5, 10), 200)
    assert plateCarreeResolution(BoundingBox(60, -180, 90, 180), 100) == plateCarreeResolution_py(BoundingBox(60, -180, 90, 180), 100) == (36.0, 0.0)
    # the C ABI's error behaviour: a distinct status for the all-round box (a host that checks it cannot lay out a grid without
    # columns), the reference's values in the outputs; AMT_EINVAL for a resolution that is not positive and for NULL outputs
    import ctypes as C
    la, lo = C.c_double(-1), C.c_double(-1)
    assert lib.amt_plate_carree_resolution(60.0, -180.0, 90.0, 180.0, 100.0, C.byref(la), C.byref(lo)) == -5      # AMT_EDOMAIN
    assert (la.value, lo.value) == (36.0, 0.0)
    assert lib.amt_plate_carree_resolution(40.0, -10.0, 50.0, 10.0, 100.0, C.byref(la), C.byref(lo)) == 0 and lo.value > 0
    assert lib.amt_plate_carree_resolution(40.0, -10.0, 50.0, 10.0, 0.0, C.byref(la), C.byref(lo)) == -1
    assert lib.amt_plate_carree_resolution(40.0, -10.0, 50.0, 10.0, 100.0, None, C.byref(lo)) == -1


def test_public_header_is_plain_c():
    """include/auromat_hip.h is the C ABI: it must compile as C99 (and C++) without a HIP toolchain."""
    import subprocess
    import tempfile
    inc = os.path.join(ROOT, 'include')
    with tempfile.TemporaryDirectory() as tmp:
        for name, cc, std in (('t.c', 'gcc', '-std=c99'), ('t.cpp', 'g++', '-std=c++11')):
            src = os.path.join(tmp, name)
            with open(src, 'w') as fp:
                fp.write('#include "auromat_hip.h"\nint main(void) { amt_pipe_result r; amt_georef_out o; '
                         'return (int)(sizeof(r) + sizeof(o)) == 0; }\n')
            res = subprocess.run([cc, std, '-pedantic', '-Wall', '-Wextra', '-Werror', '-I', inc, '-c', src, '-o',
                                  src + '.o'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
            assert res.returncode == 0, res.stdout


# ---- geodesic helpers and BoundingBox centre / size against the reference's own known answers --------------------
def _known():
    import json
    with open(os.path.join(GOLDEN, 'known_answers.json')) as fp:
        return json.load(fp)


def test_contains_or_crosses_pole_reference_vectors():
    """geodesic_test.py:14-29 literal polygons, and the large outline polygons of testPoleBug / testPoleBug2
    (geodesic_test.py:41-2644, data in tests/golden/geodesic_polygons.npz): none of those contains a pole."""
    from auromat_amd.coordinates.geodesic import containsOrCrossesPole
    from auromat_amd.utils import convexHull
    for case in _known()['contains_pole']:
        assert containsOrCrossesPole(case['poly']) == case['expect'], case
    z = load_golden('geodesic_polygons.npz')
    r100 = z['testPoleBug_outlineReduced100']
    assert not containsOrCrossesPole(convexHull(r100[::2]))
    assert not containsOrCrossesPole(convexHull(r100))
    assert not containsOrCrossesPole(z['testPoleBug_outlineFull'])
    assert not containsOrCrossesPole(z['testPoleBug2_outlineFull'])
    assert not containsOrCrossesPole(z['testPoleBug2_outlineHull'])
    assert not containsOrCrossesPole(z['testPoleBug2_outlineHullReduced'])
    # the same polygons rotated onto a pole do contain it
    assert containsOrCrossesPole([[80, lon] for lon in range(-180, 180, 30)])
    assert containsOrCrossesPole([[-80, lon] for lon in range(170, -190, -30)])


def test_bounding_box_center_and_size_reference_vectors():
    """boundingbox_test.py:12-50 to the 6 decimals its assert_array_almost_equal asks for (km and degrees)."""
    from auromat_amd.mapping.mapping import BoundingBox
    ka = _known()
    for case in ka['bounding_box']:
        s, w, n, e = case['box']
        bb = BoundingBox(latSouth=s, lonWest=w, latNorth=n, lonEast=e)
        np.testing.assert_array_almost_equal(bb.center, case['center'])
        np.testing.assert_array_almost_equal(bb.size, case['size'])
    m = ka['bounding_box_merge']
    boxes = [BoundingBox(latSouth=b[0], lonWest=b[1], latNorth=b[2], lonEast=b[3]) for b in m['boxes']]
    bb = BoundingBox.mergedBoundingBoxes(boxes)
    assert [bb.latSouth, bb.lonWest, bb.latNorth, bb.lonEast] == m['merged']
    np.testing.assert_array_almost_equal(bb.center, m['center'])


def test_angular_distance_on_parallel_against_vincenty():
    """A second independent route to geographiclib's a12 (absent offline): Vincenty's inverse formulae (Survey Review 23, 1975 —
    another published algorithm, a series to the third order in the flattening iterated on the longitude on the auxiliary
    sphere), whose sigma IS the arc on the auxiliary sphere.  Good to ~1e-10 relative away from the antipode."""
    import math
    from auromat_amd.coordinates.geodesic import WGS84_f, angularDistanceOnParallel
    f = WGS84_f

    def vincenty_sigma(lat1, lat2, dlon):
        u1, u2 = math.atan((1 - f) * math.tan(math.radians(lat1))), math.atan((1 - f) * math.tan(math.radians(lat2)))
        L = math.radians(dlon)
        lam = L
        for _ in range(200):
            sin_s = math.hypot(math.cos(u2) * math.sin(lam), math.cos(u1) * math.sin(u2) - math.sin(u1) * math.cos(u2) * math.cos(lam))
            cos_s = math.sin(u1) * math.sin(u2) + math.cos(u1) * math.cos(u2) * math.cos(lam)
            sigma = math.atan2(sin_s, cos_s)
            sin_a = math.cos(u1) * math.cos(u2) * math.sin(lam) / sin_s
            cos2_a = 1 - sin_a ** 2
            cos_2sm = cos_s - 2 * math.sin(u1) * math.sin(u2) / cos2_a if cos2_a > 0 else 0.0
            c = f / 16 * cos2_a * (4 + f * (4 - 3 * cos2_a))
            new = L + (1 - c) * f * sin_a * (sigma + c * sin_s * (cos_2sm + c * cos_s * (-1 + 2 * cos_2sm ** 2)))
            if abs(new - lam) < 1e-15:
                break
            lam = new
        return math.degrees(sigma)

    for lat, dlon in ((55.0, 25.0), (-57.5, 25.5), (80.0, 120.0), (30.0, 1.0), (10.0, 40.0), (51.05, 10.3), (-75.0, 90.0), (5.0, 0.02),
                      (0.0, 17.0)):
        got, want = angularDistanceOnParallel(lat, dlon), vincenty_sigma(lat, lat, dlon)
        # (2e-10 relative; a few 1e-11 deg absolute where the arc itself is a fraction of a degree and both routes are at the
        # resolution of their own arithmetic)
        assert abs(got - want) < 2e-10 * want + 3e-11, (lat, dlon, got, want)


def test_geodesic_direct_inverse_consistency():
    """destination / intermediate / line invert distance / course; the parallel arc of plateCarreeResolution equals
    the general inverse solution."""
    from auromat_amd.coordinates import geodesic as G
    rng = np.random.RandomState(3)
    for _ in range(200):
        p = G.Location(rng.uniform(-89, 89), rng.uniform(-180, 180))
        azi, dist = rng.uniform(-180, 180), rng.uniform(1.0, 9.0e6)
        q = G.destination(p, azi, dist)
        assert abs(G.distance(p, q) - dist) < 1e-3                          # < 1 mm over up to 9000 km
        assert abs((G.course(p, q) - azi + 180) % 360 - 180) < 1e-7
        mid = G.intermediate(p, q, 0.5)
        assert abs(G.distance(p, mid) - dist / 2) < 1e-3 and abs(G.distance(mid, q) - dist / 2) < 1e-3
    for lat, dlon in [(0.0, 10.0), (35.0, 60.0), (-62.0, 25.0), (80.0, 120.0)]:
        a = G.angularDistance(G.Location(lat, -20.0), G.Location(lat, -20.0 + dlon))
        assert abs(a - G.angularDistanceOnParallel(lat, dlon)) < 1e-9
    pts = G.line(G.Location(10, 20), G.Location(12, 25), resolution=10000)
    assert len(pts) == int(G.distance(G.Location(10, 20), G.Location(12, 25)) // 10000)
    np.testing.assert_allclose(pts[0], [10, 20], atol=1e-12)
    np.testing.assert_allclose(pts[-1], [12, 25], atol=1e-8)
    assert G.line(G.Location(10, 20), G.Location(10.001, 20), resolution=1000).shape == (2, 2)
    assert G.distance(G.Location(5, 5), G.Location(5, 5)) == 0


def test_contour_links_are_ranked_like_a_walk():
    """utils.contours_from_links (pointer doubling over the device's contour links) against a plain walk of random
    link sets made of several cycles, shuffled records, with pixels that repeat along a contour."""
    from auromat_amd.utils import contours_from_links
    rng = np.random.RandomState(0)
    for trial in range(40):
        n = rng.randint(1, 400)
        perm = rng.permutation(n)
        cuts = sorted(set(rng.randint(1, n + 1, size=rng.randint(1, 6)).tolist() + [n]))
        keys = np.sort(rng.choice(4 * n + 8, n, replace=False))         # dense keys: neighbouring keys share pixels
        nxt = np.empty(n, dtype=np.int64)
        cycles, a = [], 0
        for b in cuts:
            cyc = perm[a:b]
            a = b
            if len(cyc):
                cycles.append(cyc)
                nxt[cyc] = np.roll(cyc, -1)
        links = np.transpose([keys, keys[nxt]])[rng.permutation(n)]
        got = contours_from_links(links)
        want = []
        for cyc in cycles:
            c = np.roll(cyc, -int(np.argmin(keys[cyc])))                 # contours start at their smallest key
            px = keys[c] // 4
            keep = np.ones(len(px), bool)
            keep[1:] = px[1:] != px[:-1]
            px = px[keep]
            want.append(px[:-1] if len(px) > 1 and px[0] == px[-1] else px)
        key = lambda c: (len(c), c.tolist())
        assert [c.tolist() for c in sorted(got, key=key)] == [c.tolist() for c in sorted(want, key=key)], trial
    assert contours_from_links(np.zeros((0, 2), np.int64)) == []


def test_getMapping_has_the_references_positional_order():
    """reference spacecraft.py:380-382: a 4th positional argument is noradId, not the altitude."""
    import inspect
    from auromat_amd.mapping.spacecraft import getMapping, getMappingSequence
    names = list(inspect.signature(getMapping).parameters)
    assert names[:11] == ['imagePathOrArray', 'wcsPathOrHeader', 'timeshift', 'noradId', 'tleFolder', 'spacetrack',
                          'altitude', 'fastCenterCalculation', 'metadata', 'nosanitize', 'identifier']
    sig = inspect.signature(getMapping)
    assert sig.parameters['altitude'].default == 110 and sig.parameters['fastCenterCalculation'].default is False
    hdr = {'DATE-OBS': '2012-01-25T09:26:55.060', 'POSX': 1.0, 'POSY': 2.0, 'POSZ': 3.0}
    img = np.zeros((2, 2, 3), np.uint8)
    for bad in (dict(noradId=25544), dict(tleFolder='/tmp'), dict(spacetrack=object())):
        with pytest.raises(NotImplementedError):
            getMapping(img, hdr, **bad)
    with pytest.raises(NotImplementedError):
        getMapping(img, hdr, None, 25544)          # positionally, as a caller of the reference would
    with pytest.raises((IOError, OSError)):
        getMapping('no-such-frame.jpg', hdr)       # paths are read (auromat_amd.util.image.loadImage)
    assert list(inspect.signature(getMappingSequence).parameters) == ['imagePathsOrArrays', 'wcsPaths', 'metadatas', 'timeshift',
                                                                       'noradId', 'tleFolder', 'spacetrack', 'altitude', 'parallel',
                                                                       'fastCenterCalculation']      # spacecraft.py:308-310


def test_seq_unpack_and_payload_size_are_host_functions():
    """amt_seq_payload_size / amt_seq_unpack (the C side of the gather's wire format) run without a GPU and read what
    auromat_amd.sequence.pack_results writes."""
    import ctypes as C
    import torch
    from auromat_amd._native import SeqFrame, lib
    from auromat_amd.sequence import DESC_LEN, pack_results
    from auromat_amd.resample import _Grid
    assert C.sizeof(SeqFrame) == 16 + 32 + 16 + 8 + 16
    L = lib()
    res = []
    for k in (2, 5, 7):
        grid = _Grid((4, 5), 40.0 + 0.3 * k, 43.0 + 0.37 * k, -100.0 + k, -96.5 + 1.2 * k)
        rs = np.random.RandomState(k)
        res.append(dict(mean=torch.from_numpy(rs.uniform(0, 9, (grid.ny, grid.nx, 4))), count=torch.from_numpy(
            rs.randint(0, 50, (grid.ny, grid.nx)).astype(np.float64)), grid=grid, contains_pole=k == 5,
            contains_discontinuity=k == 7, altitude=100.0 + k, magnetic=k == 2))
    res.insert(1, None)                                  # a frame without valid pixels
    descs, payload = pack_results(res, [2, 3, 5, 7], torch.device('cpu'))
    max_frames = 6
    buf = np.zeros(max_frames * DESC_LEN + payload.numel())
    buf[:descs.numel()] = descs.numpy().ravel()
    buf[max_frames * DESC_LEN:] = payload.numpy()
    out = (SeqFrame * 8)()
    n = C.c_int32()
    rc = L.amt_seq_unpack(buf.ctypes.data_as(C.c_void_p), buf.size, 4, max_frames, out, 8, C.byref(n))
    assert rc == 0 and n.value == 3
    for f, r in zip(out[:3], [res[0], res[2], res[3]]):
        ny, nx, nc = f.ny, f.nx, f.nc
        assert (ny, nx, nc) == tuple(r['mean'].shape)
        mean = np.ctypeslib.as_array(C.cast(f.mean, C.POINTER(C.c_double)), (ny, nx, nc))
        cnt = np.ctypeslib.as_array(C.cast(f.count, C.POINTER(C.c_double)), (ny, nx))
        assert np.array_equal(mean, r['mean'].numpy()) and np.array_equal(cnt, r['count'].numpy())
        assert (f.lat0, f.lon0, f.dlat, f.dlon) == (r['grid'].lat0, r['grid'].lon0, r['grid'].latStep, r['grid'].lonStep)
        assert (bool(f.contains_pole), bool(f.contains_discontinuity), bool(f.magnetic), f.altitude) == \
            (r['contains_pole'], r['contains_discontinuity'], r['magnetic'], r['altitude'])
    assert [f.index for f in out[:3]] == [2, 5, 7]
    size = C.c_int64()
    assert L.amt_seq_payload_size(out, 3, C.byref(size)) == 0 and size.value == payload.numel()
    # truncated buffers and impossible descriptors are refused, not read past
    assert L.amt_seq_unpack(buf.ctypes.data_as(C.c_void_p), buf.size - 1, 4, max_frames, out, 8, C.byref(n)) != 0
    assert L.amt_seq_unpack(buf.ctypes.data_as(C.c_void_p), buf.size, 4, 3, out, 8, C.byref(n)) != 0


def test_fits_header_files_and_image_paths(tmp_path):
    """auromat_amd.fits.readHeader on a real astrometry.net .wcs file of the reference's test resources (header-only FITS:
    80-column cards in 2880-byte blocks), writeHeader round trip, and getMapping(imagePath, wcsPath)."""
    from conftest import GOLDEN, header_from, load_golden
    from auromat_amd import fits
    from auromat_amd.mapping.spacecraft import getMapping
    path = os.path.join(GOLDEN, 'resources', 'ISS030-E-102170_dc.wcs')
    hdr = fits.readHeader(path)
    z = load_golden('georef_full_iss030_fast.npz')          # made from the same file by the reference's own reader
    want = header_from(z)
    for k, v in want.items():
        if k.startswith('CTYPE'):
            continue
        assert float(hdr[k]) == float(v), k
    assert hdr['CTYPE1'] == 'RA---TAN' and hdr['CTYPE2'] == 'DEC--TAN' and hdr['SIMPLE'] is True
    assert fits.getPhotoTime(hdr).isoformat() == '2012-01-25T09:27:08.060000'
    # the frame was shifted by -13 s (DATESHIF, POS?SHIF): the mapping uses the shifted cards, the fixture holds them
    cam, t, shift = fits.getShiftedSpacecraftPosition(hdr)
    assert np.array_equal(cam, z['cam']) and t.isoformat() == str(z['time_iso']) and shift.total_seconds() == -13.0
    assert np.array_equal(fits.getSpacecraftPosition(hdr)[0], [hdr['POSX'], hdr['POSY'], hdr['POSZ']])
    out = str(tmp_path / 'copy.wcs')
    fits.writeHeader(out, hdr)
    assert os.path.getsize(out) % 2880 == 0
    again = fits.readHeader(out)
    assert again == {k: v for k, v in hdr.items() if v is not None}
    with pytest.raises(IOError):
        fits.writeHeader(out, hdr)
    # strings with quotes, exponents, comments
    fits.writeHeader(out, {'OBJECT': "it's", 'SMALL': 1.5e-12, 'FLAG': False, 'N': -3}, overwrite=True)
    assert fits.readHeader(out) == {'SIMPLE': True, 'BITPIX': 8, 'NAXIS': 0, 'OBJECT': "it's", 'SMALL': 1.5e-12, 'FLAG': False,
                                    'N': -3}
    # image + header given as paths
    img = np.random.RandomState(0).randint(0, 255, (int(hdr['IMAGEH']) // 16, int(hdr['IMAGEW']) // 16, 3)).astype(np.uint8)
    np.save(str(tmp_path / 'ISS030-E-102170.npy'), img)
    small = dict(hdr, IMAGEW=img.shape[1], IMAGEH=img.shape[0], CRPIX1=hdr['CRPIX1'] / 16, CRPIX2=hdr['CRPIX2'] / 16,
                 **{k: hdr[k] * 16 for k in ('CD1_1', 'CD1_2', 'CD2_1', 'CD2_2')})
    fits.writeHeader(str(tmp_path / 'small.wcs'), small)
    m = getMapping(str(tmp_path / 'ISS030-E-102170.npy'), str(tmp_path / 'small.wcs'))
    assert m.identifier == 'ISS030-E-102170' and m.altitude == 110
    assert m.photoTime == t and np.array_equal(m.cameraPosGCRS, cam)


def test_spacecraft_mapping_providers(tmp_path):
    """SpacecraftMappingProvider / SpacecraftMappingPathProvider (reference spacecraft.py:40-300) over a folder with the
    ten real headers of the reference's seq/ resources and an image file each."""
    from datetime import timedelta
    from auromat_amd import fits
    from auromat_amd.mapping.spacecraft import SpacecraftMappingPathProvider, SpacecraftMappingProvider
    from auromat_amd.utils import findNearest
    assert [findNearest([1, 3, 7], x) for x in (0, 1, 2, 2.1, 5, 6, 9)] == [0, 0, 0, 1, 1, 2, 2]
    src = os.path.join(GOLDEN, 'resources', 'seq')
    d = str(tmp_path / 'frames')
    os.makedirs(d)
    names = sorted(os.listdir(src))
    img = np.zeros((4, 6, 3), np.uint8)
    for n in names:
        # the real cards, for a 6 x 4 image (a mapping checks its image against IMAGEW / IMAGEH)
        hdr = fits.readHeader(os.path.join(src, n))
        fits.writeHeader(os.path.join(d, n), dict(hdr, IMAGEW=6, IMAGEH=4))
        np.save(os.path.join(d, n[:-4] + '.npy'), img)
    np.save(os.path.join(d, 'ISS029-E-9999.npy'), img)               # an image without a solution
    with open(os.path.join(d, 'metadata.json'), 'w') as fp:
        json.dump({'sequence_metadata': {'lens': '24mm', 'start': '2011-09-18T11:54:59'},
                   'image_metadata': {n[:-4]: {'iso': 3200 + k} for k, n in enumerate(names)}}, fp)
    prov = SpacecraftMappingProvider(d, altitude=120, fastCenterCalculation=True)
    assert len(prov) == 10 and prov.imageFileExtension == 'npy' and prov.unsolvedIds == ['ISS029-E-9999']
    assert prov.ids == [n[:-4] for n in names]
    t0 = fits.getPhotoTime(fits.readHeader(os.path.join(src, names[0])))
    assert prov.range[0] == t0 and (prov.range[1] - t0).total_seconds() == 27.0
    assert prov.contains(t0 + timedelta(seconds=1)) and not prov.contains(t0 - timedelta(seconds=4))
    m = prov.get(t0 + timedelta(seconds=4))                          # nearest within maxTimeOffset = 3 s: the second frame
    assert m.identifier == 'ISS029-E-8494' and m.altitude == 120 and m.photoTime == t0 + timedelta(seconds=3)
    assert m.metadata == {'lens': '24mm', 'start': datetime(2011, 9, 18, 11, 54, 59), 'iso': 3201}
    with pytest.raises(ValueError):
        prov.get(t0 - timedelta(seconds=10))
    assert prov.getById('8502').identifier == 'ISS029-E-8502'
    with pytest.raises(ValueError):
        prov.getById('ISS029')
    seq = list(prov.getSequence())
    assert [q.identifier for q in seq] == prov.ids and all(q.fastCenterCalculation for q in seq)
    # explicit path lists, deliberately out of order
    wcs = [os.path.join(d, n) for n in reversed(names)]
    imgs = [w[:-4] + '.npy' for w in wcs]
    both = SpacecraftMappingProvider(imgs, wcs)
    assert both.ids == prov.ids and both.imageFileExtension == 'npy'
    pp = SpacecraftMappingPathProvider(imgs, wcs, metadataPath=os.path.join(d, 'metadata.json'))
    assert len(pp) == 10 and pp.range == prov.range and pp.imageFileExtension == 'npy'
    assert [q.identifier for q in pp.getSequence()] == prov.ids
    with pytest.raises(NotImplementedError):
        pp.get(t0)


def test_native_frame_params_equal_the_python_ones():
    """amt_frame_params_from_wcs (the host scalars the frame pipelines and the native sequence runner use) against the
    Python functions, which are pinned to the reference's doubles: WCS rotation, J2000 -> GEO and J2000 -> SM over 24
    years of dates and random pointings — equal to 4e-16 (most elements to the last bit; NumPy's BLAS and the C++
    products round alike but for the odd element), the scalar fields exactly."""
    from datetime import datetime, timedelta
    from auromat_amd.mapping.astrometry import frame_params, frame_params_python
    from auromat_amd.synthetic import frame_header
    rs = np.random.RandomState(7)
    same = total = 0
    for k in range(200):
        hdr, cam, t = frame_header(640, 420, ('iss030', 'iss029')[k % 2])
        hdr = dict(hdr, CRVAL1=float(rs.uniform(0, 360)), CRVAL2=float(rs.uniform(-89, 89)), LONPOLE=float(rs.choice([180.0, 93.5])))
        t = datetime(1995, 1, 1) + timedelta(seconds=float(rs.uniform(0, 24 * 365.25 * 86400)))
        for magnetic in (True, False):
            a, b = frame_params(hdr, 110 + k, cam, t, k % 3 == 0, magnetic), frame_params_python(hdr, 110 + k, cam, t, k % 3 == 0, magnetic)
            for name in ('rot', 'm_geo', 'm_sm'):
                x, y = np.array(getattr(a, name)[:]), np.array(getattr(b, name)[:])
                assert np.max(np.abs(x - y)) <= 4e-16, (name, k)
                same += int((x == y).sum())
                total += x.size
            for name in ('cd', 'crpix', 'cam'):
                assert list(getattr(a, name)) == list(getattr(b, name)), name
            assert (a.a, a.b, a.a0, a.b0, a.width, a.height, a.fast_center) == (b.a, b.b, b.a0, b.b0, b.width, b.height, b.fast_center)
    assert same > 0.9 * total
    hdr, cam, t = frame_header(64, 42)
    with pytest.raises(ValueError, match='IGRF'):
        frame_params(hdr, 110, cam, datetime(2031, 1, 1), True, magnetic=True)
    frame_params(hdr, 110, cam, datetime(2031, 1, 1), True, magnetic=False)       # J2000 -> GEO needs no IGRF


def test_zenithal_direction_generator():
    """coordinates.wcs.zenithal_pix2world (non-TAN headers, which the reference hands to astropy.wcs: wcs.py:54-56;
    astropy is absent, restated from Calabretta & Greisen 2002): for a TAN header the oracle's pinned TAN directions; for
    SIN / ARC / STG / ZEA the projection's own radius law R(theta) around the reference pixel, the same position angles as
    TAN, the reference pixel on CRVAL; SIP terms move a pixel like the polynomial says."""
    from oracle import ref_numpy as O
    from auromat_amd.coordinates.wcs import is_plain_tan, projection_of, zenithal_pix2world
    from auromat_amd.synthetic import frame_header
    hdr, cam, t = frame_header(96, 64, 'iss030')
    assert is_plain_tan(hdr) and projection_of(hdr) == ('TAN', False)
    for corner in (True, False):
        got = zenithal_pix2world(hdr, 96, 64, corner=corner)
        want = O.pixel_directions(hdr, corner=corner)
        assert got.shape == want.shape and np.max(np.abs(got - want)) < 5e-15
    tan = zenithal_pix2world(hdr, 96, 64, corner=False)
    a0, d0 = np.deg2rad(hdr['CRVAL1']), np.deg2rad(hdr['CRVAL2'])
    bore = np.array([np.cos(d0) * np.cos(a0), np.cos(d0) * np.sin(a0), np.sin(d0)])
    k = 180 / np.pi
    # offsets of the pixel centres from the reference pixel in intermediate world coordinates (degrees)
    u = np.arange(96) - hdr['CRPIX1'] + 1
    v = (np.arange(64) - hdr['CRPIX2'] + 1)[:, None]
    r = np.sqrt((hdr['CD1_1'] * u + hdr['CD1_2'] * v) ** 2 + (hdr['CD2_1'] * u + hdr['CD2_2'] * v) ** 2)
    laws = {'SIN': lambda th: k * np.cos(th), 'ARC': lambda th: 90 - np.rad2deg(th),
            'STG': lambda th: 2 * k * np.tan((np.pi / 2 - th) / 2), 'ZEA': lambda th: 2 * k * np.sin((np.pi / 2 - th) / 2)}
    for proj, law in laws.items():
        h2 = dict(hdr, CTYPE1='RA---' + proj, CTYPE2='DEC--' + proj)
        assert not is_plain_tan(h2)
        d = zenithal_pix2world(h2, 96, 64, corner=False)
        assert np.allclose(np.linalg.norm(d, axis=2), 1, atol=1e-14)
        theta = np.arcsin(np.clip(d.dot(bore), -1, 1))                   # native latitude = 90 deg - distance from CRVAL
        assert np.max(np.abs(law(theta) - r)) < 1e-9, proj
        # same position angle about the boresight as TAN: the components perpendicular to it are parallel
        pt, pd = tan - tan.dot(bore)[..., None] * bore, d - d.dot(bore)[..., None] * bore
        cosang = (pt * pd).sum(-1) / (np.linalg.norm(pt, axis=2) * np.linalg.norm(pd, axis=2))
        assert np.min(cosang) > 1 - 1e-12, proj
    # the reference pixel looks at CRVAL
    h3 = dict(hdr, CTYPE1='RA---ARC', CTYPE2='DEC--ARC', CRPIX1=11.0, CRPIX2=7.0)
    assert np.allclose(zenithal_pix2world(h3, 96, 64, corner=False)[6, 10], bore, atol=1e-15)
    # SIP: u' = u + A_2_0 u^2 — the direction of pixel (x, y) is the undistorted direction of the shifted pixel
    hs = dict(hdr, CTYPE1='RA---TAN-SIP', CTYPE2='DEC--TAN-SIP', A_ORDER=2, B_ORDER=2, A_2_0=1e-4, B_1_1=-2e-4)
    assert projection_of(hs) == ('TAN', True) and not is_plain_tan(hs)
    ds = zenithal_pix2world(hs, 96, 64, corner=False)
    x, y = 80, 50
    uu, vv = x - hdr['CRPIX1'] + 1, y - hdr['CRPIX2'] + 1
    shifted = O.pixel_directions(dict(hdr, IMAGEW=1, IMAGEH=1, CRPIX1=1 - (uu + 1e-4 * uu * uu), CRPIX2=1 - (vv - 2e-4 * uu * vv)), corner=False)[0, 0]
    assert np.max(np.abs(ds[y, x] - shifted)) < 1e-13
    with pytest.raises(NotImplementedError):
        zenithal_pix2world(dict(hdr, CTYPE1='RA---AIT', CTYPE2='DEC--AIT'), 8, 8)
