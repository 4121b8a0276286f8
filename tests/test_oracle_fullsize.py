"""
CPU: the oracle at the reference's native frame size (4256x2832, real header ISS029-E-8492.wcs)
against strided samples and whole-array digests of the real reference's arrays
(tests/golden/georef_full_iss029_fast.npz, made by oracle/make_golden.py).  ~25 s.
"""
import numpy as np

from conftest import header_from, load_golden
from oracle import ref_numpy as O


def test_oracle_full_size_bit_exact():
    z = load_golden('georef_full_iss029_fast.npz')
    hdr = header_from(z)
    step = int(z['step'])
    g = O.georef_frame(hdr, float(z['altitude']), z['cam'], z['m_geo'], z['m_sm'], fast=True)
    for k in ('p_corner', 'lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlt', 'mlat_c', 'mlt_c'):
        a, b = g[k][::step, ::step], z[k]
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), k
    for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev'):
        a = g[k]
        ok = ~np.isnan(a)
        n, s, lo, hi, sabs = z['digest_' + k]
        assert ok.sum() == n and a[ok].min() == lo and a[ok].max() == hi
        assert a[ok].sum() == s and np.abs(a[ok]).sum() == sabs
    # the survey's anchors for this fixture
    assert abs(np.nanmin(g['lat']) - (-68.3209)) < 1e-4 and abs(np.nanmax(g['lon']) - 179.9564) < 1e-4
    assert np.isnan(g['lat'][0, 2000]) and not np.isnan(g['lat'][-1, 2000])      # intersection_test.py:155-169


def test_mapping_centroid_known_answer():
    """outline_test.py:151-158 testMappingCentroid on the real ISS030-E-102170 header: centroid of the traced outline
    of the valid corners = [55.00295889563608, -99.21825084682715] to 6 decimals (the literal is an OpenCV float32
    moment; "the two solutions differ slightly after the 7th decimal").  Pins the restated find_contours including
    its orientation (the centroid formula flips sign with it)."""
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, 'known_answers.json')) as fp:
        ka = json.load(fp)['outline']['mapping_centroid']
    z = load_golden('georef_full_iss030_fast.npz')
    hdr = header_from(z)
    dirs = O.pixel_directions(hdr, corner=True)
    hit = O.inflated_earth_intersection(dirs.reshape(-1, 3), z['cam'], float(z['altitude']))
    lat, lon = O.j2000_to_latlon(hit, z['m_geo'])
    lat, lon = lat.reshape(dirs.shape[:2]), lon.reshape(dirs.shape[:2])
    outl = O.outline(~np.isnan(lat))
    assert len(outl) > 10000
    centroid = O.polygon_centroid(np.transpose([lat[outl[:, 1], outl[:, 0]], lon[outl[:, 1], outl[:, 0]]]))
    np.testing.assert_almost_equal(centroid, ka['expect'], decimal=ka['decimals'])
