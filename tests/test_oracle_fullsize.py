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
