import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def header_from(z):
    hdr = {'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN'}
    for k in z.files:
        if k.startswith('hdr_'):
            v = float(z[k])
            hdr[k[4:]] = int(v) if k[4:] in ('IMAGEW', 'IMAGEH') else v
    return hdr


_ORACLE_FRAMES = {}


def oracle_frame(hdr, altitude, cam, m_geo, m_sm, fast=True):
    """oracle.ref_numpy.georef_frame, remembered for the session: the full-size frames of the reference's tests are each asked
    for by several test files (half a minute of NumPy per run).  Keyed by every input; the arrays come back read-only, the two
    latest frames are kept (1.3 GB each)."""
    from oracle import ref_numpy as O
    key = (tuple(sorted((k, repr(v)) for k, v in hdr.items())), float(altitude), np.asarray(cam, dtype=np.float64).tobytes(),
           np.asarray(m_geo, dtype=np.float64).tobytes(), None if m_sm is None else np.asarray(m_sm, dtype=np.float64).tobytes(),
           bool(fast))
    if key not in _ORACLE_FRAMES:
        while len(_ORACLE_FRAMES) >= 2:
            del _ORACLE_FRAMES[next(iter(_ORACLE_FRAMES))]
        g = O.georef_frame(hdr, altitude, cam, m_geo, m_sm, fast=fast)
        for v in g.values():
            if isinstance(v, np.ndarray):
                v.setflags(write=False)
        _ORACLE_FRAMES[key] = g
    return _ORACLE_FRAMES[key]


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def assert_counts_equal_up_to_edge_pixels(want, got_count, lat_c, lon_c, what=''):
    """
    Bin counts of the HIP path against the oracle's (`want`: its resample_mean result, grid included).  They must be equal —
    except for a pixel that sits on a bin edge to rounding: the device's coordinates are within ~1e-11 deg of the oracle's,
    and such a pixel may fall on the other side.  That case is not waved through as "a few cells may differ": the two cells
    must be neighbours, one a count up and one down, and a pixel of the ORACLE must be found within 1e-9 deg of their common
    edge; the assertion messages say how far the nearest one is.
    """
    import numpy as np
    diff = np.argwhere(want['count'] != got_count)
    if len(diff) == 0:
        return 0
    assert len(diff) % 2 == 0 and want['count'].sum() == got_count.sum(), \
        '%s: %d cells differ and the totals are %d / %d' % (what, len(diff), want['count'].sum(), got_count.sum())
    lat_edges, lon_edges = np.asarray(want['lat'])[:, 0], np.asarray(want['lon'])[0]      # corner grid = bin edges
    la, lo = np.asarray(lat_c, dtype=np.float64).ravel(), np.asarray(lon_c, dtype=np.float64).ravel()
    ok = ~np.isnan(la)
    la, lo = la[ok], lo[ok]
    left = [tuple(c) for c in diff]
    pairs = 0
    while left:
        a = left.pop()
        partner = [b for b in left if abs(a[0] - b[0]) + abs(a[1] - b[1]) == 1]
        assert partner, '%s: cell %s differs by %d without a neighbour that differs' % (
            what, a, int(got_count[a] - want['count'][a]))
        b = partner[0]
        left.remove(b)
        assert int(got_count[a] - want['count'][a]) == -int(got_count[b] - want['count'][b]) and \
            abs(int(got_count[a] - want['count'][a])) == 1, (what, a, b)
        if a[0] != b[0]:                                    # the common edge is a latitude edge (rows run north to south)
            edge = lat_edges[max(a[0], b[0])]
            inside = (lo >= min(lon_edges[a[1]], lon_edges[a[1] + 1])) & (lo <= max(lon_edges[a[1]], lon_edges[a[1] + 1]))
            dist = np.abs(la[inside] - edge).min() if inside.any() else np.inf
        else:
            edge = lon_edges[max(a[1], b[1])]
            inside = (la >= min(lat_edges[a[0]], lat_edges[a[0] + 1])) & (la <= max(lat_edges[a[0]], lat_edges[a[0] + 1]))
            dist = np.abs(lo[inside] - edge).min() if inside.any() else np.inf
        assert dist < 1e-9, '%s: cells %s / %s differ by one pixel, but the oracle\'s pixel nearest to their common edge is ' \
                            '%.3e deg away from it' % (what, a, b, dist)
        pairs += 1
    return pairs
