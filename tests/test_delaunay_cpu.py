"""
The triangulation behind method='linear' / 'cubic' (reference resample.py:323-326: scipy.interpolate.griddata, i.e.
scipy.spatial.Delaunay = Qhull, a third-party dependency that is not part of /root/reference): amt_delaunay_* (host code of the
library, csrc/amt_delaunay.hip) against scipy.spatial.Delaunay itself — the Delaunay triangulation of a point set is unique
unless four points are cocircular, so the two must agree triangle for triangle — on the inputs of every 'cubic' fixture made
with the real reference, on a jittered lattice with holes and on degenerate inputs.  No GPU needed: these entry points are host
arithmetic.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden


def triangulate(pts):
    from auromat_amd._native import lib
    L = lib()
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    h = C.c_void_p()
    rc = L.amt_delaunay_create(pts.ctypes.data_as(C.c_void_p), len(pts), C.byref(h))
    if rc != 0:
        return rc
    nt, nn, nd = C.c_int64(), C.c_int64(), C.c_int64()
    assert L.amt_delaunay_sizes(h, C.byref(nt), C.byref(nn), C.byref(nd)) == 0
    tri, nbr = np.empty((nt.value, 3), np.int32), np.empty((nt.value, 3), np.int32)
    assert L.amt_delaunay_triangles(h, tri.ctypes.data_as(C.c_void_p), nbr.ctypes.data_as(C.c_void_p)) == 0
    indptr, ind = np.empty(len(pts) + 1, np.int64), np.empty(nn.value, np.int32)
    assert L.amt_delaunay_vertex_neighbours(h, indptr.ctypes.data_as(C.c_void_p), ind.ctypes.data_as(C.c_void_p)) == 0
    stats = (C.c_int64 * 4)()
    assert L.amt_delaunay_stats(h, stats) == 0
    return dict(handle=h, tri=tri, nbr=nbr, indptr=indptr, ind=ind, dup=nd.value, lib=L, stats=list(stats))


def canon(t):
    return set(map(tuple, np.sort(np.asarray(t), axis=1)))


def fixture_points(name):
    z = load_golden(name)
    lat, lon = z['lats_c'], z['lons_c']
    ok = ~np.isnan(lat.ravel())
    pts = np.column_stack((lat.ravel()[ok], lon.ravel()[ok]))
    if bool(z['contains_discontinuity']) and not bool(z['contains_pole']):
        pts[:, 1] = ((pts[:, 1] + 360.0) % 360.0) - 180.0              # the reference shifts by 180 deg there (resample.py:212-218)
    return pts


@pytest.mark.parametrize('name', ['resample_nearest_iss030.npz', 'resample_nearest_iss029.npz', 'resample_nearest_synth_plain.npz',
                                  'resample_nearest_synth_disc.npz', 'resample_nearest_synth_pole.npz'])
def test_triangulation_equals_qhulls_on_the_reference_fixtures(name):
    import scipy.spatial
    pts = fixture_points(name)
    ref = scipy.spatial.Delaunay(pts)
    d = triangulate(pts)
    try:
        assert d['dup'] == 0 and len(ref.coplanar) == 0
        # no tie was met while triangulating (no three points collinear, no four cocircular, even at 113 bits): this IS the
        # unique Delaunay triangulation of the fixture's points — equal to Qhull's by uniqueness, not by luck
        print(name, 'predicates beyond double precision: orientation %d (zeros %d), in-circle %d (undecided %d)' % tuple(d['stats']))
        assert d['stats'][1] == 0 and d['stats'][3] == 0, d['stats']
        assert canon(d['tri']) == canon(ref.simplices)
        # counter-clockwise, and every neighbour relation mutual across the shared edge
        a, b, c = (pts[d['tri'][:, k]] for k in range(3))
        assert ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]) > 0).all()
        for t in range(0, len(d['tri']), 97):
            for k in range(3):
                n = d['nbr'][t, k]
                if n >= 0:
                    edge = {d['tri'][t, (k + 1) % 3], d['tri'][t, (k + 2) % 3]}
                    assert edge <= set(d['tri'][n]) and t in d['nbr'][n]
        ip, ii = ref.vertex_neighbor_vertices
        for v in range(len(pts)):
            assert set(ii[ip[v]:ip[v + 1]]) == set(d['ind'][d['indptr'][v]:d['indptr'][v + 1]]), v
        # point location: scipy's find_simplex on a grid over the box (inside / outside and the triangle's vertices)
        gx, gy = np.meshgrid(np.linspace(pts[:, 0].min() - 0.2, pts[:, 0].max() + 0.2, 40),
                             np.linspace(pts[:, 1].min() - 0.2, pts[:, 1].max() + 0.2, 40), indexing='ij')
        tg = np.ascontiguousarray(np.column_stack((gx.ravel(), gy.ravel())))
        m = len(tg)
        vert, cen, has = np.empty((m, 3), np.int32), np.empty((m, 3, 2)), np.empty((m, 3), np.uint8)
        assert d['lib'].amt_delaunay_locate(d['handle'], tg.ctypes.data_as(C.c_void_p), m, vert.ctypes.data_as(C.c_void_p),
                                            cen.ctypes.data_as(C.c_void_p), has.ctypes.data_as(C.c_void_p)) == 0
        simplex = ref.find_simplex(tg)
        assert np.array_equal(vert[:, 0] >= 0, simplex >= 0)
        inside = np.flatnonzero(simplex >= 0)
        assert len(inside) > 100
        for i in inside[::7]:
            assert set(vert[i]) == set(ref.simplices[simplex[i]]), i
            for k in range(3):
                # the triangle across the edge opposite vertex k, by its centroid
                opp = [s for s in range(3) if ref.simplices[simplex[i]][s] == vert[i, k]][0]
                n = ref.neighbors[simplex[i]][opp]
                assert (n >= 0) == bool(has[i, k])
                if n >= 0:
                    assert np.allclose(cen[i, k], pts[ref.simplices[n]].sum(axis=0) / 3, rtol=0, atol=1e-12)
    finally:
        d['lib'].amt_delaunay_destroy(d['handle'])


def test_triangulation_of_a_jittered_lattice_with_holes_and_of_degenerate_inputs():
    import scipy.spatial
    rng = np.random.RandomState(4)
    ii, jj = np.mgrid[0:90, 0:120].astype(np.float64)
    lat = 40 + 0.05 * ii + 0.004 * jj + rng.uniform(-0.01, 0.01, ii.shape)
    lon = 5 + 0.01 * ii + 0.07 * jj + rng.uniform(-0.01, 0.01, ii.shape)
    keep = rng.rand(*ii.shape) > 0.05
    keep[20:40, 30:70] = False
    pts = np.column_stack((lat[keep], lon[keep]))
    d = triangulate(pts)
    ref = scipy.spatial.Delaunay(pts)
    assert canon(d['tri']) == canon(ref.simplices)
    d['lib'].amt_delaunay_destroy(d['handle'])
    # any order of the points gives the same triangles
    perm = rng.permutation(len(pts))
    e = triangulate(pts[perm])
    assert canon(perm[e['tri']]) == canon(ref.simplices)
    e['lib'].amt_delaunay_destroy(e['handle'])
    # a duplicate point is left out (Qhull lists it as coplanar); collinear points cannot be triangulated
    dup = np.vstack((pts[:50], pts[7:8]))
    f = triangulate(dup)
    assert f['dup'] == 1 and canon(f['tri']) == canon(scipy.spatial.Delaunay(pts[:50]).simplices)
    f['lib'].amt_delaunay_destroy(f['handle'])
    assert triangulate(np.column_stack((np.arange(10.0), 2 * np.arange(10.0)))) != 0
    assert triangulate(pts[:2]) != 0
    # an exactly regular lattice: every cell cocircular — a valid triangulation (two triangles per cell), whichever diagonals
    g = np.column_stack([a.ravel() for a in np.mgrid[0:12, 0:9].astype(np.float64)])
    h = triangulate(g)
    assert len(h['tri']) == 2 * 11 * 8 and h['dup'] == 0
    assert h['stats'][3] > 0 and h['stats'][1] > 0                      # ... and the library says that it met ties
    a, b, c = (g[h['tri'][:, k]] for k in range(3))
    area = 0.5 * ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]))
    assert (area > 0).all() and abs(area.sum() - 11 * 8) < 1e-9
    h['lib'].amt_delaunay_destroy(h['handle'])
