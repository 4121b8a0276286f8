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


def triangulate(pts, threads=None, parallel_min=1000):
    """threads: None = the library's defaults (small inputs: the sequential build), else amt_delaunay_create_threads"""
    from auromat_amd._native import lib
    L = lib()
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    h = C.c_void_p()
    if threads is None:
        rc = L.amt_delaunay_create(pts.ctypes.data_as(C.c_void_p), len(pts), C.byref(h))
    else:
        rc = L.amt_delaunay_create_threads(pts.ctypes.data_as(C.c_void_p), len(pts), threads, parallel_min, C.byref(h))
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
    info = (C.c_int64 * 2)()
    assert L.amt_delaunay_build_info(h, info) == 0
    return dict(handle=h, tri=tri, nbr=nbr, indptr=indptr, ind=ind, dup=nd.value, lib=L, stats=list(stats), strips=info[0], flips=info[1])


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


def oriented(t):
    """triangles as rotations that start at their smallest vertex: equal sets = the same triangles with the same orientation"""
    t = np.asarray(t)
    k = np.argmin(t, axis=1)
    return set(map(tuple, np.stack([t[np.arange(len(t)), (k + i) % 3] for i in range(3)], axis=1)))


def check_structure(pts, d, exact=False):
    """counter-clockwise triangles, mutual neighbours across the shared edge, every inner edge locally Delaunay, the triangles
    cover the convex hull once, the vertex lists are the edges.  exact: decided in exact arithmetic — Python integers for
    integer coordinates, fractions for doubles (slivers of 1e-16 have no sign in floating point)."""
    tri, nbr = d['tri'], d['nbr']
    if exact and pts.dtype.kind == 'f':
        from fractions import Fraction
        P = np.array([[Fraction(float(x)), Fraction(float(y))] for x, y in pts], dtype=object)
    else:
        P = pts.astype(object) if exact else pts
    src = P if exact else pts
    a, b, c = (src[tri[:, k]] for k in range(3))
    area2 = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    assert (area2 > 0).all()
    import scipy.spatial
    total = float(area2.sum())
    assert abs(total / 2 - scipy.spatial.ConvexHull(np.asarray(pts, dtype=np.float64)).volume) <= 1e-9 * abs(total)
    edges = set()
    for t in range(len(tri)):
        for k in range(3):
            x, y = int(tri[t, (k + 1) % 3]), int(tri[t, (k + 2) % 3])
            edges.add((x, y))
            n = nbr[t, k]
            if n < 0:
                edges.add((y, x))
                continue
            m = [j for j in range(3) if nbr[n, j] == t]
            assert len(m) == 1 and {int(tri[n, (m[0] + 1) % 3]), int(tri[n, (m[0] + 2) % 3])} == {x, y}
            # the vertex across the edge is not inside this triangle's circle
            q = P[tri[n, m[0]]]
            A, B, Cc = P[tri[t, 0]], P[tri[t, 1]], P[tri[t, 2]]
            rows = [(u[0] - q[0], u[1] - q[1], (u[0] - q[0]) ** 2 + (u[1] - q[1]) ** 2) for u in (A, B, Cc)]
            det = (rows[0][0] * (rows[1][1] * rows[2][2] - rows[1][2] * rows[2][1])
                   - rows[0][1] * (rows[1][0] * rows[2][2] - rows[1][2] * rows[2][0])
                   + rows[0][2] * (rows[1][0] * rows[2][1] - rows[1][1] * rows[2][0]))
            if exact:
                assert det <= 0, (t, k)
            else:
                scale = max(abs(v) for r in rows for v in r) ** 4
                assert det <= 1e-9 * scale, (t, k, det)
    got = set((v, int(w)) for v in range(len(pts)) for w in d['ind'][d['indptr'][v]:d['indptr'][v + 1]])
    assert got == edges and len(d['ind']) == len(edges)


def bent_lattice(h, w, seed=0):
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(float)
    lon = x * 0.01 + 1e-6 * y * y + 3e-7 * x * y + rs.uniform(-2e-4, 2e-4, x.shape)
    lat = y * 0.008 + 2e-6 * x * x - 1e-7 * x * y + rs.uniform(-2e-4, 2e-4, x.shape)
    keep = rs.rand(h, w) > 0.03                                        # a few holes
    return np.column_stack((lat[keep], lon[keep]))


@pytest.mark.parametrize('kind,threads', [('lattice', 2), ('lattice', 3), ('lattice', 8), ('cloud', 5), ('cloud', 8), ('clusters', 4)])
def test_parallel_build_gives_the_sequential_triangulation(kind, threads):
    """Vertical strips triangulated side by side and joined at their seams (common tangents, the gap filled, Lawson flips):
    the Delaunay triangulation is unique, so every triangle — and its orientation — must be the sequential build's."""
    rs = np.random.RandomState(threads)
    if kind == 'lattice':
        pts = bent_lattice(150, 230, seed=threads)
    elif kind == 'cloud':
        pts = rs.rand(40000, 2) * [3.0, 1.0]
    else:
        centres = rs.rand(12, 2) * 10
        pts = np.concatenate([c + rs.normal(size=(3000, 2)) * rs.uniform(0.05, 1.5) for c in centres])
    seq = triangulate(pts, threads=1)
    par = triangulate(pts, threads=threads, parallel_min=1000)
    try:
        assert seq['strips'] == 1 and par['strips'] >= 2 and par['flips'] > 0
        assert seq['stats'][1] == 0 and seq['stats'][3] == 0 and par['stats'][1] == 0 and par['stats'][3] == 0      # no ties: unique
        assert oriented(par['tri']) == oriented(seq['tri'])
        for v in range(0, len(pts), 7):
            assert set(par['ind'][par['indptr'][v]:par['indptr'][v + 1]]) == set(seq['ind'][seq['indptr'][v]:seq['indptr'][v + 1]])
        if kind == 'lattice' and threads == 3:
            check_structure(pts, par)
    finally:
        seq['lib'].amt_delaunay_destroy(seq['handle'])
        par['lib'].amt_delaunay_destroy(par['handle'])


@pytest.mark.parametrize('name', ['resample_nearest_iss030.npz', 'resample_nearest_iss029.npz', 'resample_nearest_synth_plain.npz',
                                  'resample_nearest_synth_disc.npz', 'resample_nearest_synth_pole.npz'])
def test_parallel_build_equals_qhull_on_the_reference_fixtures(name):
    """(the curved outline of a camera frame makes strips whose hulls lie inside the wedge of their neighbours' tangents: only one
    vertex of such a strip is on the joint hull and the whole ring of its hull edges faces the seam)"""
    import scipy.spatial
    pts = fixture_points(name)
    want = canon(scipy.spatial.Delaunay(pts).simplices)
    for threads in (2, 3, 5, 8):
        d = triangulate(pts, threads=threads, parallel_min=300)
        try:
            assert d['strips'] >= 2, (name, threads, len(pts))
            assert canon(d['tri']) == want, (name, threads)
        finally:
            d['lib'].amt_delaunay_destroy(d['handle'])


@pytest.mark.parametrize('shape', ['arc', 'annulus', 'turned lattice', 'wedge'])
def test_parallel_build_on_shapes_that_are_not_convex(shape):
    rs = np.random.RandomState(11)
    if shape == 'arc':
        th, r = rs.uniform(0.2, 2.9, 30000), rs.uniform(9.0, 10.0, 30000)
        pts = np.column_stack((r * np.cos(th), r * np.sin(th)))
    elif shape == 'annulus':
        th, r = rs.uniform(0, 2 * np.pi, 30000), rs.uniform(4.0, 5.0, 30000)
        pts = np.column_stack((r * np.cos(th), r * np.sin(th)))
    elif shape == 'turned lattice':
        q = bent_lattice(120, 200, seed=5)
        c, s_ = np.cos(0.7), np.sin(0.7)
        pts = q.dot(np.array([[c, -s_], [s_, c]]))
    else:
        u = rs.rand(30000, 2)
        pts = np.column_stack((u[:, 0] * 10, (u[:, 1] - 0.5) * u[:, 0] * 0.3))            # a thin wedge opening to the right
    seq = triangulate(pts, threads=1)
    try:
        for threads in (2, 4, 7):
            par = triangulate(pts, threads=threads, parallel_min=300)
            try:
                assert par['strips'] >= 2
                assert oriented(par['tri']) == oriented(seq['tri']), (shape, threads)
            finally:
                par['lib'].amt_delaunay_destroy(par['handle'])
    finally:
        seq['lib'].amt_delaunay_destroy(seq['handle'])


@pytest.mark.parametrize('threads', [1, 3])
def test_exact_lattices_give_a_valid_delaunay_triangulation(threads):
    """Integer lattices: every cell cocircular, every hull side and every seam between strips a run of collinear points.  No
    unique answer exists (the header of csrc/amt_delaunay.hip) — but whatever comes out must be A Delaunay triangulation: checked
    in exact integer arithmetic."""
    y, x = np.mgrid[0:70, 0:190]
    pts = np.column_stack((x.ravel(), y.ravel())).astype(np.float64)
    d = triangulate(pts, threads=threads, parallel_min=1000)
    try:
        assert d['dup'] == 0 and len(d['tri']) == 2 * 69 * 189
        assert (d['strips'] >= 2) == (threads > 1)
        check_structure(pts.astype(np.int64), d, exact=True)
    finally:
        d['lib'].amt_delaunay_destroy(d['handle'])
