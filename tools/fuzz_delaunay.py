"""Fuzz of the parallel build of the host triangulator (csrc/amt_delaunay.hip: strips side by side, joined at their seams) against
the sequential build: random sizes, thread counts and point sets — clouds, clusters, bent lattices with holes, thin slivers, rows
that are collinear up to rounding, exact lattices, points on circles, duplicates.  Where no tie was met (amt_delaunay_stats) the
two must be the same triangles with the same orientation; with ties the parallel result is checked as A Delaunay triangulation
(tests/test_delaunay_cpu.check_structure).  CPU only.  usage: fuzz_delaunay.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import test_delaunay_cpu as T

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = ties = fallbacks = 0
for case in range(cases):
    kind = ('cloud', 'clusters', 'lattice', 'sliver', 'rows', 'exact', 'circles', 'dups', 'arc', 'turned', 'blobs')[rs.randint(11)]
    n = int(rs.randint(3000, 60000))
    if kind == 'cloud':
        pts = rs.rand(n, 2) * [rs.uniform(0.1, 10), 1.0]
    elif kind == 'clusters':
        pts = np.concatenate([rs.rand(2) * 10 + rs.normal(size=(n // 8, 2)) * rs.uniform(0.01, 2) for _ in range(8)])
    elif kind == 'lattice':
        pts = T.bent_lattice(int(rs.randint(40, 200)), int(rs.randint(60, 300)), seed=case)
    elif kind == 'sliver':
        pts = rs.rand(n, 2) * [1000.0, 1e-3]
    elif kind == 'rows':
        y, x = np.mgrid[0:60, 0:n // 60].astype(float)
        pts = np.column_stack(((x * 0.013 + y * 0.0007).ravel(), (y * 0.011 - x * 0.0003).ravel()))      # straight rows, rounded
    elif kind == 'exact':
        y, x = np.mgrid[0:int(rs.randint(20, 90)), 0:int(rs.randint(80, 300))]
        pts = np.column_stack((x.ravel(), y.ravel())).astype(float)
    elif kind == 'circles':
        th = rs.rand(n) * 2 * np.pi
        r = rs.randint(1, 6, n).astype(float)
        pts = np.concatenate([np.column_stack((r * np.cos(th), r * np.sin(th))), rs.rand(n // 4, 2) * 12 - 6])
    elif kind == 'arc':
        th, r = rs.uniform(rs.uniform(0, 1), rs.uniform(2, 6.2), n), rs.uniform(9.0, 9.0 + rs.uniform(0.1, 3), n)
        pts = np.column_stack((r * np.cos(th), r * np.sin(th)))
    elif kind == 'turned':
        q = T.bent_lattice(int(rs.randint(40, 160)), int(rs.randint(60, 260)), seed=case)
        ang = rs.uniform(0, np.pi)
        pts = q.dot(np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]))
    elif kind == 'blobs':
        # blobs strung along a curve: strips whose hulls sit inside their neighbours' tangent wedges
        t = np.sort(rs.rand(12)) * 10
        pts = np.concatenate([np.array([u, 3 * np.sin(u)]) + rs.normal(size=(n // 12, 2)) * rs.uniform(0.02, 0.4) for u in t])
    else:
        base = rs.rand(n // 2, 2)
        pts = np.concatenate([base, base[rs.randint(0, len(base), n // 3)], rs.rand(n // 4, 2)])
    threads = int(rs.randint(2, 9))
    seq = T.triangulate(pts, threads=1)
    par = T.triangulate(pts, threads=threads, parallel_min=500)
    tag = '%d: %s n %d threads %d strips %s flips %s' % (case, kind, len(pts), threads, par['strips'], par['flips'])
    try:
        fallbacks += par['strips'] == 1
        if par['dup'] != seq['dup'] or len(par['tri']) != len(seq['tri']):
            bad += 1
            print('SIZES DIFFER', tag, par['dup'], seq['dup'], len(par['tri']), len(seq['tri']))
            continue
        tie = seq['stats'][1] or seq['stats'][3] or par['stats'][1] or par['stats'][3]
        if not tie:
            if T.oriented(par['tri']) != T.oriented(seq['tri']):
                bad += 1
                print('TRIANGLES DIFFER', tag)
        else:
            ties += 1
            if kind == 'exact' or len(pts) < 20000:
                try:
                    T.check_structure(pts.astype(np.int64) if kind == 'exact' else pts, par, exact=True)
                except AssertionError as e:
                    bad += 1
                    print('NOT A DELAUNAY TRIANGULATION', tag, e)
    finally:
        seq['lib'].amt_delaunay_destroy(seq['handle'])
        par['lib'].amt_delaunay_destroy(par['handle'])
print('cases', cases, 'with ties', ties, 'sequential fallbacks', fallbacks, 'failures', bad)
sys.exit(1 if bad else 0)
