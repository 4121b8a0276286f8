"""CPU experiment behind DESIGN.md section 6: could method='linear' be pinned to the reference?  The reference calls
scipy.interpolate.griddata(method='linear'), i.e. Qhull's Delaunay triangulation of the pixel centres.  This script
interpolates the elevation of the 256 x 170 fixture frame on the triangulation a GPU kernel could build without Qhull —
every pixel quad split along the diagonal the incircle test picks — and compares triangles and values with scipy's.
Result (2026-10): 25 373 of 41 460 structured triangles are triangles of scipy's triangulation (61 %), 1 977 of 4 063
common target values differ by more than 1e-6 (max 8.4e-4 deg of elevation), 103 targets only scipy fills (hull
triangles).  The quads of a smoothly mapped pixel grid are so close to cocircular that Qhull's choices are not the
incircle test's."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from scipy.interpolate import griddata
from scipy.spatial import Delaunay
from conftest import load_golden
z = load_golden('resample_geo_iss030_ppd10x10.npz')
lat, lon = z['lats_c'], z['lons_c']          # (h, w) NaN = masked
h, w = lat.shape
ok = ~np.isnan(lat)
pts = np.column_stack([lat[ok], lon[ok]])
vals = z['elev'][ok]
tlat, tlon = z['out_lat_c'], z['out_lon_c']
t0 = time.time()
ref = griddata(pts, vals, (tlat, tlon), method='linear')
print('scipy', time.time() - t0, 's; targets', ref.size, 'finite', np.isfinite(ref).sum())
# structured mesh: quads (i,j)-(i+1,j+1), diagonal by the Delaunay criterion (incircle test), only quads with 4 valid corners
def incircle(a, b, c, d):
    # > 0 if d inside the circumcircle of (a, b, c) given counter-clockwise orientation
    m = np.array([[a[0]-d[0], a[1]-d[1], (a[0]-d[0])**2 + (a[1]-d[1])**2],
                  [b[0]-d[0], b[1]-d[1], (b[0]-d[0])**2 + (b[1]-d[1])**2],
                  [c[0]-d[0], c[1]-d[1], (c[0]-d[0])**2 + (c[1]-d[1])**2]])
    return np.linalg.det(m)
def orient(a, b, c):
    return (b[0]-a[0])*(c[1]-a[1]) - (b[1]-a[1])*(c[0]-a[0])
P = np.dstack([lat, lon])
E = z['elev']
tri = []   # list of (p0,p1,p2,(idx...))
for i in range(h - 1):
    for j in range(w - 1):
        q = [(i, j), (i, j + 1), (i + 1, j + 1), (i + 1, j)]
        if not all(ok[a] for a in q):
            continue
        A, B, C, D = [P[a] for a in q]
        o = orient(A, B, C)
        inc = incircle(A, B, C, D) if o > 0 else incircle(A, C, B, D)
        if inc > 0:      # D inside circle(ABC): use diagonal B-D
            tri.append((q[0], q[1], q[3])); tri.append((q[1], q[2], q[3]))
        else:            # diagonal A-C
            tri.append((q[0], q[1], q[2])); tri.append((q[0], q[2], q[3]))
tri = np.array(tri)
print('structured triangles', len(tri))
# locate targets by brute force over triangles (vectorised per target chunk)
T0 = P[tri[:, 0, 0], tri[:, 0, 1]]; T1 = P[tri[:, 1, 0], tri[:, 1, 1]]; T2 = P[tri[:, 2, 0], tri[:, 2, 1]]
V0 = E[tri[:, 0, 0], tri[:, 0, 1]]; V1 = E[tri[:, 1, 0], tri[:, 1, 1]]; V2 = E[tri[:, 2, 0], tri[:, 2, 1]]
den = (T1[:, 1]-T2[:, 1])*(T0[:, 0]-T2[:, 0]) + (T2[:, 0]-T1[:, 0])*(T0[:, 1]-T2[:, 1])
mine = np.full(ref.shape, np.nan)
tl, to = tlat.ravel(), tlon.ravel()
for k in range(tl.size):
    l0 = ((T1[:, 1]-T2[:, 1])*(tl[k]-T2[:, 0]) + (T2[:, 0]-T1[:, 0])*(to[k]-T2[:, 1])) / den
    l1 = ((T2[:, 1]-T0[:, 1])*(tl[k]-T2[:, 0]) + (T0[:, 0]-T2[:, 0])*(to[k]-T2[:, 1])) / den
    l2 = 1 - l0 - l1
    inside = np.nonzero((l0 >= -1e-12) & (l1 >= -1e-12) & (l2 >= -1e-12))[0]
    if inside.size:
        t = inside[0]
        mine.flat[k] = l0[t]*V0[t] + l1[t]*V1[t] + l2[t]*V2[t]
both = np.isfinite(ref) & np.isfinite(mine)
d = np.abs(ref - mine)[both]
print('both finite', both.sum(), 'only scipy', (np.isfinite(ref) & ~np.isfinite(mine)).sum(), 'only mine', (~np.isfinite(ref) & np.isfinite(mine)).sum())
print('max diff', d.max(), 'n > 1e-9', (d > 1e-9).sum(), 'n > 1e-6', (d > 1e-6).sum(), 'median', np.median(d))
# compare with scipy's own triangulation: how many of the structured triangles are Delaunay triangles
dl = Delaunay(pts)
idx = -np.ones((h, w), int); idx[ok] = np.arange(ok.sum())
mine_set = set(tuple(sorted(idx[a[0], a[1]] for a in t)) for t in tri)
sc_set = set(tuple(sorted(s)) for s in dl.simplices)
print('structured', len(mine_set), 'scipy', len(sc_set), 'common', len(mine_set & sc_set))
