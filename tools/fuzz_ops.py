"""Fuzz of the operator-level entry points against the oracle: histogram2d (random ranges / bin counts / explicit
edges, samples exactly on edges and on the right-most edge, NaN samples, weights), ellipsoid and sphere intersections
(inside / outside origins, tangent and missing rays, directed or not), ECEF <-> geodetic, rotations to lat/lon and
MLat/MLT.  usage: fuzz_ops.py [rounds] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datetime import datetime
import numpy as np
from auromat_amd.coordinates.intersection import ellipsoidLineIntersection, sphereLineIntersection
from auromat_amd.coordinates import transform as T
from auromat_amd.util.histogram import histogram2d
from oracle import ref_numpy as O

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    ok = ~np.isnan(a)
    return np.max(np.abs(a[ok] - b[ok]), initial=0.0) <= tol


for it in range(rounds):
    # ---- histogram2d ----
    n = int(rng.randint(1, 4000))
    lo_x, lo_y = rng.uniform(-200, 100), rng.uniform(-90, 50)
    hi_x, hi_y = lo_x + rng.uniform(0.5, 60), lo_y + rng.uniform(0.5, 40)
    nx, ny = int(rng.randint(1, 60)), int(rng.randint(1, 60))
    x = rng.uniform(lo_x - 2, hi_x + 2, n)
    y = rng.uniform(lo_y - 2, hi_y + 2, n)
    ex, ey = np.linspace(lo_x, hi_x, nx + 1), np.linspace(lo_y, hi_y, ny + 1)
    k = n // 5
    x[:k] = ex[rng.randint(0, nx + 1, k)]                          # exactly on edges, the last one included
    y[k:2 * k] = ey[rng.randint(0, ny + 1, k)]
    if n > 10:
        x[-1] = np.nan
        y[-2] = np.nan
    w = rng.uniform(0, 255, n).round()
    explicit = rng.randint(3) == 0
    if explicit:
        bx = np.sort(rng.uniform(lo_x, hi_x, nx + 1))
        by = np.sort(rng.uniform(lo_y, hi_y, ny + 1))
        if np.any(np.diff(bx) <= 0) or np.any(np.diff(by) <= 0):
            explicit = False
    kw = dict(bins=[bx, by]) if explicit else dict(bins=(nx, ny), range=[[lo_x, hi_x], [lo_y, hi_y]])
    got, gx, gy = histogram2d(x, y, weights=[None, w], **kw)
    want, wx, wy = O.histogram2d(x, y, weights=[None, w], **kw)
    if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(gx, wx)
            and np.array_equal(gy, wy)):
        bad += 1
        print('HIST', it, n, nx, ny, explicit, int((got[0] != want[0]).sum()))
    # ---- intersections ----
    m = int(rng.randint(1, 3000))
    a, b = rng.uniform(6300, 6600), rng.uniform(6200, 6550)
    origin = rng.normal(size=3) * rng.choice([3000.0, 6800.0, 20000.0])
    dirs = rng.normal(size=(m, 3))
    dirs[: m // 4] = -origin / np.linalg.norm(origin) + rng.normal(size=(m // 4, 3)) * 0.3       # towards the body
    directed = bool(rng.randint(2))
    if not close(ellipsoidLineIntersection(a, b, origin, dirs, directed), O.ellipsoid_line_intersection(a, b, origin, dirs, directed), 1e-6):
        bad += 1
        print('ELLIPSOID', it, m, directed)
    un = dirs / np.linalg.norm(dirs, axis=1)[:, None]
    if not close(sphereLineIntersection(a, origin, un, directed), O.sphere_line_intersection(a, origin, un, directed), 1e-6):
        bad += 1
        print('SPHERE', it, m, directed)
    # ---- geodetic and frame rotations ----
    p = rng.normal(size=(m, 3)) * 6500
    p[0] = [0.0, 0.0, 6400.0]                                       # on the axis
    la, lo = T.ecef2Geodetic(p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy())
    wla, wlo = O.ecef_to_geodetic(p[:, 0], p[:, 1], p[:, 2])
    if not (close(la, wla, 1e-13) and close(lo, wlo, 1e-13)):
        bad += 1
        print('GEODETIC', it, m)
    t = datetime(int(rng.randint(1990, 2020)), int(rng.randint(1, 13)), int(rng.randint(1, 28)), int(rng.randint(24)),
                 int(rng.randint(60)), int(rng.randint(60)))
    et = O.date2es(t)
    gla, glo = T.j2000ToLatLon(p, t)
    ola, olo = O.j2000_to_latlon(p, O.mat_j2000_to_geo(et))
    gml, gmt = T.j2000ToMLatMLT(p, t)
    oml, omt = O.j2000_to_mlat_mlt(p, O.mat_j2000_to_sm(et))
    if not (close(gla, ola, 1e-10) and close((np.asarray(glo) - olo + 180) % 360 - 180, np.zeros(m), 1e-10)
            and close(gml, oml, 1e-10) and close((np.asarray(gmt) - omt + 12) % 24 - 12, np.zeros(m), 1e-10)):
        bad += 1
        print('ROTATIONS', it, m, t)
print('rounds', rounds, 'failures', bad)
sys.exit(1 if bad else 0)
