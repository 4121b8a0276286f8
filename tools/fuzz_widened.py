"""Fuzz of the widened paths: traced outline vs the oracle's marching squares on random masks, point-in-polygon vs
matplotlib on random (also self-intersecting) polygons with on-vertex / on-edge points, nearest-neighbour grid search
vs brute force on random point clouds.  usage: fuzz_widened.py [rounds] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import matplotlib.path
from auromat_amd._native import Context, ptr, to_host
from auromat_amd.resample import _Grid, nearest_indices
from auromat_amd.utils import outline
from oracle import ref_numpy as O

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context.current()
bad = 0


def same_polygon(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    return any(np.array_equal(np.roll(b, -s, axis=0), a) for s in np.nonzero(np.all(b == a[0], axis=1))[0])


for it in range(rounds):
    # --- outline ---
    h, w = int(rng.randint(3, 70)), int(rng.randint(3, 70))
    yy, xx = np.mgrid[0:h, 0:w]
    im = ((yy - h * rng.uniform(0.3, 0.7)) / (h * rng.uniform(0.2, 0.6))) ** 2 + \
         ((xx - w * rng.uniform(0.3, 0.7)) / (w * rng.uniform(0.2, 0.6))) ** 2 <= 1
    im &= rng.rand(h, w) > rng.uniform(0, 0.25)
    im |= rng.rand(h, w) > rng.uniform(0.9, 1.0)
    if im.sum() == 0:
        im[h // 2, w // 2] = True
    try:
        want = O.outline(im)
    except (ValueError, IndexError):
        want = None                                    # only degenerate contours: the reference fails as well
    try:
        got = outline(im)
    except ValueError:
        got = None
    tie = want is not None and got is not None and not same_polygon(got, want) and \
        O.polygon_area(want) == O.polygon_area(got)        # several contours of the same (largest) area: any is right
    if not tie and ((want is None) != (got is None) or (want is not None and not same_polygon(got, want))):
        bad += 1
        print('OUTLINE', it, h, w)
        print(im.astype(int).tolist())
        print('oracle', None if want is None else want.tolist())
        print('device', None if got is None else got.tolist())
    # --- point in polygon ---
    m = int(rng.randint(3, 40))
    if rng.randint(2):
        ang = np.sort(rng.uniform(0, 2 * np.pi, m))
        poly = np.transpose([rng.uniform(1, 5, m) * np.cos(ang), rng.uniform(1, 5, m) * np.sin(ang)])
    else:
        poly = rng.randint(-5, 6, (m, 2)).astype(np.float64)            # integer, usually self-intersecting
    pts = np.concatenate((rng.uniform(-6, 6, (2000, 2)), rng.randint(-6, 7, (500, 2)).astype(np.float64), poly,
                          (poly + np.roll(poly, 1, axis=0)) / 2))
    want = matplotlib.path.Path(poly).contains_points(pts)
    out = ctx.empty((len(pts),), torch.uint8)
    px, py, pg = ctx.to_device(np.ascontiguousarray(pts[:, 0])), ctx.to_device(np.ascontiguousarray(pts[:, 1])), \
        ctx.to_device(poly)                             # (named: the tensors must outlive the call)
    ctx.call('amt_points_in_polygon', ptr(px), ptr(py), len(pts), ptr(pg), len(poly), ptr(out))
    if not np.array_equal(to_host(out).astype(bool), want):
        bad += 1
        print('PIP', it, m, int((to_host(out).astype(bool) != want).sum()))
    # --- nearest ---
    hh, ww = int(rng.randint(2, 40)), int(rng.randint(2, 50))
    lat = rng.uniform(40, 48, (hh, ww))
    lon = rng.uniform(9, 22, (hh, ww))
    lat[rng.rand(hh, ww) < rng.uniform(0, 0.5)] = np.nan
    lon[np.isnan(lat)] = np.nan
    ppd = (float(rng.choice([1, 2, 5, 10])), float(rng.choice([1, 2, 5, 10])))
    grid = _Grid(ppd, 41.3, 46.8, 10.2, 20.9)
    dlat, dlon = ctx.to_device(lat), ctx.to_device(lon)
    idx = to_host(nearest_indices(ctx, dlat, dlon, None, None, hh, ww, None, grid, 0, None), dtype=np.int64)
    src = np.nonzero(~np.isnan(lat.ravel()))[0]
    if len(src) == 0:
        ok = (idx == -1).all()
    else:
        d = (lat.ravel()[src][None, None, :] - grid.latCenters[:, None, None]) ** 2 + \
            (lon.ravel()[src][None, None, :] - grid.lonCenters[None, :, None]) ** 2
        ok = np.array_equal(idx, src[np.argmin(d, axis=2)])
    if not ok:
        bad += 1
        print('NEAREST', it, hh, ww, ppd)
    # --- all-sky camera model and altitude reprojection vs the oracle ---
    if it % 10 == 0:
        from datetime import datetime
        from auromat_amd.mapping.miracle import CalibrationData, MIRACLEMapping
        from auromat_amd.mapping.mapping import BoundingBox
        from auromat_amd.mapping.themis import reproject
        n = int(rng.randint(4, 90))
        cal = dict(lat=float(rng.uniform(-80, 80)), lon=float(rng.uniform(-179, 179)), xc=float(rng.uniform(200, 300)),
                   yc=float(rng.uniform(200, 300)), k=float(rng.uniform(120, 200)), rotation=float(rng.uniform(-3, 3)))
        alt = float(rng.uniform(80, 300))
        cd = CalibrationData(station='XXX', validFrom=None, validTo=None, boundingBoxSimple=BoundingBox(0, 0, 1, 1), **cal)
        mp = MIRACLEMapping(cd, np.zeros((n, n, 3), np.uint8), datetime(2012, 3, 4, 17, 19), alt)
        g = O.allsky_georef(n, cal, alt)
        err = max(np.nanmax(np.abs(mp.lats.data - g['lat'])), np.nanmax(np.abs(mp.latsCenter.data - g['lat_c'])),
                  np.nanmax(np.abs(mp.elevation.data - g['elev'])),
                  np.nanmax(np.abs((mp.lons.data - g['lon'] + 180) % 360 - 180)))
        la, lo = reproject((cal['lat'], cal['lon']), g['lat'], g['lon'], alt, alt * 1.4)
        wla, wlo = O.themis_reproject((cal['lat'], cal['lon']), g['lat'], g['lon'], alt, alt * 1.4)
        ok = np.array_equal(np.isnan(la), np.isnan(wla))
        fin = ~np.isnan(la)
        err2 = max(np.max(np.abs(la[fin] - wla[fin]), initial=0), np.max(np.abs((lo[fin] - wlo[fin] + 180) % 360 - 180), initial=0))
        if not ok or err > 1e-9 or err2 > 1e-9:
            bad += 1
            print('ALLSKY', it, n, cal, alt, err, err2, ok)
print('rounds', rounds, 'failures', bad)
sys.exit(1 if bad else 0)
