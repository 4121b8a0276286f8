"""Differential fuzz of method='cubic' (round 5: amt_delaunay_* + amt_cubic_gradients_csr + amt_cubic_eval through `_resample`)
against scipy's CloughTocher2DInterpolator with griddata's defaults (tol 1e-6, maxiter 400): random sheared / stretched / gently
warped lattices, smooth and noisy channels, optional holes in the footprint.  The triangulation is Qhull's and the relaxation
runs in scipy's order with scipy's stopping rule, so EVERY filled cell — next to the hull and to holes as well — must agree to
1e-9 of the channel's span (METHOD=linear: the same against scipy's LinearNDInterpolator, on the same triangulation).
usage: fuzz_cubic.py [rounds] [seed]      (METHOD=linear: the same against scipy's LinearNDInterpolator, to 1e-9)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.interpolate, scipy.spatial
from auromat_amd.mapping.mapping import BoundingBox
from auromat_amd.resample import _resample

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
method = os.environ.get('METHOD', 'cubic')
limit = 1e-9
fails = 0
worst = 0.0
compared = cells = 0
for it in range(rounds):
    h, w = int(rng.randint(28, 70)), int(rng.randint(28, 70))
    ii, jj = np.mgrid[0:h, 0:w].astype(np.float64)
    step = rng.uniform(0.02, 0.2)
    shear = rng.uniform(0.15, 0.45) * rng.choice([-1, 1])
    aspect = rng.uniform(0.5, 3.0)
    lat = 30.0 + step * (ii + shear * jj) + 0.3 * step * np.sin(ii / 9.0) * np.cos(jj / 11.0) * rng.uniform(0, 1)
    lon = -20.0 + step * aspect * (jj + 0.1 * ii) + 0.3 * step * np.cos(ii / 7.0) * rng.uniform(0, 1)
    # every row and column gently curved, like a projected camera grid: rows that are straight up to rounding make the
    # hull's triangles a matter of Qhull's roundoff handling (slivers of 1e-16 kept or merged), which nothing reproduces
    bend = rng.uniform(1e-3, 1e-2) * step
    lat = lat + bend * ((jj - w / 2.0) ** 2 / w + 0.5 * (ii - h / 2.0) ** 2 / h)
    lon = lon + bend * aspect * ((ii - h / 2.0) ** 2 / h - 0.3 * (jj - w / 2.0) ** 2 / w)
    data = np.stack([np.sin(lat * rng.uniform(0.5, 3)) * np.cos(lon * rng.uniform(0.5, 3)) * 50,
                     rng.rand(h, w) * 20, 0.3 * lat * lat - lon + 0.1 * lat * lon], axis=2)
    hole = rng.rand() < 0.5
    valid = np.ones((h, w), bool)
    if hole:
        ci, cj, r = rng.randint(8, h - 8), rng.randint(8, w - 8), rng.randint(2, 5)
        valid[(ii - ci) ** 2 + (jj - cj) ** 2 <= r * r] = False
    la, lo, d = lat.copy(), lon.copy(), data.copy()
    la[~valid] = np.nan
    lo[~valid] = np.nan
    d[~valid] = np.nan
    pts = np.column_stack((lat[valid], lon[valid]))
    tri_ = scipy.spatial.Delaunay(pts)
    ref = scipy.interpolate.CloughTocher2DInterpolator(tri_, data[valid]) if method == 'cubic' else \
        scipy.interpolate.LinearNDInterpolator(tri_, data[valid])
    # targets: a grid over the middle of the footprint
    m = 10
    box = BoundingBox(lat[m, m] + step, min(lon[m, m], lon[h - m, m]) + step * aspect, lat[h - m, w - m] - step,
                      max(lon[m, w - m], lon[h - m, w - m]) - step * aspect)
    if not (box.latNorth - box.latSouth > 4 * step and box.lonEast - box.lonWest > 4 * step * aspect):
        continue
    ppd = (round(2.0 / step), round(2.0 / (step * aspect)))
    outline = np.array([[lat[0, 0], lon[0, 0]], [lat[0, -1], lon[0, -1]], [lat[-1, -1], lon[-1, -1]], [lat[-1, 0], lon[-1, 0]]])
    try:
        _, _, lat_c, lon_c, out = _resample(la, lo, 110.0, d, lambda: outline, box, ppd, False, False, method=method)
    except AssertionError as e:
        if 'nLon' in str(e) or 'nLat' in str(e) or 'nlon' in str(e) or 'nlat' in str(e):
            continue
        raise
    q = np.column_stack((np.repeat(lat_c[:, 0], lon_c.shape[1]), np.tile(lon_c[0], lat_c.shape[0])))
    want = ref(q).reshape(out.shape)
    got_ok = ~np.isnan(out[..., 0])
    # distance (in pixels) of each target to the nearest invalid pixel or border, through the nearest pixel centre
    tree = scipy.spatial.cKDTree(np.column_stack((lat.ravel(), lon.ravel())))
    _, near = tree.query(q)
    ni, nj = np.unravel_index(near, (h, w))
    margin = np.minimum.reduce([ni, nj, h - 1 - ni, w - 1 - nj]).astype(np.float64)
    if hole:
        margin = np.minimum(margin, np.hypot(ni - ci, nj - cj) - r)
    deep = (margin.reshape(got_ok.shape) >= 12) & ~np.isnan(want[..., 0])
    span = np.nanmax(want, axis=(0, 1)) - np.nanmin(want, axis=(0, 1))
    bad = False
    if deep.any():
        compared += 1
        cells += int(deep.sum())
        if not got_ok[deep].all():
            print('round', it, 'NaN deep inside the footprint:', int((~got_ok[deep]).sum()), 'cells')
            bad = True
        where = got_ok                                                    # every filled cell
        if np.isnan(want[got_ok]).any():
            print('round', it, 'cells filled outside scipy\'s convex hull:', int(np.isnan(want[got_ok][:, 0]).sum()))
            bad = True
        rel = (np.abs(out - want)[where] / span).max() if where.any() else 0.0
        worst = max(worst, rel)
        if rel > limit:
            print('round', it, 'h w', h, w, 'shear %.2f aspect %.2f hole %s: max relative difference %.2e' % (shear, aspect, hole, rel))
            bad = True
    if not np.isfinite(out[got_ok]).all():
        print('round', it, 'non-finite values')
        bad = True
    fails += bad
print(method, 'rounds', rounds, 'compared', compared, 'cells', cells, 'failures', fails, 'worst relative difference deep inside %.2e' % worst)
sys.exit(1 if fails else 0)
