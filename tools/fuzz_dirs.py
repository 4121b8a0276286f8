"""Differential fuzz of the directions-in plans (round 5: amt_pipe_coarse_dirs / amt_pipe_launch_dirs): corner direction arrays of
random frames — the TAN model's own directions, and the same bent by a radial lens distortion and a roll, which no camera model
of the library describes — through the single-pass plan and the two-pass plan (bit for bit), and the two-pass plan against the
oracle's chain on those very directions (intersection, mean-of-corners centres, elevation, masks, box, binning: identical masks,
counts differing in at most 2 cells, exact integer means elsewhere).  usage: [BIG=5] fuzz_dirs.py [cases] [seed]"""
import os, sys
from datetime import timedelta
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
PADDED = bool(os.environ.get('PADDED'))      # PADDED=1: the pipelines' buffers in strip-padded rows (round 6)
from auromat_amd.mapping.astrometry import frame_params, pixelDirection
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import frame_header, frame_image, pole_frame
from oracle import ref_numpy as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def bend(dirs, k, roll_deg):
    """radial distortion about the optical axis (the mean direction) and a roll about it"""
    axis = dirs.reshape(-1, 3).mean(axis=0)
    axis /= np.linalg.norm(axis)
    along = dirs.dot(axis)[..., None] * axis
    perp = dirs - along
    r2 = (perp ** 2).sum(axis=-1, keepdims=True)
    out = dirs + k * r2 * perp
    th = np.deg2rad(roll_deg)
    # Rodrigues' rotation about the axis
    out = out * np.cos(th) + np.cross(axis, out) * np.sin(th) + axis * out.dot(axis)[..., None] * (1 - np.cos(th))
    return out / np.linalg.norm(out, axis=-1, keepdims=True)


def oracle_grid(dir_c, cam, alt, t, img, thr, ppd):
    et = O.date2es(t)
    m_geo = O.mat_j2000_to_geo(et)
    p_c = O.inflated_earth_intersection(dir_c.reshape(-1, 3), np.asarray(cam, float), alt).reshape(dir_c.shape)
    dir_m = O.calc_centers(dir_c)
    with np.errstate(invalid='ignore'):
        p_m = O.calc_centers(p_c)
    lat, lon = (a.reshape(p_c.shape[:2]) for a in O.j2000_to_latlon(p_c.reshape(-1, 3), m_geo))
    latc, lonc = (a.reshape(p_m.shape[:2]) for a in O.j2000_to_latlon(p_m.reshape(-1, 3), m_geo))
    elev = O.elevation_deg(dir_m, p_m)
    cm0, ce0 = O.sanitize_masks(np.isnan(lat), np.isnan(latc), after_masking=False)
    if thr is None:
        corner_mask, center_mask = cm0, ce0
    else:
        with np.errstate(invalid='ignore'):
            ce = ce0 | ~(elev >= thr)
        corner_mask, center_mask = O.sanitize_masks(cm0, ce, after_masking=True)
    if corner_mask.all():
        return None
    bbox, disc = O.bbox_of_corners(lat, lon, corner_mask)
    data = np.dstack((img.astype(np.float64), elev))
    data[center_mask] = np.nan
    outline = np.transpose([lat[~corner_mask], lon[~corner_mask]])
    return O.resample_mean(np.where(center_mask, np.nan, latc), np.where(center_mask, np.nan, lonc), alt, data, outline, bbox, ppd,
                           disc, False)


bad = skipped = n_single = n_pole = n_bent = 0
for case in range(cases):
    big = int(os.environ.get('BIG', '1'))
    w, h = int(rng.randint(40, 420 * big)), int(rng.randint(30, 300 * big))
    pointing = ('iss030', 'iss029')[rng.randint(2)]
    shift = float(rng.choice([0, 0, 20, 45, 80, 95]))
    alt = float(rng.choice([90, 100, 110, 120, 135]))
    ppd = (float(rng.choice([2, 4, 7, 10, 16, 25])), float(rng.choice([2, 4, 7, 10, 16, 25])))
    thr = [None, 5.0, 10.0, 20.0][rng.randint(4)]
    dtype = (np.uint8, np.uint16)[rng.randint(2)]
    magnetic = rng.randint(4) == 0
    hdr, cam, t = frame_header(w, h, pointing)
    t = t - timedelta(minutes=shift)
    if rng.randint(6) == 0 and not magnetic:
        hdr, cam, t = pole_frame(w, h, south=bool(rng.randint(2)))
        pointing, shift = 'pole', 0.0
    k, roll = 0.0, 0.0
    if rng.randint(3) > 0:
        k, roll = float(rng.uniform(-1.5, 1.5)), float(rng.choice([0.0, 0.4, -3.0, 25.0]))
        n_bent += 1
    img = frame_image(w, h, seed=case, dtype=dtype)
    dirs = np.ascontiguousarray(bend(pixelDirection(hdr, corner=True), k, roll))
    tag = '%d: %dx%d %s -%gmin alt %g ppd %s thr %s %s k %.2f roll %g%s' % (case, w, h, pointing, shift, alt, ppd, thr, dtype.__name__,
                                                                        k, roll, ' magnetic' if magnetic else '')
    pipe = FramePipeline(w, h, img_dtype=dtype, with_mag=magnetic, padded=PADDED)
    dev = pipe.ctx.to_device(dirs)
    p = frame_params(hdr, alt, cam, t, True, magnetic=magnetic)
    try:
        two = pipe.run(None, alt, cam, t, img=img, fast=True, min_elevation=thr, pxPerDeg=ppd, fuse=False, magnetic=magnetic,
                       params=p, dirs=dev)
    except (ValueError, AssertionError):
        skipped += 1
        continue
    one = pipe.run(None, alt, cam, t, fast=True, min_elevation=thr, pxPerDeg=ppd, fuse=True, magnetic=magnetic, params=p, dirs=dev)
    n_single += pipe.last_plan == 'single-pass'
    n_pole += bool(two['contains_pole'])
    for key in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
        if not np.array_equal(one[key], two[key], equal_nan=True):
            bad += 1
            print('PLANS DIFFER', tag, key, pipe.last_plan)
            break
    if two['contains_pole'] or magnetic:
        continue
    want = oracle_grid(dirs, cam, alt, t, img, thr, ppd)
    if want is None:
        continue
    if want['data'].shape[:2] != two['count'].shape:
        bad += 1
        print('GRID DIFFERS', tag, want['data'].shape, two['count'].shape)
        continue
    ndiff = int((want['count'] != two['count']).sum())
    same = (want['count'] == two['count']) & (want['count'] > 0)
    imgdiff = int((two['mean'][..., :3][same] != want['data'][..., :3][same]).sum())
    if ndiff > 2 or imgdiff:
        bad += 1
        print('ORACLE DIFFERS', tag, 'cells', ndiff, 'means', imgdiff, want['count'].sum(), two['count'].sum())
print('cases', cases, 'skipped', skipped, 'bent', n_bent, 'single-pass', n_single, 'pole', n_pole, 'failures', bad)
sys.exit(1 if bad else 0)
