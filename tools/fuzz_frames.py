"""Differential fuzz: random frame sizes, pointings, times, shells, resolutions, centre modes, thresholds and image
types — single-pass plan == two-pass plan bit for bit, and the two-pass plan against the oracle (identical masks,
counts differing in at most 2 cells, exact integer means elsewhere).  usage: [BIG=5] fuzz_frames.py [cases] [seed]"""
import os, sys
from datetime import timedelta
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
PADDED = bool(os.environ.get('PADDED'))      # PADDED=1: the pipelines' buffers in strip-padded rows (round 6)
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import frame_header, frame_image, pole_frame
from oracle import ref_numpy as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = skipped = n_single = n_pole = n_box_first = 0
for case in range(cases):
    big = int(os.environ.get('BIG', '1'))                # BIG=5: frames up to 2100 x 1500
    w, h = int(rng.randint(40, 420 * big)), int(rng.randint(30, 300 * big))
    pointing = ('iss030', 'iss029')[rng.randint(2)]
    shift = float(rng.choice([0, 0, 20, 45, 80, 95]))
    alt = float(rng.choice([90, 100, 110, 120, 135]))
    ppd = (float(rng.choice([2, 4, 7, 10, 16, 25])), float(rng.choice([2, 4, 7, 10, 16, 25])))
    fast = bool(rng.randint(2))
    thr = [None, 5.0, 10.0, 20.0][rng.randint(4)]
    dtype = (np.uint8, np.uint16)[rng.randint(2)]
    magnetic = rng.randint(4) == 0                     # MLat/MLT grid: the two plans against each other only
    with_mag = magnetic or rng.randint(3) == 0         # MLat / MLT arrays written beside a geodetic grid as well
    hdr, cam, t = frame_header(w, h, pointing)
    t = t - timedelta(minutes=shift)
    if rng.randint(6) == 0 and not magnetic:           # a camera looking across a pole (the pole plans)
        hdr, cam, t = pole_frame(w, h, south=bool(rng.randint(2)))
        pointing, shift = 'pole', 0.0
    img = frame_image(w, h, seed=case, dtype=dtype)
    tag = '%d: %dx%d %s -%gmin alt %g ppd %s %s thr %s %s%s' % (case, w, h, pointing, shift, alt, ppd,
                                                                 'fast' if fast else 'exact', thr, dtype.__name__,
                                                                 ' magnetic' if magnetic else (' with_mag' if with_mag else ''))
    pipe = FramePipeline(w, h, img_dtype=dtype, with_mag=with_mag, padded=PADDED)
    try:
        two = pipe.run(hdr, alt, cam, t, img=img, fast=fast, min_elevation=thr, pxPerDeg=ppd, fuse=False,
                       magnetic=magnetic)
    except (ValueError, AssertionError) as e:
        skipped += 1                                   # nothing above the threshold / degenerate grid: as the reference
        continue
    arrays_two = pipe.host_arrays() if with_mag else None
    pole_two_pass = bool(two['contains_pole'])
    one = pipe.run(hdr, alt, cam, t, fast=fast, min_elevation=thr, pxPerDeg=ppd, fuse=True, magnetic=magnetic)
    for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
        if not np.array_equal(one[k], two[k], equal_nan=True):
            bad += 1
            print('PLANS DIFFER', tag, k, pipe.last_plan)
            break
    if with_mag:
        arrays_one = pipe.host_arrays()
        for k in ('mlat', 'mlt', 'mlat_c', 'mlt_c', 'lat_c', 'elev'):
            if not np.array_equal(arrays_one[k], arrays_two[k], equal_nan=True):
                bad += 1
                print('ARRAYS DIFFER', tag, k, pipe.last_plan)
                break
    if magnetic:
        # round 4, MLat / MLT-only mode (k_georef_rows<SECOND = 4>): the same grid, and the five arrays it keeps, bit for bit
        lean = FramePipeline(w, h, img_dtype=dtype, with_mag=True, with_geo=False, padded=PADDED)
        five = lean.run(hdr, alt, cam, t, img=img, fast=fast, min_elevation=thr, pxPerDeg=ppd, fuse=True, magnetic=True)
        for k in ('mean', 'count', 'img', 'mask'):
            if not np.array_equal(five[k], two[k], equal_nan=True):
                bad += 1
                print('MLAT/MLT-ONLY GRID DIFFERS', tag, k, lean.last_plan, lean.ctx.last_variant())
                break
        if lean.last_plan == 'single-pass' and not five['contains_pole']:
            kept = lean.host_arrays(kept_only=True)
            for k, v in kept.items():
                if not np.array_equal(v, arrays_two[k], equal_nan=True):
                    bad += 1
                    print('MLAT/MLT-ONLY ARRAYS DIFFER', tag, k)
                    break
        del lean
    if not two['contains_pole'] and rng.randint(2) == 0:
        # round 4, box-first plan: arcsecPerPx -> a box pass, px/deg from the frame's own box, the single-pass launch; against
        # the two-pass plan at that px/deg
        arcsec = float(rng.choice([150, 300, 600, 1200]))
        try:
            bf = pipe.run(hdr, alt, cam, t, fast=fast, min_elevation=thr, arcsecPerPx=arcsec, fuse=True, magnetic=magnetic)
            plan_bf = pipe.last_plan
            tp = pipe.run(hdr, alt, cam, t, fast=fast, min_elevation=thr, pxPerDeg=bf['pxPerDeg'], fuse=False, magnetic=magnetic)
            for k in ('mean', 'count', 'img', 'mask', 'lat', 'lon'):
                if not np.array_equal(bf[k], tp[k], equal_nan=True):
                    bad += 1
                    print('BOX-FIRST DIFFERS', tag, 'arcsec', arcsec, k, plan_bf, bf['pxPerDeg'])
                    break
            n_box_first += plan_bf == 'single-pass'
        except AssertionError:
            pass                                       # (a degenerate grid at that resolution: as the reference)
        pipe.run(hdr, alt, cam, t, fast=fast, min_elevation=thr, pxPerDeg=ppd, fuse=True, magnetic=magnetic)   # (last_plan below)
    n_single += pipe.last_plan == 'single-pass'
    if pipe.last_plan != 'single-pass' and os.environ.get('SHOW_PLANS'):
        print('two-pass:', tag, 'pole' if two['contains_pole'] else '', 'dateline' if two['contains_discontinuity'] else '')
    n_pole += bool(two['contains_pole'])
    if two['contains_pole'] or magnetic:
        continue
    et = O.date2es(t)
    g = O.georef_frame(hdr, alt, cam, O.mat_j2000_to_geo(et), None, fast=fast)
    if thr is None:
        corner_mask, center_mask = O.sanitize_masks(np.isnan(g['lat']), np.isnan(g['lat_c']), after_masking=False)
    else:
        cm0, ce0 = O.sanitize_masks(np.isnan(g['lat']), np.isnan(g['lat_c']), after_masking=False)
        with np.errstate(invalid='ignore'):
            ce = ce0 | ~(g['elev'] >= thr)
        corner_mask, center_mask = O.sanitize_masks(cm0, ce, after_masking=True)
    bbox, disc = O.bbox_of_corners(g['lat'], g['lon'], corner_mask)
    data = np.dstack((img.astype(np.float64), g['elev']))
    data[center_mask] = np.nan
    outline = np.transpose([g['lat'][~corner_mask], g['lon'][~corner_mask]])
    want = O.resample_mean(np.where(center_mask, np.nan, g['lat_c']), np.where(center_mask, np.nan, g['lon_c']), alt, data,
                           outline, bbox, ppd, disc, False)
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
print('cases', cases, 'skipped', skipped, 'single-pass', n_single, 'pole', n_pole, 'box-first single-pass', n_box_first, 'failures', bad)
sys.exit(1 if bad else 0)
