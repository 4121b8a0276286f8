"""CPU fuzz of amt_georef_image_rows (host function): random cameras — position, look direction from nadir to beyond the limb, roll,
field of view —, shells and thresholds on tall narrow frames (many bands of 16 rows); SOUND = every row that holds a pixel the binning
reads (centre on the shell, elevation >= min_elevation: the oracle's, every pixel, fast and exact centres) lies inside the rows the
function names.  usage: fuzz_image_rows.py [cases] [seed]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datetime import datetime, timedelta
import numpy as np
from auromat_amd import _native
from auromat_amd.coordinates import transform as T
from auromat_amd.mapping.astrometry import frame_params
from oracle import ref_numpy as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = _native.lib()
bad = loose = empty = clipped = 0
for case in range(n):
    w, h = int(rng.choice([24, 40, 63, 64, 90])), int(rng.choice([160, 208, 333]))
    r_cam = 6378.0 + rng.uniform(250, 1500)
    v = rng.standard_normal(3); cam = v / np.linalg.norm(v) * r_cam
    nadir = -cam / r_cam
    # look direction: nadir tilted by 0 .. 100 deg towards a random azimuth
    a = rng.standard_normal(3); a -= a.dot(nadir) * nadir; a /= np.linalg.norm(a)
    tilt = np.deg2rad(rng.uniform(0, 100))
    look = np.cos(tilt) * nadir + np.sin(tilt) * a
    ra, dec = float(np.rad2deg(np.arctan2(look[1], look[0])) % 360), float(np.rad2deg(np.arcsin(look[2])))
    scale = rng.uniform(0.02, 0.6)                      # deg per pixel
    roll = rng.uniform(0, 2 * np.pi)
    cd = scale * np.array([[-np.cos(roll), -np.sin(roll)], [np.sin(roll), -np.cos(roll)]])
    hdr = {'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN', 'LONPOLE': 180.0, 'LATPOLE': 0.0, 'CRVAL1': ra, 'CRVAL2': dec,
           'CRPIX1': w / 2 + 0.5, 'CRPIX2': h / 2 + 0.5, 'CD1_1': cd[0, 0], 'CD1_2': cd[0, 1], 'CD2_1': cd[1, 0], 'CD2_2': cd[1, 1],
           'IMAGEW': w, 'IMAGEH': h}
    t = datetime(2012, 1, 25, 9, 26, 55) + timedelta(seconds=float(rng.uniform(0, 86400 * 300)))
    alt = float(rng.choice([80.0, 110.0, 250.0, 500.0]))
    if r_cam <= 6378.2 + alt:
        continue
    fast = bool(rng.randint(2))
    me = float(rng.choice([-np.inf, 0.0, 2.0, 10.0, 30.0, 55.0, 80.0]))
    p = frame_params(hdr, alt, cam, t, fast, magnetic=False)
    r0, r1 = C.c_int32(-1), C.c_int32(-1)
    assert lib.amt_georef_image_rows(C.byref(p), me, C.byref(r0), C.byref(r1)) == 0
    r0, r1 = r0.value, r1.value
    g = O.georef_frame(hdr, alt, cam, O.mat_j2000_to_geo(T.date2es(t)), None, fast=fast)
    with np.errstate(invalid='ignore'):
        need = (~np.isnan(g['lat_c']) & (g['elev'] >= me)).any(axis=1)
    if need[:r0].any() or need[r1:].any():
        bad += 1
        rows = np.flatnonzero(need)
        print('UNSOUND case %d: %dx%d tilt %.1f scale %.3f alt %.0f min_elev %s fast %s rows [%d, %d) needed %d..%d' % (
            case, w, h, np.rad2deg(tilt), scale, alt, me, fast, r0, r1, rows[0], rows[-1]))
        continue
    if not need.any():
        empty += 1
        continue
    first, last = np.flatnonzero(need)[[0, -1]]
    clipped += (r1 - r0) < h
    if r0 < (first // 16 - 3) * 16 or r1 > (last // 16 + 4) * 16:
        loose += 1
print('cases %d unsound %d, frames without a pixel to bin %d, bands more than three too wide %d, frames clipped %d' % (n, bad, empty, loose, clipped))
sys.exit(1 if bad else 0)
