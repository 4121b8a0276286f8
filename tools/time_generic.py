"""Wall time of the generic (array-level) entry points at 12 Mpx: util.histogram.histogram2d with weights, the
array-level _resample(method='mean'), ellipsoidLineIntersection, ecef2Geodetic — including their host <-> device copies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd.coordinates.intersection import ellipsoidLineIntersection
from auromat_amd.coordinates.transform import ecef2Geodetic
from auromat_amd.mapping.mapping import BoundingBox
from auromat_amd.resample import _resample
from auromat_amd.util.histogram import histogram2d

n = 4240 * 2832
rng = np.random.RandomState(0)
x, y = rng.uniform(-110, -90, n), rng.uniform(48, 61, n)
w = rng.uniform(0, 1, n)


def timed(label, fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print('%-58s %8.1f ms' % (label, (time.perf_counter() - t0) / reps * 1e3))


timed('histogram2d 12M points, 200x130 bins, count only', lambda: histogram2d(x, y, bins=(200, 130)))
timed('histogram2d 12M points, 200x130 bins, 1 weight', lambda: histogram2d(x, y, bins=(200, 130), weights=[None, w]))
lat_c, lon_c = y.reshape(2832, 4240), x.reshape(2832, 4240)
data = np.dstack([w.reshape(2832, 4240)] * 4)
timed('_resample array level, 4 channels, 0.1 deg', lambda: _resample(lat_c, lon_c, 110, data, None, BoundingBox(48, -110, 61, -90), (10, 10)))
dirs = rng.normal(size=(n, 3))
dirs /= np.linalg.norm(dirs, axis=1)[:, None]
timed('ellipsoidLineIntersection 12M rays', lambda: ellipsoidLineIntersection(6488.0, 6466.0, [7000.0, 0.0, 0.0], dirs))
p = rng.normal(size=(3, n)) * 6000
timed('ecef2Geodetic 12M points', lambda: ecef2Geodetic(p[0], p[1], p[2]))
