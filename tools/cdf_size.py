"""Size and write / read-back time of one frame exported as a version-3 CDF (auromat_amd/export/cdf.py over _cdf3.py): the
reference's own test frame (4256 x 2832) unresampled with pixel bounds and MLat/MLT — what `auromat-convert --format cdf` stores
by default —, the same without bounds / MLat/MLT, and its resampled grid; every file read back with the package's reader
(CDFMapping) and compared.  A compressed CDF variable is ONE gzip stream per record (level 5, pycdf's default): the nine
96-MB arrays of a frame deflate side by side on the writer's threads, each on one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from auromat_amd.export.cdf import write
from auromat_amd.mapping.cdf import read_arrays
from auromat_amd.mapping.spacecraft import getMapping
from auromat_amd.resample import resample
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'resources')
img, wcs = os.path.join(R, 'ISS030-E-102170_dc.jpg'), os.path.join(R, 'ISS030-E-102170_dc.wcs')
out = os.environ.get('TMPDIR', '/tmp')
m = getMapping(img, wcs, altitude=110, fastCenterCalculation=True)
cases = [('unresampled, bounds + MLat/MLT', m, {}), ('unresampled, --without-bounds --without-mag', m, dict(includeBounds=False, includeMagCoords=False)),
         ('unresampled, bounds + MLat/MLT, uncompressed', m, dict(compress=False)),
         ('resampled to 0.1 deg (maskedByElevation(10)), bounds + MLat/MLT', resample(m.maskedByElevation(10), pxPerDeg=10), {})]
for name, mp, kw in cases:
    p = os.path.join(out, 'size_probe.cdf')
    mp.lats, mp.mLatMlt, mp.mLatMltCenter, mp.elevation, mp.img          # (the arrays come to the host before the clock starts)
    t0 = time.time()
    write(p, mp, **kw)
    tw = time.time() - t0
    size = os.path.getsize(p)
    same, tr = 'not read (no bounds)', 0.0
    if kw.get('includeBounds', True):
        t0 = time.time()
        a = read_arrays(p)
        tr = time.time() - t0
        same = all(np.array_equal(a[k].filled(np.nan), getattr(mp, k).filled(np.nan), equal_nan=True) for k in ('lats', 'lons', 'latsCenter', 'lonsCenter'))
        same = same and np.array_equal(a['img'].filled(0), mp.img.filled(0))
    print('%-64s CDF %8.1f MB (written in %5.1f s)  read back equal: %s (%.1f s)' % (name, size / 1e6, tw, same, tr))
    os.remove(p)
