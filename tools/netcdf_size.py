"""File size of one exported frame (VERDICT r2 missing #2): the reference writes NETCDF4 with zlib and
(1, w)-row chunks (auromat/export/netcdf.py:48,128-326; its user guide quotes 373 MB per frame).  This package writes the
same container (auromat_amd/export/_nc4.py, the default) or netCDF classic (CDF-2, no compression).  Writes the reference's own
test frame (4256 x 2832, unresampled, with pixel bounds and MLat/MLT: what `auromat-convert --format netcdf` stores by
default), the same without bounds / MLat/MLT, and its resampled grid, in both containers; reads the netCDF-4 file back with this
package's reader and compares."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from auromat_amd.export import _nc4
from auromat_amd.export.netcdf import write
from auromat_amd.mapping.spacecraft import getMapping
from auromat_amd.resample import resample
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'resources')
img, wcs = os.path.join(R, 'ISS030-E-102170_dc.jpg'), os.path.join(R, 'ISS030-E-102170_dc.wcs')
out = os.environ.get('TMPDIR', '/tmp')
m = getMapping(img, wcs, altitude=110, fastCenterCalculation=True)
cases = [('unresampled, bounds + MLat/MLT', m, {}), ('unresampled, --without-bounds --without-mag', m, dict(includeBounds=False, includeMagCoords=False)),
         ('resampled to 0.1 deg (maskedByElevation(10)), bounds + MLat/MLT', resample(m.maskedByElevation(10), pxPerDeg=10), {})]
for name, mp, kw in cases:
    p4, p3 = os.path.join(out, 'size_probe4.nc'), os.path.join(out, 'size_probe3.nc')
    t0 = time.time()
    write(p4, mp, **kw)
    t4 = time.time() - t0
    t0 = time.time()
    write(p3, mp, format='NETCDF3_64BIT', **kw)
    t3 = time.time() - t0
    s4, s3 = os.path.getsize(p4), os.path.getsize(p3)
    t0 = time.time()
    f4, f3 = _nc4.open_file(p4), _nc4.open_file(p3)
    tr = time.time() - t0
    same = all(np.array_equal(f4.vars[k].data, v.data, equal_nan=True) for k, v in f3.vars.items() if k not in ('crs', 'mcrs'))
    print('%-64s netCDF-4 %8.1f MB (written in %5.1f s)  classic %8.1f MB (%4.1f s)  ratio %.2f  read back equal: %s (%.1f s)' % (
        name, s4 / 1e6, t4, s3 / 1e6, t3, s4 / s3, same, tr))
    os.remove(p4)
    os.remove(p3)
