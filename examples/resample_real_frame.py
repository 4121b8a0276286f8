"""
The reference's user-guide flow on its own test frame (needs an MI355X):

    python examples/resample_real_frame.py [out.nc]

Image file + astrometry.net .wcs file -> mapping on the 110 km shell -> mask below 10 deg elevation -> resample to a
0.1 deg geographic grid and to an MLat/MLT grid -> netCDF (CF-1.6) file of the geographic one.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from auromat_amd.export.netcdf import write                      # auromat.export.netcdf
from auromat_amd.mapping.spacecraft import getMapping            # auromat.mapping.spacecraft
from auromat_amd.resample import resample, resampleMLatMLT       # auromat.resample

res = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'resources')
m = getMapping(os.path.join(res, 'ISS030-E-102170_dc.jpg'), os.path.join(res, 'ISS030-E-102170_dc.wcs'),
               altitude=110, fastCenterCalculation=True)
print('photo time', m.photoTime, 'camera [km, GCRS]', m.cameraPosGCRS)
m = m.maskedByElevation(10)
bb = m.boundingBox
print('footprint: lat %.2f .. %.2f, lon %.2f .. %.2f' % (bb.latSouth, bb.latNorth, bb.lonWest, bb.lonEast))
geo = resample(m, pxPerDeg=10)
mag = resampleMLatMLT(m, pxPerDeg=10)
print('geographic grid', geo.img.shape, 'valid cells', int((~geo.img.mask[..., 0]).sum()), '| MLat/MLT grid', mag.img.shape)
out = sys.argv[1] if len(sys.argv) > 1 else 'ISS030-E-102170_dc.nc'
write(out, geo)
print('wrote', out)
