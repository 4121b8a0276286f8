"""Where the time of export.netcdf.write() of one unresampled full-size frame goes (cProfile, top entries)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auromat_amd.export.netcdf import write
from auromat_amd.mapping.spacecraft import getMapping
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'resources')
m = getMapping(os.path.join(R, 'ISS030-E-102170_dc.jpg'), os.path.join(R, 'ISS030-E-102170_dc.wcs'), altitude=110, fastCenterCalculation=True)
p = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'prof_probe.nc')
t0 = time.time(); write(p, m); print('first write %.2f s, %.1f MB' % (time.time() - t0, os.path.getsize(p) / 1e6))
pr = cProfile.Profile(); pr.enable(); t0 = time.time(); write(p, m); el = time.time() - t0; pr.disable()
print('second write %.2f s' % el)
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
os.remove(p)
