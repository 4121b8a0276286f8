"""
WGS84 constants and the ``Location`` tuple (reference auromat/coordinates/geodesic.py:20-23).
The geodesic distance/azimuth helpers of the reference module wrap geographiclib and feed
bounding-box / pole detection; that step is done on the device here (see
``auromat_amd.mapping.mapping.BaseMapping.boundingBox``), so they are not part of this package.
"""
from collections import namedtuple

# geographiclib.constants.Constants.WGS84_a / WGS84_f (geographiclib 1.34, reference requirements.txt:9)
WGS84_a_m = 6378137.0
WGS84_f = 1 / 298.257223563

wgs84A = WGS84_a_m / 1000
wgs84B = wgs84A * (1 - WGS84_f)

Location = namedtuple('Location', ['lat', 'lon'])  # in degrees
