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


def angularDistanceOnParallel(lat, dlon):
    """
    ``angularDistance(Location(lat, lon0), Location(lat, lon0 + dlon))`` of the reference
    (geodesic.py:35-44: the ``a12`` of geographiclib's ``Geodesic.WGS84.Inverse``, the arc length in degrees
    on the auxiliary sphere of the shortest geodesic) for two points on ONE parallel, which is all that
    ``plateCarreeResolution`` (resample.py:36-61) asks for.

    geographiclib (pinned 1.34 by the reference) is not available offline, so this is the published
    formulation (C. F. F. Karney, "Algorithms for geodesics", J. Geodesy 87, 2013) restated for the symmetric
    case instead of its truncated series: with reduced latitude beta of both end points and alpha0 the azimuth of
    the geodesic at its node, a point sits at arc sigma_p from the node with sin(beta) = cos(alpha0) sin(sigma_p),
    the vertex at sigma = pi/2 is the mid point, and the longitude difference is (Eq. 8)

        lambda12 = omega12 - f sin(alpha0) * 2 * Int_{sigma_p}^{pi/2} (2 - f) / (1 + (1 - f) sqrt(1 + k^2 sin^2 s)) ds,
        k^2 = e'^2 cos^2(alpha0),  tan(omega) = sin(alpha0) tan(sigma).

    alpha0 is found by bracketing + bisection/secant to machine precision, the integral by 48-point Gauss-Legendre
    quadrature (smooth integrand: converged to 1e-16).  The result agrees with a 6th-order series in f to about
    1e-14 relative; tests/test_host_cpu.py checks it against a direct numerical integration of the geodesic
    equations.  Valid for |dlon| < 180 (mappings are narrower than that, mapping.py:722-737).
    """
    import math
    import numpy as np
    dlon = abs(float(dlon))
    if dlon == 0:
        return 0.0
    assert dlon < 180, 'only for boxes narrower than 180 degrees'
    f = WGS84_f
    ep2 = f * (2 - f) / (1 - f) ** 2                       # second eccentricity squared
    beta = math.atan((1 - f) * math.tan(math.radians(abs(float(lat)))))
    lam = math.radians(dlon)
    sb = math.sin(beta)
    if sb < 1e-12:
        # along the equator the geodesic is the equator itself: lambda12 = (1 - f) omega12 and sigma12 = omega12
        # (up to lambda12 = (1 - f) 180 deg, beyond which the shortest path leaves the equator)
        assert dlon <= 180 * (1 - f)
        return dlon / (1 - f)
    nodes, weights = np.polynomial.legendre.leggauss(48)

    def lam_of(ca0):
        """lambda12 for cos(alpha0) = ca0 in [sin(beta), 1]"""
        sa0 = math.sqrt(max(0.0, 1 - ca0 * ca0))
        s_p = math.asin(min(1.0, sb / ca0)) if ca0 > 0 else math.pi / 2
        om_p = math.atan2(sa0 * math.sin(s_p), math.cos(s_p))
        half, mid = (math.pi / 2 - s_p) / 2, (math.pi / 2 + s_p) / 2
        s = mid + half * nodes
        k2 = ep2 * ca0 * ca0
        integral = half * float(np.sum(weights * (2 - f) / (1 + (1 - f) * np.sqrt(1 + k2 * np.sin(s) ** 2))))
        return 2 * (math.pi / 2 - om_p) - f * sa0 * 2 * integral, math.pi - 2 * s_p

    # lambda12 grows from 0 (vertex at the end points: cos(alpha0) = sin(beta)) as the vertex moves polewards
    lo, hi = max(sb, 1e-300), 1.0
    if lam_of(hi)[0] < lam:                                # beyond the equatorial / meridional limit: not for us
        raise ValueError('no symmetric geodesic for this longitude difference')
    for _ in range(200):
        midc = 0.5 * (lo + hi)
        if lam_of(midc)[0] < lam:
            lo = midc
        else:
            hi = midc
        if hi - lo <= 4e-16 * hi:
            break
    return math.degrees(lam_of(0.5 * (lo + hi))[1])
