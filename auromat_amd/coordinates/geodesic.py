"""
WGS84 constants, the ``Location`` tuple and the geodesic helpers of the reference's
auromat/coordinates/geodesic.py.  The reference wraps geographiclib (absent offline); bounding box and pole
detection of whole frames run on the device (``auromat_amd.mapping.mapping.BaseMapping.boundingBox``), these host
routines serve the small-polygon API (``containsOrCrossesPole``, ``BoundingBox.center/size``, ``course``, ...).
"""
import math
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


# ---------------------------------------------------------------------------------------------------------------
# General geodesics on WGS84 (reference geodesic.py:25-202 wraps geographiclib's Geodesic.WGS84.Inverse / Direct, which
# is absent offline).  Vincenty's (1975) iteration for the inverse and direct problems: sub-millimetre distances and
# 1e-9 deg azimuths for everything except nearly antipodal points, which mappings (< 180 deg wide) never produce;
# anchored on the reference's own known answers (boundingbox_test.py:12-50, geodesic_test.py:14-29 and its large
# outline polygons).  Host arithmetic on a handful of points: polygon pole tests, bounding-box centre / size.
# ---------------------------------------------------------------------------------------------------------------
def _wrap180(deg):
    d = math.fmod(deg, 360.0)
    if d < -180.0:
        d += 360.0
    elif d >= 180.0:
        d -= 360.0
    return d


def _inverse(lat1, lon1, lat2, lon2):
    """(s12 [m], azi1 [deg], azi2 [deg], a12 [deg on the auxiliary sphere]) of the shortest geodesic"""
    f, a = WGS84_f, WGS84_a_m
    b = a * (1 - f)
    L = math.radians(_wrap180(lon2 - lon1))
    if L == -math.pi:
        L = math.pi
    U1 = math.atan((1 - f) * math.tan(math.radians(lat1)))
    U2 = math.atan((1 - f) * math.tan(math.radians(lat2)))
    sU1, cU1, sU2, cU2 = math.sin(U1), math.cos(U1), math.sin(U2), math.cos(U2)
    if abs(lat1) == 90:
        cU1 = 0.0
    if abs(lat2) == 90:
        cU2 = 0.0
    lam = L
    for _ in range(1000):
        sl, cl = math.sin(lam), math.cos(lam)
        t1, t2 = cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl
        ss = math.hypot(t1, t2)
        if ss == 0:
            return 0.0, 0.0, 0.0, 0.0                    # coincident points
        cs = sU1 * sU2 + cU1 * cU2 * cl
        sigma = math.atan2(ss, cs)
        sa = cU1 * cU2 * sl / ss
        c2a = 1 - sa * sa
        c2sm = cs - 2 * sU1 * sU2 / c2a if c2a > 1e-300 else 0.0
        C = f / 16 * c2a * (4 + f * (4 - 3 * c2a))
        new = L + (1 - C) * f * sa * (sigma + C * ss * (c2sm + C * cs * (-1 + 2 * c2sm * c2sm)))
        done = abs(new - lam) < 1e-14
        lam = new
        if done:
            break
    else:
        raise ValueError('geodesic inverse did not converge (nearly antipodal points)')
    sl, cl = math.sin(lam), math.cos(lam)
    u2 = c2a * (a * a - b * b) / (b * b)
    A = 1 + u2 / 16384 * (4096 + u2 * (-768 + u2 * (320 - 175 * u2)))
    B = u2 / 1024 * (256 + u2 * (-128 + u2 * (74 - 47 * u2)))
    ds = B * ss * (c2sm + B / 4 * (cs * (-1 + 2 * c2sm * c2sm) -
                                   B / 6 * c2sm * (-3 + 4 * ss * ss) * (-3 + 4 * c2sm * c2sm)))
    s12 = b * A * (sigma - ds)
    azi1 = math.degrees(math.atan2(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl))
    azi2 = math.degrees(math.atan2(cU1 * sl, -sU1 * cU2 + cU1 * sU2 * cl))
    return s12, azi1, azi2, math.degrees(sigma)


def _direct(lat1, lon1, azi1, s12):
    """(lat2, lon2, azi2) in degrees when travelling s12 metres from (lat1, lon1) at azimuth azi1"""
    f, a = WGS84_f, WGS84_a_m
    b = a * (1 - f)
    al = math.radians(azi1)
    sa1, ca1 = math.sin(al), math.cos(al)
    tU1 = (1 - f) * math.tan(math.radians(lat1))
    cU1 = 1 / math.sqrt(1 + tU1 * tU1)
    sU1 = tU1 * cU1
    sigma1 = math.atan2(tU1, ca1)
    sa = cU1 * sa1
    c2a = 1 - sa * sa
    u2 = c2a * (a * a - b * b) / (b * b)
    A = 1 + u2 / 16384 * (4096 + u2 * (-768 + u2 * (320 - 175 * u2)))
    B = u2 / 1024 * (256 + u2 * (-128 + u2 * (74 - 47 * u2)))
    sigma = s12 / (b * A)
    for _ in range(1000):
        c2sm = math.cos(2 * sigma1 + sigma)
        ss, cs = math.sin(sigma), math.cos(sigma)
        ds = B * ss * (c2sm + B / 4 * (cs * (-1 + 2 * c2sm * c2sm) -
                                       B / 6 * c2sm * (-3 + 4 * ss * ss) * (-3 + 4 * c2sm * c2sm)))
        new = s12 / (b * A) + ds
        done = abs(new - sigma) < 1e-15
        sigma = new
        if done:
            break
    c2sm = math.cos(2 * sigma1 + sigma)
    ss, cs = math.sin(sigma), math.cos(sigma)
    t = sU1 * ss - cU1 * cs * ca1
    lat2 = math.atan2(sU1 * cs + cU1 * ss * ca1, (1 - f) * math.hypot(sa, t))
    lam = math.atan2(ss * sa1, cU1 * cs - sU1 * ss * ca1)
    C = f / 16 * c2a * (4 + f * (4 - 3 * c2a))
    L = lam - (1 - C) * f * sa * (sigma + C * ss * (c2sm + C * cs * (-1 + 2 * c2sm * c2sm)))
    azi2 = math.degrees(math.atan2(sa, -t))
    return math.degrees(lat2), _wrap180(lon1 + math.degrees(L)), azi2


def distance(location1, location2):
    """Return the shortest distance in meters between two locations (reference geodesic.py:25-33)."""
    return _inverse(location1.lat, location1.lon, location2.lat, location2.lon)[0]


def angularDistance(location1, location2):
    """Shortest angular distance in degrees on an auxiliary sphere between two locations (geodesic.py:35-44)."""
    return _inverse(location1.lat, location1.lon, location2.lat, location2.lon)[3]


def course(location1, location2):
    """Return the azimuth in degrees when travelling from `location1` to `location2` (geodesic.py:111-119)."""
    return _inverse(location1.lat, location1.lon, location2.lat, location2.lon)[1]


def destination(location, azimuth, distance):
    """Location reached from `location` in direction `azimuth` [deg] after `distance` meters (geodesic.py:80-91)."""
    lat, lon, _ = _direct(location.lat, location.lon, azimuth, distance)
    return Location(lat, lon)


def intermediate(location1, location2, f=0.5):
    """Location at the fraction `f` of the way from `location1` to `location2` (geodesic.py:93-109)."""
    s12, azi1, _, _ = _inverse(location1.lat, location1.lon, location2.lat, location2.lon)
    lat, lon, _ = _direct(location1.lat, location1.lon, azi1, s12 * f)
    return Location(lat, lon)


def line(location1, location2, resolution=1000):
    """
    Points on the geodesic between the locations every `resolution` meters, (n,2) [lat,lon] in degrees; the two
    end points alone when the line is shorter than two steps (geodesic.py:46-78).
    """
    import numpy as np
    s12, azi1, _, _ = _inverse(location1.lat, location1.lon, location2.lat, location2.lon)
    num = s12 // resolution
    if num < 2:
        return np.array([[location1.lat, location1.lon], [location2.lat, location2.lon]])
    return np.array([_direct(location1.lat, location1.lon, azi1, d)[:2] for d in np.linspace(0, s12, int(num))])


def _courseDelta(a1, a2):
    """left-turn amount between two courses in degrees, in (-180, 180] with 180 -> 0 (geodesic.py:121-137)"""
    if a2 < a1:
        a2 += 360
    left_turn_amount = a2 - a1
    if left_turn_amount == 180:
        return 0
    elif left_turn_amount > 180:
        return left_turn_amount - 360
    return left_turn_amount


def _courseDeltaSum(points):
    """
    Sum of the course changes along an unclosed, non-self-intersecting polygon of (lat, lon) points in degrees:
    -360, -180, 0, 180 or 360 (geodesic.py:139-184).
    """
    import numpy as np
    points = np.asarray(points, dtype=np.float64)
    assert points.ndim == 2 and points.shape[1] == 2
    points = np.concatenate((points, [points[0]]))
    arcs = len(points) - 1
    courses = np.empty(arcs * 2)
    for i in range(arcs):
        lat1, lon1 = points[i]
        lat2, lon2 = points[i + 1]
        _, azi1, azi2, _ = _inverse(lat1, lon1, lat2, lon2)
        courses[2 * i] = azi1
        # the reference asks for the course of the reversed arc and adds 180: the arrival course of this arc
        courses[2 * i + 1] = _inverse(lat2, lon2, lat1, lon1)[1] + 180
    deltas = np.empty(arcs * 2)
    deltas[0] = _courseDelta(courses[arcs * 2 - 1], courses[0])
    for i in range(1, arcs * 2):
        deltas[i] = _courseDelta(courses[i - 1], courses[i])
    deltaSum = np.around(np.sum(deltas), decimals=1)
    assert deltaSum in [-360, -180, 0, 180, 360], deltaSum
    return deltaSum


def containsOrCrossesPole(points):
    """
    Return whether the given polygon contains or crosses one of the poles (geodesic.py:186-202).

    :param points: ordered points forming a non-intersecting unclosed polygon
    :type points: ndarray of shape (n,2) with lat,lon coordinates in degrees
    :rtype: bool
    """
    return abs(_courseDeltaSum(points)) != 360
