"""
TEST INFRASTRUCTURE — CPU oracle for the georeferencing + resampling hot path.

A plain-NumPy restatement of esa/auromat's per-pixel path, kept
in reference operation order so that it reproduces the reference's own NumPy
(`_np`) results bit for bit wherever NumPy itself is deterministic.  Every
function cites the reference lines it follows (paths relative to
/root/reference/).

Who may import this file: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — as the checker / reported baseline only.
The product (``auromat_amd``) never imports it and has no CPU fallback.

Parity pin: ``oracle/make_golden.py`` runs the *real* reference (imported from
/root/reference through ``oracle/refshim.py``) and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those vectors and
against the reference's own known-answer tests (intersection_test.py:26-137,
transform_test.py:70-129), stored as data in ``tests/golden/known_answers.json``.

Third-party arithmetic on the path that is *not* in /root/reference:
``astropy.time.Time(...).jd`` (astropy >= 0.4.1, setup.py:49; used once per
frame at transform.py:529).  ``date2es`` below restates it as a single-double
Julian date; all fixtures record the resulting ``et`` so no parity result
depends on that restatement.
"""
from __future__ import division

import math
from datetime import datetime

import numpy as np

# ---------------------------------------------------------------------------
# constants (geodesic.py:20-21 with geographiclib 1.34 Constants)
# ---------------------------------------------------------------------------
WGS84_A = 6378137.0 / 1000
WGS84_B = WGS84_A * (1 - 1 / 298.257223563)

# igrf.py:25-37 — IGRF g01/g11/h11 [nT], 1900..2020 in 5-year steps (published model coefficients)
IGRF_G01 = [-31543, -31464, -31354, -31212, -31060, -30926, -30805, -30715,
            -30654, -30594, -30554, -30500, -30421, -30334, -30220, -30100,
            -29992, -29873, -29775, -29692, -29619.4, -29554.63, -29496.5,
            -29442, -29390.5]
IGRF_G11 = [-2298, -2298, -2297, -2306, -2317, -2318, -2316, -2306, -2292, -2285,
            -2250, -2215, -2169, -2119, -2068, -2013, -1956, -1905, -1848, -1784,
            -1728.2, -1669.05, -1585.9, -1501, -1410.5]
IGRF_H11 = [5922, 5909, 5898, 5875, 5845, 5817, 5808, 5812, 5821, 5810, 5815,
            5820, 5791, 5776, 5737, 5675, 5604, 5500, 5406, 5306, 5186.1, 5077.99,
            4944.26, 4797.1, 4664.1]


# ---------------------------------------------------------------------------
# host scalars: time, IGRF dipole, frame rotation matrices
# ---------------------------------------------------------------------------
def date2es(date):
    """transform.py:525-530 (UTC datetime -> ephemeris seconds past J2000) with a single-double JD."""
    delta = date - datetime(2000, 1, 1, 12)
    jd = 2451545.0 + (delta.days + (delta.seconds + delta.microseconds / 1e6) / 86400.0)
    return (jd - 2451545) * 86400


def _igrf(table, et):
    """igrf.py:40-58: linear interpolation between 5-year epochs; raises past the last epoch."""
    idx = (et + 3155803200.0) / 157788000.0
    frac = math.fmod(idx, 1.0)
    if idx >= len(table) - 1:
        raise ValueError('IGRF coefficients are not defined for this date')
    return table[int(math.floor(idx))] * (1.0 - frac) + table[int(math.ceil(idx))] * frac


def mag_lon(et):
    """transform.py:497-508"""
    return math.atan2(_igrf(IGRF_H11, et), _igrf(IGRF_G11, et)) + math.pi


def mag_lat(et):
    """transform.py:510-523"""
    g01, g11, h11 = _igrf(IGRF_G01, et), _igrf(IGRF_G11, et), _igrf(IGRF_H11, et)
    lam = mag_lon(et)
    return math.pi / 2 - math.atan((g11 * math.cos(lam) + h11 * math.sin(lam)) / g01)


def rotation_matrix3(angle, direction):
    """transformations.py:295-336 restricted to the 3x3 block and rotation about the origin."""
    sina, cosa = math.sin(angle), math.cos(angle)
    d = np.array(direction[:3], dtype=np.float64)
    d /= math.sqrt(np.dot(d, d))
    rot = np.diag([cosa, cosa, cosa])
    rot += np.outer(d, d) * (1.0 - cosa)
    d *= sina
    rot += np.array([[0.0, -d[2], d[1]],
                     [d[2], 0.0, -d[0]],
                     [-d[1], d[0], 0.0]])
    return rot


def euler_matrix_rzxz(ai, aj, ak):
    """transformations.py:1042-1102 for axes='rzxz' = (firstaxis 2, parity 0, repetition 1, frame 1)."""
    i, j, k = 2, 0, 1
    ai, ak = ak, ai                      # rotating frame
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs = ci * ck, ci * sk
    sc, ss = si * ck, si * sk
    m = np.identity(3)
    m[i, i] = cj
    m[i, j] = sj * si
    m[i, k] = sj * ci
    m[j, i] = sj * sk
    m[j, j] = -cj * ss + cc
    m[j, k] = -cj * cs - sc
    m[k, i] = -sj * ck
    m[k, j] = cj * sc + cs
    m[k, k] = cj * cc - ss
    return m


_X, _Y, _Z = [-1, 0, 0], [0, 1, 0], [0, 0, -1]      # transform.py:491-494


def _T0(et):
    return (et / 86400.0) / 36525.0               # transform.py:534-538


def _H(et):
    jd = (et / 86400.0) - 0.5                     # transform.py:540-551
    hh = (jd - int(jd)) * 24.0
    return hh + 24.0 if hh < 0.0 else hh


def mat_P(et):
    """transform.py:568-581 (J2000 -> GEI precession)"""
    t0 = _T0(et)
    m = rotation_matrix3(np.deg2rad(-1.0 * (0.64062 * t0 + 0.00030 * t0 * t0)), _Z)
    m = np.dot(m, rotation_matrix3(np.deg2rad(0.55675 * t0 - 0.00012 * t0 * t0), _Y))
    m = np.dot(m, rotation_matrix3(np.deg2rad(-1.0 * (0.64062 * t0 + 0.00008 * t0 * t0)), _Z))
    return m


def mat_T1(et):
    """transform.py:583-590 (GEI -> GEO)"""
    theta = 100.461 + 36000.770 * _T0(et) + 360.0 * (_H(et) / 24.0)
    return rotation_matrix3(np.deg2rad(theta), _Z)


def mat_T2(et):
    """transform.py:553-566,592-599 (GEI -> GSE)"""
    t0 = _T0(et)
    M = 357.528 + 35999.050 * t0
    lambd = 280.460 + 36000.772 * t0
    lam0 = lambd + (1.915 - 0.0048 * t0) * math.sin(np.deg2rad(M)) + 0.020 * math.sin(np.deg2rad(2 * M))
    eps = 23.439 - 0.013 * t0
    return np.dot(rotation_matrix3(np.deg2rad(lam0), _Z), rotation_matrix3(np.deg2rad(eps), _X))


def vec_Qe(et):
    """transform.py:601-620"""
    lat, lon = mag_lat(et), mag_lon(et)
    qg = [math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat)]
    return np.dot(np.dot(mat_T2(et), mat_T1(et).T), qg)


def mat_T3(et):
    """transform.py:622-629 (GSE -> GSM)"""
    qe = vec_Qe(et)
    psi = math.atan2(np.deg2rad(qe[1]), np.deg2rad(qe[2]))
    return rotation_matrix3(-psi, _X)


def mat_T4(et):
    """transform.py:631-638 (GSM -> SM)"""
    qe = vec_Qe(et)
    mu = math.atan2(np.deg2rad(qe[0]), np.deg2rad(math.sqrt(qe[1] * qe[1] + qe[2] * qe[2])))
    return rotation_matrix3(-mu, _Y)


def mat_j2000_to_geo(et):
    return np.dot(mat_T1(et), mat_P(et))                                      # transform.py:683-686


def mat_j2000_to_sm(et):
    return mat_T4(et).dot(mat_T3(et)).dot(mat_T2(et)).dot(mat_P(et))         # transform.py:688-691


def mat_geo_to_sm(et):
    return mat_T4(et).dot(mat_T3(et)).dot(mat_T2(et)).dot(mat_T1(et).T)      # transform.py:693-696


def rotate_vectors(mat, vecs):
    """transform.py:728-738 (x_to_y): batched 3x3 @ (N,3,1)."""
    vecs = np.asarray(vecs)
    assert vecs.ndim == 2 and vecs.shape[1] == 3
    return np.matmul(mat, vecs[..., np.newaxis]).reshape(vecs.shape)


# ---------------------------------------------------------------------------
# camera model: TAN WCS -> unit direction per pixel corner / centre
# ---------------------------------------------------------------------------
def wcs_rotation(hdr):
    """wcs.py:135-139: native -> celestial rotation."""
    return euler_matrix_rzxz(np.deg2rad(hdr['CRVAL1'] + 90), np.deg2rad(90 - hdr['CRVAL2']),
                             np.deg2rad(-(hdr['LONPOLE'] - 90)))


def pixel_directions(hdr, corner=True):
    """wcs.py:18-64 + 66-144 (TAN only, ascartesian=True) and astrometry.py:245-269."""
    assert hdr['CTYPE1'] == 'RA---TAN' and hdr['CTYPE2'] == 'DEC--TAN' and hdr['LATPOLE'] == 0.0
    w, h = hdr['IMAGEW'], hdr['IMAGEH']
    start = -0.5 if corner else 0
    x, y = np.meshgrid(np.arange(start, start + w + corner), np.arange(start, start + h + corner))
    shape = x.shape
    pxy = np.empty((x.size, 2), float)
    pxy[:, 0] = x.ravel()
    pxy[:, 0] -= hdr['CRPIX1']
    pxy[:, 1] = y.ravel()
    pxy[:, 1] -= hdr['CRPIX2']
    pxy += 1
    cd = np.array([[hdr['CD1_1'], hdr['CD1_2']], [hdr['CD2_1'], hdr['CD2_2']]])
    xy = np.matmul(cd, pxy[..., np.newaxis]).reshape(pxy.shape)
    r = np.sqrt((xy * xy).sum(axis=1))
    lon = np.arctan2(xy[:, 0], -xy[:, 1])
    with np.errstate(divide='ignore'):
        np.reciprocal(r, r)
    np.multiply(180 / np.pi, r, r)
    np.arctan(r, r)
    lat = r
    lmn = np.empty((3,) + lat.shape)
    np.cos(lat, lmn[0])
    lmn[1][:] = lmn[0]
    lmn[0] *= np.cos(lon)
    lmn[1] *= np.sin(lon)
    np.sin(lat, lmn[2])
    lmn = np.rollaxis(lmn, 0, lmn.ndim)
    rot = np.matmul(wcs_rotation(hdr), lmn[..., np.newaxis]).reshape(lmn.shape)
    return rot.reshape(shape + (3,))


# ---------------------------------------------------------------------------
# ray / ellipsoid of revolution
# ---------------------------------------------------------------------------
def _scaled_dots(a, b, origin, dirs):
    origin = np.require(origin, dtype=np.float64)
    dirs = np.require(dirs, dtype=np.float64)
    direction = dirs.T
    org = -origin[:, None]
    radius = np.array([[1 / a], [1 / a], [1 / b]])
    ds = direction * radius
    os_ = org * radius
    d_o = np.einsum('ij,ij->j', ds, os_)
    d_d = np.einsum('ij,ij->j', ds, ds)
    o_o = np.einsum('ij,ij->j', os_, os_)
    return direction, org, ds, d_o, d_d, o_o


def _inside(origin, a, b):
    x, y, z = origin
    return (x / a) ** 2 + (y / a) ** 2 + (z / b) ** 2 < 1          # intersection.py:239-241


def ellipsoid_line_intersection(a, b, origin, dirs, directed=True):
    """intersection.py:58-104 (`_np` twin), incl. NaN for misses / points behind a directed ray."""
    direction, org, ds, d_o, d_d, o_o = _scaled_dots(a, b, origin, dirs)
    root = np.square(d_o)
    root -= o_o * d_d
    root += d_d
    with np.errstate(invalid='ignore'):
        np.sqrt(root, root)
        if directed:
            if _inside(np.asarray(origin, dtype=np.float64), a, b):
                t = d_o
                t += root
            else:
                t = d_o
                t -= root
            t[t < 0] = np.nan                                       # intersection.py:50-56
        else:
            t1 = d_o - root
            t2 = d_o
            t2 += root
            t = np.where(np.abs(t1) < np.abs(t2), t1, t2)           # intersection.py:243-250
    t /= d_d
    res = ds
    np.multiply(direction, t, res)
    res -= org
    return res.T


def ellipsoid_line_intersects(a, b, origin, dirs, directed=True):
    """intersection.py:165-201"""
    _, _, _, d_o, d_d, o_o = _scaled_dots(a, b, origin, dirs)
    root = np.square(d_o)
    root -= o_o * d_d
    root += d_d
    with np.errstate(invalid='ignore'):
        if not directed:
            return root >= 0
        np.sqrt(root, root)
        if _inside(np.asarray(origin, dtype=np.float64), a, b):
            d_o += root
        else:
            d_o -= root
        return d_o >= 0


def sphere_line_intersection(radius, origin, dirs, directed=True):
    """intersection.py:12-48"""
    origin = np.asarray(origin, dtype=np.float64)
    dirs = np.asarray(dirs, dtype=np.float64)
    dp = np.dot(dirs, origin)
    root = np.square(dp)
    root -= np.dot(origin, origin)
    root += np.square(radius)
    with np.errstate(invalid='ignore'):
        root = np.sqrt(root)
        if directed:
            inside = math.sqrt(float(np.dot(origin, origin))) < radius
            t = np.atleast_1d(-dp + root if inside else -dp - root).astype(np.float64)
            t[t < 0] = np.nan
            if np.ndim(dp) == 0:
                t = t[0]
        else:
            t1, t2 = -dp - root, -dp + root
            t = np.where(np.abs(t1) < np.abs(t2), t1, t2)
    return origin + np.asarray(t)[..., None] * dirs


def inflated_earth_intersection(dirs, cam, inflation=110):
    """mapping.py:1474-1510 (earthModel='wgs84')"""
    return ellipsoid_line_intersection(WGS84_A + inflation, WGS84_B + inflation, cam, dirs)


# ---------------------------------------------------------------------------
# ECEF <-> geodetic, spherical helpers
# ---------------------------------------------------------------------------
def ecef_to_geodetic(x, y, z, a=WGS84_A, b=WGS84_B):
    """transform.py:252-297 (Bowring 1985, in-place array form). Returns radians."""
    x, y, z = (np.array(v, dtype=np.float64, ndmin=1) for v in (x, y, z))
    e2 = (a * a - b * b) / (a * a)
    d = (a * a - b * b) / b
    p2 = np.square(x)
    p = np.square(y)
    p2 += p
    np.sqrt(p2, p)
    r = p2
    tu = z * z
    r += tu
    np.sqrt(r, r)
    np.divide(d, r, tu)
    tu += 1
    tu *= b
    tu *= z
    ap = a * p
    tu /= ap
    tu2 = np.square(tu, r)
    cu3 = ap
    np.add(1, tu2, cu3)
    np.sqrt(cu3, cu3)
    np.reciprocal(cu3, cu3)
    cu3 **= 3                    # transform.py:279: must stay a power, not c*c*c
    su3 = tu
    su3 *= cu3
    su3 *= tu2
    tp = tu2
    np.multiply(d, su3, tp)
    tp += z
    cu3 *= e2 * a
    p -= cu3
    tp /= p
    np.arctan(tp, tp)
    lon = cu3
    np.arctan2(y, x, lon)
    return tp, lon


def geodetic_to_ecef(lat, lon, h, a=WGS84_A, b=WGS84_B):
    """transform.py:156-178 (radians in)"""
    lat, lon, h = np.asarray(lat), np.asarray(lon), np.asarray(h)
    e2 = (a * a - b * b) / (a * a)
    n = a / np.sqrt(1 - e2 * np.sin(lat) ** 2)
    lat_cos = np.cos(lat)
    nh = n + h
    return nh * lat_cos * np.cos(lon), nh * lat_cos * np.sin(lon), (n * (1 - e2) + h) * np.sin(lat)


def geodetic_to_ecef_zero(lat, lon, a=WGS84_A, b=WGS84_B):
    """transform.py:180-197"""
    lat, lon = np.asarray(lat), np.asarray(lon)
    e2 = (a * a - b * b) / (a * a)
    n = a / np.sqrt(1 - e2 * np.sin(lat) ** 2)
    latn = n * np.cos(lat)
    return latn * np.cos(lon), latn * np.sin(lon), n * (1 - e2) * np.sin(lat)


def cartesian_to_spherical(x, y, z):
    """transform.py:104-127 -> (r, lat, lon) radians"""
    xsq, ysq, zsq = x * x, y * y, z * z
    r = xsq + ysq
    r += zsq
    np.sqrt(r, r)
    s = xsq
    s += ysq
    np.sqrt(s, s)
    lon = np.arctan2(y, x)
    lat = np.arctan2(z, s)
    return r, lat, lon


def spherical_to_cartesian(r, lat, lon):
    """transform.py:38-63 -> (x, y, z)"""
    x = np.cos(lat)
    if r is not None:
        x *= r
    y = x.copy()
    x *= np.cos(lon)
    y *= np.sin(lon)
    z = np.sin(lat)
    if r is not None:
        z *= r
    return x, y, z


def j2000_to_latlon(vecs, m_geo):
    """transform.py:324-343 with the matrix passed in. Degrees out."""
    gx, gy, gz = rotate_vectors(m_geo, vecs).T
    lat, lon = ecef_to_geodetic(gx, gy, gz)
    np.rad2deg(lat, lat)
    np.rad2deg(lon, lon)
    return lat, lon


def sm_lon_to_mlt(smlon):
    mlt = smlon * (24 / 360)                    # transform.py:373-386
    mlt += 12
    return mlt


def mlt_to_sm_lon(mlt):
    smlon = mlt - 12                            # transform.py:388-401
    smlon /= (24 / 360)
    return smlon


def _to_mlat_mlt(mat, vecs):
    sx, sy, sz = rotate_vectors(mat, vecs).T
    _, smlat, smlon = cartesian_to_spherical(sx, sy, sz)
    np.rad2deg(smlat, smlat)
    np.rad2deg(smlon, smlon)
    return smlat, sm_lon_to_mlt(smlon)


def j2000_to_mlat_mlt(vecs, m_sm):
    """transform.py:403-430 -> (mlat deg, mlt h)"""
    return _to_mlat_mlt(m_sm, vecs)


def geo_to_mlat_mlt(vecs, m_geo_sm):
    """transform.py:432-459"""
    return _to_mlat_mlt(m_geo_sm, vecs)


def sm_to_latlon(smlats, smlons, m_geo_sm):
    """transform.py:461-485 (degrees in/out); m_geo_sm is GEO->SM, applied transposed."""
    smlats, smlons = np.deg2rad(smlats), np.deg2rad(smlons)
    sx, sy, sz = spherical_to_cartesian(1, smlats.ravel(), smlons.ravel())
    gx, gy, gz = rotate_vectors(m_geo_sm.T, np.array([sx, sy, sz]).T).T
    lats, lons = ecef_to_geodetic(gx, gy, gz)
    np.rad2deg(lats, lats)
    np.rad2deg(lons, lons)
    return lats.reshape(smlats.shape), lons.reshape(smlons.shape)


def rotate_pole(lats, lons, altitude, angle=90, axis=(1, 0, 0)):
    """transform.py:301-322 (radians in/out)"""
    assert lats.ndim == 1 and lons.ndim == 1
    x, y, z = geodetic_to_ecef(lats, lons, altitude)
    xyz = np.asarray([x, y, z]).T
    rot = rotation_matrix3(np.deg2rad(angle), list(axis))
    r = rotate_vectors(rot, xyz)
    return ecef_to_geodetic(r[:, 0], r[:, 1], r[:, 2])


def wrap_at(deg, wrap):
    """astropy Angle.wrap_at (astropy >= 1.0 `_wrap_at`), as used at resample.py:213,218,276-277."""
    a = np.array(deg, dtype=np.float64, copy=True)
    wraps = (a - (wrap - 360.0)) // 360.0
    a = a - wraps * 360.0
    a = np.where(a >= wrap, a - 360.0, a)
    a = np.where(a < wrap - 360.0, a + 360.0, a)
    return a


# ---------------------------------------------------------------------------
# frame-level georeferencing (astrometry.py:49-212)
# ---------------------------------------------------------------------------
def calc_centers(corners):
    c = corners[:-1, :-1] + corners[:-1, 1:]       # astrometry.py:154-160
    c += corners[1:, 1:]
    c += corners[1:, :-1]
    c /= 4
    return c


def elevation_deg(dir_center, p_center):
    """astrometry.py:200-212 + utils.py:28-46"""
    to_cam = -dir_center.reshape(-1, 3)
    p = p_center.reshape(-1, 3)
    with np.errstate(invalid='ignore'):
        unit = p / np.sqrt((p * p).sum(axis=1))[..., None]
        dot = np.clip(np.einsum('...i,...i->...', to_cam, unit), -1, 1)
        alpha = np.arccos(dot)
    alpha = alpha.reshape(p_center.shape[:2])
    np.rad2deg(alpha, alpha)
    np.subtract(90, alpha, alpha)
    return alpha


def georef_frame(hdr, altitude, cam, m_geo, m_sm=None, fast=True):
    """
    All raw (NaN-for-missing) arrays a BaseAstrometryMapping derives lazily
    (astrometry.py:49-212): corner/centre directions, intersections, lat/lon,
    elevation and, when `m_sm` is given, MLat/MLT.
    """
    cam = np.asarray(cam, dtype=np.float64)
    out = {}
    dir_c = pixel_directions(hdr, corner=True)
    p_c = inflated_earth_intersection(dir_c.reshape(-1, 3), cam, altitude).reshape(dir_c.shape)
    if fast:
        dir_m = calc_centers(dir_c)
        with np.errstate(invalid='ignore'):
            p_m = calc_centers(p_c)
    else:
        dir_m = pixel_directions(hdr, corner=False)
        p_m = inflated_earth_intersection(dir_m.reshape(-1, 3), cam, altitude).reshape(dir_m.shape)
    lat, lon = j2000_to_latlon(p_c.reshape(-1, 3), m_geo)
    latc, lonc = j2000_to_latlon(p_m.reshape(-1, 3), m_geo)
    out.update(dir_corner=dir_c, p_corner=p_c, dir_center=dir_m, p_center=p_m,
               lat=lat.reshape(p_c.shape[:2]), lon=lon.reshape(p_c.shape[:2]),
               lat_c=latc.reshape(p_m.shape[:2]), lon_c=lonc.reshape(p_m.shape[:2]),
               elev=elevation_deg(dir_m, p_m))
    if m_sm is not None:
        mlat, mlt = j2000_to_mlat_mlt(p_c.reshape(-1, 3), m_sm)
        mlatc, mltc = j2000_to_mlat_mlt(p_m.reshape(-1, 3), m_sm)
        out.update(mlat=mlat.reshape(p_c.shape[:2]), mlt=mlt.reshape(p_c.shape[:2]),
                   mlat_c=mlatc.reshape(p_m.shape[:2]), mlt_c=mltc.reshape(p_m.shape[:2]))
    return out


# ---------------------------------------------------------------------------
# other camera models on the same intersection + geodetic steps: all-sky fisheye (mapping/miracle.py),
# THEMIS altitude reprojection (mapping/themis.py)
# ---------------------------------------------------------------------------
def allsky_az_el(size, xc, yc, k, rotation, center=True, center_offset=0.5):
    """
    mapping/miracle.py:314-347 calculateAzEl; xc, yc, k refer to a 512 px image (:320-326).
    `center_offset`: the reference writes ``ind += 0.5`` on an integer index array (:333-334), which NumPy 1.6
    (requirements.txt) silently truncates back to +0 and NumPy >= 1.10 rejects; 0.5 is the documented intent.
    """
    w = size
    scale = w / 512
    if w != 512:
        xc, yc, k = xc * scale, yc * scale, k * scale
    w_ = w if center else w + 1
    ind = np.indices((w_, w_)).astype(np.float64)
    if center:
        ind += center_offset
    ind = np.dstack((ind[0], ind[1])).reshape(w_ * w_, 2)
    vecs = ind - np.array([xc, yc])
    north = np.repeat([[-1, 0]], len(vecs), axis=0)
    # utils.py:48-56 signedAngleBetween
    az = np.arctan2(vecs[:, 0] * north[:, 1] - vecs[:, 1] * north[:, 0],
                    vecs[:, 0] * north[:, 0] + vecs[:, 1] * north[:, 1]).reshape(w_, w_)
    az -= rotation
    az = wrap_at(az * (180.0 / np.pi), 360.0)
    z = np.sqrt((vecs * vecs).sum(axis=1)).reshape(w_, w_) / k
    np.rad2deg(z, z)
    return az, 90 - z


def allsky_directions(el, az, station_lat, station_lon):
    """mapping/miracle.py:239-258 _calculateCameraToPixelDirection (degrees in, GEO unit vectors out)"""
    el = np.deg2rad(el)
    az = np.deg2rad(-(az - 180))
    x, y, z = spherical_to_cartesian(1, el, az)
    vecs = np.dstack((x, y, z))
    mat_lat = rotation_matrix3(np.deg2rad(90 - station_lat), [0, 1, 0])       # Y, transform.py:493
    mat_lon = rotation_matrix3(np.deg2rad(-station_lon), [0, 0, -1])          # Z, transform.py:494
    mat = np.dot(mat_lon, mat_lat)
    return np.matmul(mat, vecs.reshape(-1, 3)[..., None]).reshape(el.shape[0], el.shape[1], 3)


def allsky_georef(size, cal, altitude=110, center_offset=0.5):
    """
    mapping/miracle.py:139-140,196-237,260-272: dict(lat, lon, lat_c, lon_c [deg], elev, az, az_c, el_corner) of an
    all-sky frame; `cal` = dict(lat, lon, xc, yc, k, rotation).
    """
    cam = np.array(geodetic_to_ecef_zero(np.deg2rad(cal['lat']), np.deg2rad(cal['lon'])))
    out = {}
    for center in (False, True):
        az, el = allsky_az_el(size, cal['xc'], cal['yc'], cal['k'], cal['rotation'], center, center_offset)
        dirs = allsky_directions(el, az, cal['lat'], cal['lon'])
        hit = inflated_earth_intersection(dirs.reshape(-1, 3), cam, altitude)
        lat, lon = ecef_to_geodetic(hit[:, 0], hit[:, 1], hit[:, 2])
        lat, lon = np.rad2deg(lat).reshape(el.shape), np.rad2deg(lon).reshape(el.shape)
        if center:
            out.update(lat_c=lat, lon_c=lon, elev=el, az_c=az, dirs_c=dirs)
        else:
            out.update(lat=lat, lon=lon, el_corner=el, az=az, dirs=dirs)
    return out


def themis_reproject(lat_lon_asi, lats_ref, lons_ref, height_ref, height_new):
    """mapping/themis.py:224-253 reproject (degrees in and out)"""
    lat_asi, lon_asi = lat_lon_asi
    cam = np.array(geodetic_to_ecef_zero(np.deg2rad(lat_asi), np.deg2rad(lon_asi)))
    x, y, z = geodetic_to_ecef(np.deg2rad(lats_ref), np.deg2rad(lons_ref), height_ref)
    direction = np.transpose([x - cam[0], y - cam[1], z - cam[2]])
    hit = ellipsoid_line_intersection(WGS84_A + height_new, WGS84_B + height_new, cam, direction.reshape(-1, 3))
    hit = hit.reshape(direction.shape)
    x, y, z = hit.transpose()
    lat, lon = ecef_to_geodetic(x, y, z)
    return np.rad2deg(lat), np.rad2deg(lon)


# ---------------------------------------------------------------------------
# outline of a validity mask, polygon area / centroid (utils.py:97-225, mapping.py:655-691,758-784)
# ---------------------------------------------------------------------------
def find_contours_binary(image, level=0.99):
    """
    Third-party step of utils.py:97-103: ``skimage.measure.find_contours(image, level)`` of scikit-image 0.10.1
    (requirements.txt; absent from this image).  Restated from its published algorithm (marching squares with linear
    interpolation, `fully_connected='low'`, `positive_orientation='low'`; Lorensen & Cline 1987 as implemented in
    skimage/measure/_find_contours*.py): every 2x2 square contributes 0, 1 or 2 directed segments between points
    interpolated on its edges, high values on the right-hand side of the direction of travel (image coordinates);
    segments are joined into contours, closed ones repeat their first point at the end.  Returns a list of (n,2)
    arrays in (row, col) order.  Parity of this restatement is anchored on the reference's own vectors: the literal
    polygon of outline_test.py:110-131 (the outline of its `_testIm(10)`), polygon area / centroid known answers and
    the mapping centroid of outline_test.py:151-158.
    """
    a = np.asarray(image, dtype=np.float64)

    def frac(v_from, v_to):
        return (level - v_from) / (v_to - v_from)
    segments = {}
    hi = a > level
    cases = hi[:-1, :-1] * 1 + hi[:-1, 1:] * 2 + hi[1:, :-1] * 4 + hi[1:, 1:] * 8
    for r0, c0 in zip(*np.nonzero((cases != 0) & (cases != 15))):       # squares in raster order, as the scan visits them
        r0, c0 = int(r0), int(c0)
        if True:
            ul, ur, ll, lr = a[r0, c0], a[r0, c0 + 1], a[r0 + 1, c0], a[r0 + 1, c0 + 1]
            case = int(cases[r0, c0])
            top = (r0, c0 + frac(ul, ur)) if (ul > level) != (ur > level) else None
            bottom = (r0 + 1, c0 + frac(ll, lr)) if (ll > level) != (lr > level) else None
            left = (r0 + frac(ul, ll), c0) if (ul > level) != (ll > level) else None
            right = (r0 + frac(ur, lr), c0 + 1) if (ur > level) != (lr > level) else None
            table = {1: [(top, left)], 2: [(right, top)], 3: [(right, left)], 4: [(left, bottom)],
                     5: [(top, bottom)], 6: [(right, top), (left, bottom)], 7: [(right, bottom)],
                     8: [(bottom, right)], 9: [(top, left), (bottom, right)], 10: [(bottom, top)],
                     11: [(bottom, left)], 12: [(left, right)], 13: [(top, right)], 14: [(left, top)]}
            for p, q in table[case]:
                segments[p] = q
    contours = []
    while segments:
        start, nxt = segments.popitem()
        pts = [start, nxt]
        while nxt in segments:
            nxt = segments.pop(nxt)
            pts.append(nxt)
        # an open contour (touching the image border) may continue backwards; masks padded with False never do
        contours.append(np.array(pts, dtype=np.float64))
    return contours


def polygon_area(poly, signed=False):
    """utils.py:153-171"""
    poly = [list(p) for p in np.asarray(poly).tolist()]
    segments = zip(poly, poly[1:] + [poly[0]])
    area = 0.5 * sum(x0 * y1 - x1 * y0 for ((x0, y0), (x1, y1)) in segments)
    return area if signed else abs(area)


def polygon_centroid(poly):
    """utils.py:173-225 (moments over the unsigned area)"""
    poly = np.asarray(poly).tolist()
    area = polygon_area(poly)
    rx = ry = 0
    n = len(poly)
    for i in range(n):
        x0, y0 = poly[i]
        x1, y1 = poly[(i + 1) % n]
        cross = (x0 * y1) - (x1 * y0)
        rx += (x0 + x1) * cross
        ry += (y0 + y1) * cross
    return rx / (area * 6.0), ry / (area * 6.0)


def outline(im):
    """utils.py:97-151 _outline_skimage: (n,2) int array in x,y order, clockwise in image coordinates"""
    im = np.asarray(im, dtype=bool)
    border = np.zeros((im.shape[0] + 2, im.shape[1] + 2), dtype=bool)
    border[1:-1, 1:-1] = im
    contours = find_contours_binary(border, 0.99)

    def fix(contour):
        contour = np.fliplr(contour)                        # (row, col) -> (x, y)
        contour = np.round(contour - 1).astype(int)         # padding off, nearest pixel
        keep = np.ones(len(contour), bool)
        keep[1:] = np.any(contour[1:] != contour[:-1], axis=1)
        return contour[keep][:-1]                           # consecutive duplicates off, un-close
    if len(contours) > 1:
        contours = [c for c in map(fix, contours) if len(c) > 2]
        return contours[int(np.argmax([polygon_area(c) for c in contours]))]
    return fix(contours[0])


# ---------------------------------------------------------------------------
# mask rules (mapping.py:299-316, 845-864, 1063-1125)
# ---------------------------------------------------------------------------
def _all_neighbours_missing(center_mask):
    pad = np.ones((center_mask.shape[0] + 2, center_mask.shape[1] + 2), bool)
    pad[1:-1, 1:-1] = center_mask
    return np.logical_and.reduce((pad[1:, 1:], pad[1:, :-1], pad[:-1, :-1], pad[:-1, 1:]))


def sanitize_masks(corner_mask, center_mask, img_mask=None, after_masking=False):
    """
    mapping.py:1063-1125 on plain boolean masks (True = masked).  Returns the
    new (corner_mask, center_mask); image and elevation take the centre mask.
    """
    corner_mask = corner_mask.copy()
    center_mask = center_mask.copy()
    if img_mask is not None:
        center_mask |= img_mask
    corner_mask |= _all_neighbours_missing(center_mask)
    if not after_masking:
        any_missing = np.logical_or.reduce((corner_mask[:-1, :-1], corner_mask[1:, :-1],
                                            corner_mask[1:, 1:], corner_mask[:-1, 1:]))
        center_mask |= any_missing
        corner_mask |= _all_neighbours_missing(center_mask)
    return corner_mask, center_mask


def mask_by_elevation(elev, corner_nan_mask, min_elevation=10):
    """mapping.py:845-864 followed by the lazy `_doSanitize(afterMasking=True)` of mapping.py:1161-1213."""
    with np.errstate(invalid='ignore'):
        center_mask = ~(elev >= min_elevation)          # (elev < min).filled(True) with NaN == masked
    if np.all(center_mask):
        raise ValueError('minElevation=' + str(min_elevation) + ' would mask all pixels!')
    return sanitize_masks(corner_nan_mask, center_mask, after_masking=True)


# ---------------------------------------------------------------------------
# histogram binning (util/histogram.py:57-282) and plate-carree resampling (resample.py:159-368)
# ---------------------------------------------------------------------------
def histogram2d(x, y, bins, range=None, weights=None):
    """
    util/histogram.py:284-417 -> histogramdd :57-282, D = 2, normed=False.
    `weights` is None, an array, or a list of (array | None); returns
    (H or [H...], xedges, yedges) with H of shape (nx, ny).
    """
    sample = np.atleast_2d([x, y]).T
    n, d = sample.shape
    try:
        if len(bins) != 2:
            bins = [np.asarray(bins, float)] * 2
    except TypeError:
        bins = [bins, bins]
    as_list = isinstance(weights, (list, tuple))
    wlist = list(weights) if as_list else [weights]
    wlist = [None if w is None else np.asarray(w) for w in wlist]

    edges, nbin = [], []
    for i in (0, 1):
        if np.isscalar(bins[i]):
            if bins[i] < 1:
                raise ValueError('Element at index %s in `bins` should be a positive integer.' % i)
            if range is None:
                smin, smax = ((0.0, 1.0) if n == 0 else
                              (float(sample[:, i].min()), float(sample[:, i].max())))
            else:
                smin, smax = (float(v) for v in range[i])
            if smin == smax:
                smin, smax = smin - .5, smax + .5
            e = np.linspace(smin, smax, bins[i] + 1)
        else:
            e = np.asarray(bins[i], float)
        if np.any(np.diff(e) <= 0):
            raise ValueError('Found bin edge of size <= 0. Did you specify `bins` with non-monotonic sequence?')
        edges.append(e)
        nbin.append(len(e) + 1)

    if n == 0:
        hs = [np.zeros((nbin[0] - 2, nbin[1] - 2)) for _ in wlist]
        return (hs if as_list else hs[0]), edges[0], edges[1]

    idx = []
    for i in (0, 1):
        c = np.searchsorted(edges[i], sample[:, i], 'right')
        mindiff = np.diff(edges[i]).min()
        if not np.isinf(mindiff):
            decimal = int(-np.log10(mindiff)) + 6
            on_edge = np.around(sample[:, i], decimal) == np.around(edges[i][-1], decimal)
            c[np.where(on_edge & (sample[:, i] >= edges[i][-1]))[0]] -= 1
        idx.append(c)
    flat = idx[0] * nbin[1] + idx[1]
    hs = []
    for w in wlist:
        full = np.zeros(nbin[0] * nbin[1], float)
        counts = np.bincount(flat, w)
        full[:len(counts)] = counts
        hs.append(full.reshape(nbin[0], nbin[1])[1:-1, 1:-1])
    return (hs if as_list else hs[0]), edges[0], edges[1]


def fixed_grid(px_per_deg, lat_min, lat_max, lon_min, lon_max):
    """resample.py:281-299"""
    lat_ppd, lon_ppd = px_per_deg
    lat_all = np.linspace(-90, 90, int(round(lat_ppd * 180 + 1)))
    lon_all = np.linspace(-180, 180, int(round(lon_ppd * 360 + 1)))
    lat_lo = lat_all[np.argmax(lat_all > lat_min) - 1]
    lat_hi = lat_all[np.argmax(lat_all >= lat_max)]
    lon_lo = lon_all[np.argmax(lon_all > lon_min) - 1]
    lon_hi = lon_all[np.argmax(lon_all >= lon_max)]
    n_lat = int(round(lat_ppd * (lat_hi - lat_lo) + 1))
    n_lon = int(round(lon_ppd * (lon_hi - lon_lo) + 1))
    return n_lat, n_lon, lat_lo, lat_hi, lon_lo, lon_hi


def bin_mean(lats_c, lons_c, data, lat_centers, lon_centers, lat_step, lon_step):
    """resample.py:301-368 for method='mean'. data: (h,w,n). Returns (mean (nlat,nlon,n), count (nlat,nlon))."""
    la, lo = np.ravel(lats_c), np.ravel(lons_c)
    keep = ~np.isnan(la)
    la, lo = la[keep], lo[keep]
    flat = data.reshape(-1, data.shape[2])[keep]
    bins = (len(lon_centers), len(lat_centers))
    rng = [[lon_centers[0] - lon_step / 2, lon_centers[-1] + lon_step / 2],
           [lat_centers[-1] + lat_step / 2, lat_centers[0] - lat_step / 2]]
    hs, _, _ = histogram2d(lo, la, bins=bins, range=rng,
                           weights=[None] + [flat[:, k] for k in np.arange(flat.shape[1])])
    count = hs[0].T
    planes = []
    for k in np.arange(flat.shape[1]):
        s = hs[k + 1].T
        s[count == 0.0] = np.nan
        with np.errstate(invalid='ignore'):
            s /= count
        planes.append(np.flipud(s))
    return np.dstack(planes), np.flipud(count)


def points_inside_polygon(points, polygon):
    """utils.py:58-74 pointsInsidePolygon: matplotlib's Path.contains_points, the library call the reference makes
    (matplotlib is part of this image); points, polygon: (n,2)"""
    import matplotlib.path
    return matplotlib.path.Path(polygon).contains_points(points)


def resample_nearest(lats_c, lons_c, altitude, data, outline_latlon, bbox, px_per_deg,
                     contains_discontinuity=False, contains_pole=False):
    """resample.py:159-279 for method='nearest': scipy.interpolate.griddata (the reference's own call, :323-327) and
    the outside-outline masking (:246-259).  Returns dict(lat, lon, lat_c, lon_c, data)."""
    return resample_mean(lats_c, lons_c, altitude, data, outline_latlon, bbox, px_per_deg, contains_discontinuity,
                         contains_pole, method='nearest')


def resample_mean(lats_c, lons_c, altitude, data, outline_latlon, bbox, px_per_deg,
                  contains_discontinuity=False, contains_pole=False, method='mean'):
    """
    resample.py:159-279 for method='mean' (and, through resample_nearest, 'nearest').
    bbox = (latSouth, lonWest, latNorth, lonEast); outline_latlon (n,2) is only
    used (min/max) in the pole / discontinuity branches of 'mean'.
    Returns dict(lat, lon, lat_c, lon_c, data, count).
    """
    lat_min, lon_min, lat_max, lon_max = bbox
    if outline_latlon is not None:
        outline_latlon = np.array(outline_latlon, dtype=np.float64, copy=True)
    if contains_pole:
        ol = outline_latlon
        ola, olo = rotate_pole(np.deg2rad(ol[:, 0]), np.deg2rad(ol[:, 1]), altitude, angle=90, axis=(1, 0, 0))
        ola, olo = np.rad2deg(ola), np.rad2deg(olo)
        outline_latlon[:, 0], outline_latlon[:, 1] = ola, olo        # in place, as resample.py:192-193
        lat_min, lat_max = np.min(ola), np.max(ola)
        lon_min, lon_max = np.min(olo), np.max(olo)
        la, lo = rotate_pole(np.deg2rad(np.ravel(lats_c)), np.deg2rad(np.ravel(lons_c)), altitude,
                             angle=90, axis=(1, 0, 0))
        lats_c = np.rad2deg(la.reshape(lats_c.shape))
        lons_c = np.rad2deg(lo.reshape(lons_c.shape))
    elif contains_discontinuity:
        olo = wrap_at(np.asarray(outline_latlon)[:, 1] + 180, 180)
        outline_latlon[:, 1] = olo                                   # in place, as resample.py:214
        lon_min, lon_max = np.min(olo), np.max(olo)
        lons_c = wrap_at(lons_c + 180, 180)

    lat_ppd, lon_ppd = px_per_deg
    assert lat_ppd > 0 and lon_ppd > 0
    n_lat, n_lon, lat_lo, lat_hi, lon_lo, lon_hi = fixed_grid(px_per_deg, lat_min, lat_max, lon_min, lon_max)
    assert n_lat > 1 and n_lon > 1
    lat_centers, lat_step = np.linspace(lat_hi, lat_lo, num=n_lat, retstep=True)
    lon_centers, lon_step = np.linspace(lon_lo, lon_hi, num=n_lon, retstep=True)
    lat_nodes = lat_centers[:-1] + lat_step / 2
    lon_nodes = lon_centers[:-1] + lon_step / 2
    lat_centers = lat_centers[1:-1]
    lon_centers = lon_centers[1:-1]
    lat_grid, lon_grid = np.dstack(np.meshgrid(lat_nodes, lon_nodes)).T
    lat_grid_c, lon_grid_c = np.dstack(np.meshgrid(lat_centers, lon_centers)).T

    if data.ndim == 2:
        data = data[..., None]
    if method == 'mean':
        mean, count = bin_mean(lats_c, lons_c, data, lat_centers, lon_centers, lat_step, lon_step)
    else:
        import scipy.interpolate
        ok = ~np.isnan(np.ravel(lats_c))                               # resample.py:315-321
        flat = data.reshape(-1, data.shape[2])[ok]
        mean = scipy.interpolate.griddata((np.ravel(lats_c)[ok], np.ravel(lons_c)[ok]), flat,
                                          (lat_centers[:, None], lon_centers[None, :]), method=method)
        count = None
        pts = np.asarray([np.ravel(lat_grid), np.ravel(lon_grid)]).T    # resample.py:253-259
        outside = ~points_inside_polygon(pts, outline_latlon).reshape(lat_grid.shape)
        mask = np.logical_or.reduce((outside[:-1, :-1], outside[1:, :-1], outside[:-1, 1:], outside[1:, 1:]))
        mean[mask] = np.nan

    if contains_pole:
        def back(la, lo):
            a, o = rotate_pole(np.deg2rad(la.ravel()), np.deg2rad(lo.ravel()), altitude, angle=-90, axis=(1, 0, 0))
            return np.rad2deg(a.reshape(la.shape)), np.rad2deg(o.reshape(la.shape))
        lat_grid, lon_grid = back(lat_grid, lon_grid)
        lat_grid_c, lon_grid_c = back(lat_grid_c, lon_grid_c)
    elif contains_discontinuity:
        lon_grid = wrap_at(lon_grid + 180, 180)
        lon_grid_c = wrap_at(lon_grid_c + 180, 180)
    return dict(lat=lat_grid, lon=lon_grid, lat_c=lat_grid_c, lon_c=lon_grid_c, data=mean, count=count)


def finalize_image(mean_img, dtype):
    """resample.py:128-136: integer images are rounded half-to-even and cast; NaN cells become masked."""
    with np.errstate(invalid='ignore'):
        r = np.round(mean_img)
    mask = np.isnan(r)
    return np.where(mask, 0, r).astype(dtype), mask


def bbox_of_corners(lat, lon, corner_mask):
    """
    Stand-in for mapping.py:693-743 on mappings that contain neither a pole nor
    (optionally) the discontinuity: min/max over unmasked corners (the outline's
    extremes equal the region's extremes in that case).  Returns
    ((latSouth, lonWest, latNorth, lonEast), contains_discontinuity).
    """
    la = lat[~corner_mask]
    lo = lon[~corner_mask]
    lon_min, lon_max = lo.min(), lo.max()
    if lon_max - lon_min > 180:                              # mapping.py:726-734
        return (la.min(), lo[lo > 0].min(), la.max(), lo[~(lo > 0)].max()), True
    return (la.min(), lon_min, la.max(), lon_max), False


# ---- scipy.interpolate.griddata(method='cubic') restated (reference resample.py:323-326 calls it) ---------------------------
# scipy is a third-party dependency of the reference (requirements.txt pins no version; 1.15.3 in this image).  Its 2-D cubic
# is CloughTocher2DInterpolator (interpnd): vertex gradients from a global curvature-minimising estimator, then the
# 12-parameter Clough-Tocher element made affine invariant.  The two functions below restate both on a GIVEN triangulation
# (scipy.spatial.Delaunay's arrays) and are checked against scipy itself to rounding (tests/test_oracle_golden.py); the
# device kernels (k_cubic_sweep, clough_tocher in csrc/amt_nearest.hip) follow them on the pixel-grid triangulation.

def clough_tocher_gradients(points, indptr, indices, values, maxiter=400, tol=1e-6):
    """interpnd._estimate_gradients_2d_global: point after point, the 2x2 system of the Hermite-curve energy along the
    triangulation edges, neighbours' gradients as they are at that moment; stops when the largest relative change < tol.
    points (n,2); (indptr, indices) = Delaunay.vertex_neighbor_vertices; values (n,).  Returns ((n,2) gradients, sweeps)."""
    n = len(points)
    y = np.zeros((n, 2))
    for it in range(maxiter):
        err = 0.0
        for i in range(n):
            q0 = q1 = q3 = s0 = s1 = 0.0
            for j in indices[indptr[i]:indptr[i + 1]]:
                ex, ey = points[j, 0] - points[i, 0], points[j, 1] - points[i, 1]
                l3 = np.hypot(ex, ey) ** 3
                df2 = -ex * y[j, 0] - ey * y[j, 1]
                q0 += 4 * ex * ex / l3
                q1 += 4 * ex * ey / l3
                q3 += 4 * ey * ey / l3
                t = (6 * (values[i] - values[j]) - 2 * df2) / l3
                s0 += t * ex
                s1 += t * ey
            det = q0 * q3 - q1 * q1
            r0, r1 = (q3 * s0 - q1 * s1) / det, (-q1 * s0 + q0 * s1) / det
            change = max(abs(y[i, 0] + r0), abs(y[i, 1] + r1)) / max(1.0, abs(r0), abs(r1))
            y[i] = -r0, -r1
            err = max(err, change)
        if err < tol:
            return y, it + 1
    return y, maxiter


def clough_tocher_value(tri_xy, b, f, grad, neighbour_centroids):
    """interpnd._clough_tocher_2d_single: the element of triangle tri_xy (3,2) at barycentric coordinates b (3,), vertex values
    f (3,), vertex gradients grad (3,2); neighbour_centroids[k] = centroid of the triangle across the edge opposite vertex k,
    or None on the hull."""
    P = np.asarray(tri_xy, dtype=np.float64)
    e12, e23, e31 = P[1] - P[0], P[2] - P[1], P[0] - P[2]
    f1, f2, f3 = f
    d1, d2, d3 = grad
    df12, df21 = d1 @ e12, -(d2 @ e12)
    df23, df32 = d2 @ e23, -(d3 @ e23)
    df31, df13 = d3 @ e31, -(d1 @ e31)
    c3000, c2100, c2010 = f1, (df12 + 3 * f1) / 3, (df13 + 3 * f1) / 3
    c0300, c1200, c0210 = f2, (df21 + 3 * f2) / 3, (df23 + 3 * f2) / 3
    c0030, c1020, c0120 = f3, (df31 + 3 * f3) / 3, (df32 + 3 * f3) / 3
    c2001 = (c2100 + c2010 + c3000) / 3
    c0201 = (c1200 + c0300 + c0210) / 3
    c0021 = (c1020 + c0120 + c0030) / 3
    T = np.linalg.inv(np.array([[P[0, 0] - P[2, 0], P[1, 0] - P[2, 0]], [P[0, 1] - P[2, 1], P[1, 1] - P[2, 1]]]))
    g = [-0.5, -0.5, -0.5]
    for k in range(3):
        if neighbour_centroids[k] is None:
            continue
        c01 = T @ (np.asarray(neighbour_centroids[k]) - P[2])
        c = (c01[0], c01[1], 1 - c01[0] - c01[1])
        if k == 0:
            g[k] = (2 * c[2] + c[1] - 1) / (2 - 3 * c[2] - 3 * c[1])
        elif k == 1:
            g[k] = (2 * c[0] + c[2] - 1) / (2 - 3 * c[0] - 3 * c[2])
        else:
            g[k] = (2 * c[1] + c[0] - 1) / (2 - 3 * c[1] - 3 * c[0])
    c0111 = (g[0] * (-c0300 + 3 * c0210 - 3 * c0120 + c0030) + (-c0300 + 2 * c0210 - c0120 + c0021 + c0201)) / 2
    c1011 = (g[1] * (-c0030 + 3 * c1020 - 3 * c2010 + c3000) + (-c0030 + 2 * c1020 - c2010 + c2001 + c0021)) / 2
    c1101 = (g[2] * (-c3000 + 3 * c2100 - 3 * c1200 + c0300) + (-c3000 + 2 * c2100 - c1200 + c2001 + c0201)) / 2
    c1002 = (c1101 + c1011 + c2001) / 3
    c0102 = (c1101 + c0111 + c0201) / 3
    c0012 = (c1011 + c0111 + c0021) / 3
    c0003 = (c1002 + c0102 + c0012) / 3
    m = min(b)
    b1, b2, b3, b4 = b[0] - m, b[1] - m, b[2] - m, 3 * m
    return (b1 ** 3 * c3000 + 3 * b1 ** 2 * b2 * c2100 + 3 * b1 ** 2 * b3 * c2010 + 3 * b1 ** 2 * b4 * c2001
            + 3 * b1 * b2 ** 2 * c1200 + 6 * b1 * b2 * b4 * c1101 + 3 * b1 * b3 ** 2 * c1020 + 6 * b1 * b3 * b4 * c1011
            + 3 * b1 * b4 ** 2 * c1002 + b2 ** 3 * c0300 + 3 * b2 ** 2 * b3 * c0210 + 3 * b2 ** 2 * b4 * c0201
            + 3 * b2 * b3 ** 2 * c0120 + 6 * b2 * b3 * b4 * c0111 + 3 * b2 * b4 ** 2 * c0102 + b3 ** 3 * c0030
            + 3 * b3 ** 2 * b4 * c0021 + 3 * b3 * b4 ** 2 * c0012 + b4 ** 3 * c0003)
