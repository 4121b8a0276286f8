"""
Coordinate conversions between reference frames and representations.

Mirror of the reference's auromat/coordinates/transform.py.  Everything that is a whole-array
NumPy / numexpr pass there (Bowring ECEF->geodetic, the batched 3x3 rotations, cartesian <->
spherical, MLat/MLT) runs as a HIP kernel here; what the reference computes once per frame on
scalars (ephemeris time, IGRF dipole pole, the cxform-derived rotation matrices,
transform.py:497-696) stays on the host, in the reference's operation order so that the matrices
handed to the kernels are the same doubles.

Note that some functions depend on the IGRF model whose parameters are defined in the
:mod:`igrf` module (until the year {}).
"""
from __future__ import division

import functools
import math
from datetime import datetime

import numpy as np

from .geodesic import wgs84A, wgs84B, Location
from .igrf import IGRF_DEFINED_UNTIL_YEAR, calcG01, calcG11, calcH11
from .._native import host9
from .._ops import Staged, ptr

__doc__ = __doc__.format(IGRF_DEFINED_UNTIL_YEAR)


# ---------------------------------------------------------------------------------------------
# device-side array operations
# ---------------------------------------------------------------------------------------------
def _flat(st, *arrays):
    """Move equally shaped inputs to the device as flat float64 tensors; returns (tensors, shape)."""
    shape = None
    outs = []
    for a in arrays:
        t = st.inp(a if st.on_device or hasattr(a, 'shape') else np.asarray(a, dtype=np.float64))
        if shape is None:
            shape = tuple(t.shape)
        assert tuple(t.shape) == shape, 'inputs must have the same shape'
        outs.append(t.reshape(-1))
    return outs, shape


def spherical_to_cartesian(r, lat, lon, astuple=True):
    """
    Convert spherical to cartesian coordinates (reference transform.py:89-102). Inputs must be arrays.

    :type r: ndarray or None (=1)
    :rtype: tuple (x,y,z) of arrays with shape as input, or one (...,3) array if astuple=False
    """
    st = Staged(lat, lon, r)
    (la, lo), shape = _flat(st, lat, lon)
    rr = None
    if r is not None:
        rr = st.inp(np.broadcast_to(np.asarray(r, dtype=np.float64), shape) if not st.on_device else r).reshape(-1)
    n = la.numel()
    x, y, z = st.out((n,)), st.out((n,)), st.out((n,))
    st.ctx.call('amt_spherical_to_cartesian', ptr(rr), ptr(la), ptr(lo), n, ptr(x), ptr(y), ptr(z))
    if astuple:
        return st.result(x, shape), st.result(y, shape), st.result(z, shape)
    import torch
    return st.result(torch.stack((x, y, z), dim=-1), shape + (3,))


def cartesian_to_spherical(x, y, z, with_radius=True):
    """
    Convert cartesian to spherical coordinates (reference transform.py:142-154). Inputs must be arrays.

    :rtype: tuple (r,lat,lon) or (lat,lon) of arrays with shape as input, angles in radians
    """
    st = Staged(x, y, z)
    (xx, yy, zz), shape = _flat(st, x, y, z)
    n = xx.numel()
    r = st.out((n,)) if with_radius else None
    lat, lon = st.out((n,)), st.out((n,))
    st.ctx.call('amt_cartesian_to_spherical', ptr(xx), ptr(yy), ptr(zz), n, ptr(r), ptr(lat), ptr(lon))
    if with_radius:
        return st.result(r, shape), st.result(lat, shape), st.result(lon, shape)
    return st.result(lat, shape), st.result(lon, shape)


def geodetic2Ecef(lat, lon, h, a=wgs84A, b=wgs84B):
    """
    Converts geodetic to Earth Centered, Earth Fixed coordinates (reference transform.py:156-178).

    :param lat: latitude(s) in radians
    :param lon: longitude(s) in radians
    :param h: height (scalar, same unit as a and b)
    :rtype: tuple (x,y,z)
    """
    assert np.ndim(h) == 0, 'the height must be a scalar'
    scalar = np.ndim(lat) == 0 and not hasattr(lat, 'is_cuda')
    st = Staged(lat, lon)
    (la, lo), shape = _flat(st, np.atleast_1d(lat) if scalar else lat, np.atleast_1d(lon) if scalar else lon)
    n = la.numel()
    x, y, z = st.out((n,)), st.out((n,)), st.out((n,))
    st.ctx.call('amt_geodetic_to_ecef', ptr(la), ptr(lo), float(h), n, float(a), float(b), ptr(x), ptr(y), ptr(z))
    res = st.result(x, shape), st.result(y, shape), st.result(z, shape)
    return tuple(v[0] for v in res) if scalar else res


def geodetic2EcefZero(lat, lon, a=wgs84A, b=wgs84B):
    """Version of :func:`geodetic2Ecef` for `h=0` (reference transform.py:180-197)."""
    return geodetic2Ecef(lat, lon, 0.0, a, b)


def ecef2Geodetic(x, y, z, a=wgs84A, b=wgs84B):
    """
    Convert ECEF to geodetic coordinates with the Bowring algorithm from 1985
    (reference transform.py:199-297). The accuracy is at least 11 decimals (in degrees).

    :rtype: tuple (lat,lon) in radians
    """
    scalar = np.ndim(x) == 0 and not hasattr(x, 'is_cuda')
    st = Staged(x, y, z)
    if scalar:
        x, y, z = np.atleast_1d(x), np.atleast_1d(y), np.atleast_1d(z)
    (xx, yy, zz), shape = _flat(st, x, y, z)
    n = xx.numel()
    lat, lon = st.out((n,)), st.out((n,))
    st.ctx.call('amt_ecef_to_geodetic', ptr(xx), ptr(yy), ptr(zz), n, float(a), float(b), ptr(lat), ptr(lon))
    lat, lon = st.result(lat, shape), st.result(lon, shape)
    return (lat[0], lon[0]) if scalar else (lat, lon)


def rotatePole(lats, lons, altitude, angle=90, axis=[1, 0, 0]):
    """
    Rotates the given geodetic lat/lon coordinates around the origin (reference transform.py:301-322).

    :param lats, lons: shape (n,) in radians
    :param altitude: in km
    :param angle: degrees
    :param axis: [1, 0, 0], [0, 1, 0], or [0, 0, 1] for x y z axis
    :rtype: tuple (lats, lons) in radians
    """
    assert lats.ndim == 1 and lons.ndim == 1
    assert len(axis) == 3
    rot = rotation_matrix(np.deg2rad(angle), axis)[:3, :3]
    st = Staged(lats, lons)
    (la, lo), shape = _flat(st, lats, lons)
    n = la.numel()
    ola, olo = st.out((n,)), st.out((n,))
    st.ctx.call('amt_rotate_pole', host9(rot), ptr(la), ptr(lo), float(altitude), n, wgs84A, wgs84B,
                ptr(ola), ptr(olo))
    return st.result(ola, shape), st.result(olo, shape)


def _vecs(st, vecs):
    v = vecs if st.on_device else np.asarray(vecs, dtype=np.float64)
    assert v.ndim == 2 and v.shape[1] == 3
    return st.inp(v)


def j2000ToLatLon(j2000Vecs, time_):
    """
    Convert cartesian J2000 coordinates to geodetic coordinates (reference transform.py:324-343).

    :param j2000Vecs: shape (n,3)
    :param datetime time_:
    :rtype: tuple (latitudes, longitudes) in degrees
    """
    return _rotate_to_latlon(mat_j2000_to_geo(date2es(time_)), j2000Vecs)


def _rotate_to_latlon(mat, vecs):
    st = Staged(vecs)
    v = _vecs(st, vecs)
    n = v.shape[0]
    lat, lon = st.out((n,)), st.out((n,))
    st.ctx.call('amt_rotate_to_latlon', host9(mat), ptr(v), n, wgs84A, wgs84B, ptr(lat), ptr(lon))
    return st.result(lat), st.result(lon)


def latLonToJ2000(lat, lon, h, time_):
    """
    Convert geodetic coordinates to cartesian J2000 coordinates (reference transform.py:345-371).

    :param lat, lon: scalar or 1d-array, degrees
    :param h: scalar height
    :rtype: array (3,) or (3,n)
    """
    is_scalar = np.ndim(lat) == 0
    x, y, z = geodetic2Ecef(np.deg2rad(np.atleast_1d(lat)), np.deg2rad(np.atleast_1d(lon)), h)
    j2000 = geo_to_j2000(time_, np.asarray([x, y, z]).T).T
    return j2000[:, 0] if is_scalar else j2000


def smLonToMLT(smlons, out=None):
    """Convert solar magnetic longitudes [deg, -180..180] to magnetic local time [0,24] (transform.py:373-386)."""
    if out is not None:
        np.multiply(smlons, 24 / 360, out)
        mlt = out
    else:
        mlt = smlons * (24 / 360)
    mlt += 12
    return mlt


def mltToSmLon(mlt, out=None):
    """Convert magnetic local time [0,24] to solar magnetic longitudes [deg, -180..180] (transform.py:388-401)."""
    if out is not None:
        np.subtract(mlt, 12, out)
        smlon = out
    else:
        smlon = mlt - 12
    smlon /= (24 / 360)
    return smlon


def _rotate_to_mlat_mlt(mat, vecs):
    st = Staged(vecs)
    v = _vecs(st, vecs)
    n = v.shape[0]
    mlat, mlt = st.out((n,)), st.out((n,))
    st.ctx.call('amt_rotate_to_mlat_mlt', host9(mat), ptr(v), n, ptr(mlat), ptr(mlt))
    return st.result(mlat), st.result(mlt)


def j2000ToMLatMLT(j2000Vecs, time_):
    """
    Convert cartesian J2000 coordinates to MLat/MLT coordinates using the IGRF model
    (reference transform.py:403-430).

    :param j2000Vecs: shape (n,3)
    :rtype: tuple (mlat, mlt) in (degrees,hours)
    """
    return _rotate_to_mlat_mlt(mat_j2000_to_sm(date2es(time_)), j2000Vecs)


def geoToMLatMLT(geoVecs, time_):
    """
    Convert ECEF coordinates to MLat/MLT coordinates using the IGRF model (reference transform.py:432-459).

    :param geoVecs: shape (n,3)
    :rtype: tuple (mlat, mlt) in (degrees,hours)
    """
    return _rotate_to_mlat_mlt(mat_geo_to_sm(date2es(time_)), geoVecs)


def smToLatLon(smlats, smlons, time_):
    """
    Convert solar magnetic to geodetic coordinates using the IGRF model (reference transform.py:461-485).

    :param smlats: in degrees [-90,90]
    :param smlons: in degrees [-180,180]
    :rtype: tuple (latitudes, longitudes) in degrees
    """
    mat = mat_geo_to_sm(date2es(time_)).T
    st = Staged(smlats, smlons)
    (la, lo), shape = _flat(st, smlats, smlons)
    n = la.numel()
    lat, lon = st.out((n,)), st.out((n,))
    st.ctx.call('amt_sm_to_latlon', host9(mat), ptr(la), ptr(lo), n, wgs84A, wgs84B, ptr(lat), ptr(lon))
    return st.result(lat, shape), st.result(lon, shape)


def x_to_y(matFn, date, vecs, reverse=False):
    """reference transform.py:728-738: rotate (n,3) vectors by the matrix `matFn(date2es(date))`."""
    v = vecs if hasattr(vecs, 'is_cuda') else np.asarray(vecs, dtype=np.float64)
    assert v.ndim == 2
    assert v.shape[1] == 3
    mat = matFn(date2es(date))
    if reverse:
        mat = mat.T
    st = Staged(v)
    t = st.inp(v)
    out = st.out(t.shape)
    st.ctx.call('amt_rotate_vectors', host9(mat), ptr(t), t.shape[0], ptr(out))
    return st.result(out)


def j2000_to_geo(date, vecsJ2000):
    return x_to_y(mat_j2000_to_geo, date, vecsJ2000)


def geo_to_j2000(date, vecsGeo):
    return x_to_y(mat_j2000_to_geo, date, vecsGeo, reverse=True)


def j2000_to_sm(date, vecsJ2000):
    return x_to_y(mat_j2000_to_sm, date, vecsJ2000)


def geo_to_sm(date, vecsGEO):
    return x_to_y(mat_geo_to_sm, date, vecsGEO)


def sm_to_geo(date, vecsSM):
    return x_to_y(mat_geo_to_sm, date, vecsSM, reverse=True)


def gei_to_geo(date, vecsGEI):
    return x_to_y(mat_T1, date, vecsGEI)


def geo_to_gei(date, vecsGEO):
    return x_to_y(mat_T1, date, vecsGEO, reverse=True)


def gei_to_gse(date, vecsGEI):
    return x_to_y(mat_T2, date, vecsGEI)


def gse_to_gsm(date, vecsGSE):
    return x_to_y(mat_T3, date, vecsGSE)


def gsm_to_sm(date, vecsGSM):
    return x_to_y(mat_T4, date, vecsGSM)


# ---------------------------------------------------------------------------------------------
# host scalars (once per frame): time scale, IGRF dipole, cxform rotation matrices
# ---------------------------------------------------------------------------------------------
_J2000_EPOCH = datetime(2000, 1, 1, 12)


def julian_date(date):
    """
    UTC datetime -> Julian date as one double.  The reference obtains this number from
    ``astropy.time.Time(date, scale='utc').jd`` (transform.py:529; astropy >= 0.4.1): the sum of
    astropy's two-part JD, i.e. days since J2000.0 noon plus 2451545, without leap-second smearing
    on ordinary days.
    """
    delta = date - _J2000_EPOCH
    return 2451545.0 + (delta.days + (delta.seconds + delta.microseconds / 1e6) / 86400.0)


def date2es(date):
    """Converts UTC to ephemeris seconds past J2000 (reference transform.py:525-530)."""
    return (julian_date(date) - 2451545) * 86400


def rotation_matrix(angle, direction):
    """
    4x4 homogeneous matrix rotating by `angle` (radians) about `direction` through the origin
    (the part of the vendored transformations.py:295-336 the reference uses).
    """
    s, c = math.sin(angle), math.cos(angle)
    d0, d1, d2 = direction[:3]
    if abs(d0) + abs(d1) + abs(d2) == 1 and (d0 * d0 + d1 * d1 + d2 * d2) == 1:
        # a coordinate axis (all the frame transforms use these): the same floating-point operations as the
        # array expressions below, on scalars (per-frame host set-up is on the critical path of a sequence)
        a0, a1, a2 = d0 / 1.0, d1 / 1.0, d2 / 1.0
        k = 1.0 - c
        r00, r01, r02 = c + (a0 * a0) * k, 0.0 + (a0 * a1) * k, 0.0 + (a0 * a2) * k
        r10, r11, r12 = 0.0 + (a1 * a0) * k, c + (a1 * a1) * k, 0.0 + (a1 * a2) * k
        r20, r21, r22 = 0.0 + (a2 * a0) * k, 0.0 + (a2 * a1) * k, c + (a2 * a2) * k
        a0, a1, a2 = a0 * s, a1 * s, a2 * s
        return np.array([[r00 + 0.0, r01 + -a2, r02 + a1, 0.0],
                         [r10 + a2, r11 + 0.0, r12 + -a0, 0.0],
                         [r20 + -a1, r21 + a0, r22 + 0.0, 0.0],
                         [0.0, 0.0, 0.0, 1.0]])
    axis = np.array(direction[:3], dtype=np.float64)
    axis /= math.sqrt(np.dot(axis, axis))
    rot = np.diag([c, c, c])
    rot += np.outer(axis, axis) * (1.0 - c)
    axis *= s
    rot += np.array([[0.0, -axis[2], axis[1]],
                     [axis[2], 0.0, -axis[0]],
                     [-axis[1], axis[0], 0.0]])
    out = np.identity(4)
    out[:3, :3] = rot
    return out


# axis directions that make rotation_matrix agree with cxform's hapgood_matrix (transform.py:491-494)
X = [-1, 0, 0]
Y = [0, 1, 0]
Z = [0, 0, -1]


def _frac_year(et):
    idx = (et + 3155803200.0) / 157788000.0
    return idx, math.fmod(idx, 1.0)


def mag_lon(et):
    """Longitude of Earth's magnetic pole in radians (reference transform.py:497-508)."""
    idx, frac = _frac_year(et)
    return math.atan2(calcH11(idx, frac), calcG11(idx, frac)) + math.pi


def mag_lat(et):
    """Latitude of Earth's magnetic pole in radians (reference transform.py:510-523)."""
    idx, frac = _frac_year(et)
    g01, g11, h11 = calcG01(idx, frac), calcG11(idx, frac), calcH11(idx, frac)
    lam = mag_lon(et)
    return math.pi / 2 - math.atan((g11 * math.cos(lam) + h11 * math.sin(lam)) / g01)


def T0(et):
    """Julian centuries since 1 Jan 2000 12:00 (reference transform.py:534-538)."""
    return (et / 86400.0) / 36525.0


def H(et):
    """Time, in hours, since the preceding UT midnight (reference transform.py:540-551)."""
    jd = (et / 86400.0) - 0.5
    hh = (jd - int(jd)) * 24.0
    if hh < 0.0:
        hh += 24.0
    return hh


def lambda0(et):
    """Sun's ecliptic longitude in degrees (reference transform.py:553-560)."""
    M = 357.528 + 35999.050 * T0(et)
    lambd = 280.460 + 36000.772 * T0(et)
    return lambd + (1.915 - 0.0048 * T0(et)) * math.sin(np.deg2rad(M)) + 0.020 * math.sin(np.deg2rad(2 * M))


def epsilon(et):
    """The obliquity of the ecliptic in degrees (reference transform.py:562-566)."""
    return 23.439 - 0.013 * T0(et)


@functools.lru_cache(maxsize=16)     # (several matrices of one frame share it; callers do not modify the result)
def mat_P(et):
    """J2000 to GEI matrix (reference transform.py:568-581)."""
    t0 = T0(et)
    mat = rotation_matrix(np.deg2rad(-1.0 * (0.64062 * t0 + 0.00030 * t0 * t0)), Z)
    mat = np.dot(mat, rotation_matrix(np.deg2rad(0.55675 * t0 - 0.00012 * t0 * t0), Y))
    mat = np.dot(mat, rotation_matrix(np.deg2rad(-1.0 * (0.64062 * t0 + 0.00008 * t0 * t0)), Z))
    return mat[:3, :3]


@functools.lru_cache(maxsize=16)     # (several matrices of one frame share it; callers do not modify the result)
def mat_T1(et):
    """GEI to GEO matrix (reference transform.py:583-590)."""
    theta = 100.461 + 36000.770 * T0(et) + 360.0 * (H(et) / 24.0)
    return rotation_matrix(np.deg2rad(theta), Z)[:3, :3]


@functools.lru_cache(maxsize=16)     # (several matrices of one frame share it; callers do not modify the result)
def mat_T2(et):
    """GEI to GSE matrix (reference transform.py:592-599)."""
    mat = np.dot(rotation_matrix(np.deg2rad(lambda0(et)), Z), rotation_matrix(np.deg2rad(epsilon(et)), X))
    return mat[:3, :3]


@functools.lru_cache(maxsize=16)     # (several matrices of one frame share it; callers do not modify the result)
def vec_Qe(et):
    """Dipole axis in GSE (reference transform.py:601-620)."""
    lat, lon = mag_lat(et), mag_lon(et)
    Qg = [math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat)]
    return np.dot(np.dot(mat_T2(et), mat_T1(et).T), Qg)


def mat_T3(et):
    """GSE to GSM matrix (reference transform.py:622-629)."""
    Qe = vec_Qe(et)
    psi = math.atan2(np.deg2rad(Qe[1]), np.deg2rad(Qe[2]))
    return rotation_matrix(-psi, X)[:3, :3]


def mat_T4(et):
    """GSM to SM matrix (reference transform.py:631-638)."""
    Qe = vec_Qe(et)
    mu = math.atan2(np.deg2rad(Qe[0]), np.deg2rad(math.sqrt(Qe[1] * Qe[1] + Qe[2] * Qe[2])))
    return rotation_matrix(-mu, Y)[:3, :3]


def mat_T5(et):
    """GEO to MAG matrix (reference transform.py:640-647)."""
    mat = np.dot(rotation_matrix(mag_lat(et) - np.deg2rad(90.0), Y), rotation_matrix(mag_lon(et), Z))
    return mat[:3, :3]


def mat_j2000_to_geo(et):
    return np.dot(mat_T1(et), mat_P(et))                                   # transform.py:683-686


def mat_j2000_to_sm(et):
    return mat_T4(et).dot(mat_T3(et)).dot(mat_T2(et)).dot(mat_P(et))      # transform.py:688-691


def mat_geo_to_sm(et):
    return mat_T4(et).dot(mat_T3(et)).dot(mat_T2(et)).dot(mat_T1(et).T)   # transform.py:693-696


def northGeomagneticPoleLocation(date):
    """
    Approximate position of the north geomagnetic pole for the given date using the IGRF model
    (reference transform.py:740-753).

    :rtype: named (latitude, longitude) tuple, in degrees
    """
    et = date2es(date)
    lon = np.rad2deg(mag_lon(et))
    lon = lon - 360.0 * math.floor((lon + 180.0) / 360.0)
    return Location(np.rad2deg(mag_lat(et)), lon)


__all__ = ['spherical_to_cartesian', 'cartesian_to_spherical',
           'geodetic2Ecef', 'geodetic2EcefZero', 'ecef2Geodetic',
           'j2000ToLatLon', 'latLonToJ2000',
           'smLonToMLT', 'mltToSmLon', 'j2000ToMLatMLT', 'geoToMLatMLT', 'smToLatLon',
           'northGeomagneticPoleLocation']
