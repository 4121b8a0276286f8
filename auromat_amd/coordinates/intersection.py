"""
Line / sphere and line / ellipsoid-of-revolution intersections on the GPU.

Same functions, arguments and NaN-for-miss behaviour as the reference's
auromat/coordinates/intersection.py; its `_np` / `_ne` twins (intersection.py:58-142,165-227)
are replaced by the kernels behind ``amt_intersect_ellipsoid`` / ``amt_intersects_ellipsoid`` /
``amt_intersect_sphere``.
"""
import numpy as np

from .._native import host3
from .._ops import Staged, ptr


def _dirs(st, lineDirection):
    d = lineDirection if st.on_device else np.asarray(lineDirection, dtype=np.float64)
    single = d.ndim == 1
    assert d.shape[-1] == 3
    t = st.inp(d).reshape(-1, 3)
    return t, single


def sphereLineIntersection(sphereRadius, lineOrigin, lineDirection, directed=True):
    """
    Return the sphere-line intersection points (reference intersection.py:12-48).

    :param sphereRadius: radius of sphere with origin [0,0,0]
    :param lineOrigin: point, e.g. [1,2,3]
    :param lineDirection: unit vector or array of unit vectors
    :param bool directed: first intersection along the directed line (True) or the
                          intersection closest to the origin on the infinite line (False)
    :rtype: vector or array of vectors; NaN where there is no intersection
    """
    st = Staged(lineDirection)
    dirs, single = _dirs(st, lineDirection)
    out = st.out(dirs.shape)
    st.ctx.call('amt_intersect_sphere', float(sphereRadius), host3(lineOrigin), ptr(dirs), dirs.shape[0],
                1 if directed else 0, ptr(out))
    res = st.result(out)
    return res[0] if single else res


def ellipsoidLineIntersection(a, b, lineOrigin, lineDirection, directed=True):
    """
    Return the ellipsoid-line intersection points (reference intersection.py:144-163).

    :note: The ellipsoid is assumed to be at (0,0,0).
    :param a: equatorial axis of the ellipsoid of revolution
    :param b: polar axis of the ellipsoid of revolution
    :param lineOrigin: x,y,z vector
    :param lineDirection: x,y,z array of vectors (n,3); not required to be unit vectors
    :param bool directed: see :func:`sphereLineIntersection`
    :rtype: array of vectors (n,3); NaN rows for misses, never raises for them
    """
    st = Staged(lineDirection)
    dirs, _ = _dirs(st, lineDirection)
    out = st.out(dirs.shape)
    st.ctx.call('amt_intersect_ellipsoid', float(a), float(b), host3(lineOrigin), ptr(dirs), dirs.shape[0],
                1 if directed else 0, ptr(out))
    return st.result(out)


def ellipsoidLineIntersects(a, b, lineOrigin, lineDirection, directed=True):
    """
    As :func:`ellipsoidLineIntersection` but returns an array of booleans instead
    of the intersection points (reference intersection.py:229-237).
    """
    import torch
    st = Staged(lineDirection)
    dirs, _ = _dirs(st, lineDirection)
    out = st.out((dirs.shape[0],), torch.uint8)
    st.ctx.call('amt_intersects_ellipsoid', float(a), float(b), host3(lineOrigin), ptr(dirs), dirs.shape[0],
                1 if directed else 0, ptr(out))
    if st.on_device:
        return out.bool()
    return st.result(out).astype(bool)


# name used by BASELINE.json's north star for the same operation
intersectEllipsoidLineOfSight = ellipsoidLineIntersection

__all__ = ['sphereLineIntersection', 'ellipsoidLineIntersection', 'ellipsoidLineIntersects',
           'intersectEllipsoidLineOfSight']
