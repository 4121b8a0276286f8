"""
TAN (gnomonic) WCS camera model: pixel grid -> celestial unit vectors or RA/Dec.

Mirror of the reference's auromat/coordinates/wcs.py.  The reference walks through native
spherical angles (atan2/atan) and back (cos/sin) in NumPy (wcs.py:108-142); the kernel behind
``amt_directions_tan`` evaluates the same projection algebraically,
``(-Y, X, 180/pi) / sqrt(X^2 + Y^2 + (180/pi)^2)`` with ``(X, Y) = CD (p - CRPIX + 1)``, followed by
the same native->celestial rotation, which agrees to ~1e-16 per component.
Only TAN headers are supported (the reference falls back to astropy's WCS for others).
"""
import ctypes as C
import math

import numpy as np

from .._native import FrameParams
from .._ops import Staged, ptr
from .transform import cartesian_to_spherical


def euler_matrix_rzxz(ai, aj, ak):
    """
    3x3 rotation for Euler angles about the rotating z, x, z axes — the
    ``euler_matrix(ai, aj, ak, 'rzxz')[:3,:3]`` of the vendored transformations.py:1042-1102
    that reference wcs.py:139 uses.
    """
    ai, ak = ak, ai                      # rotating frame: first and last angle swap
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs = ci * ck, ci * sk
    sc, ss = si * ck, si * sk
    i, j, k = 2, 0, 1                    # z, x, (y) for first axis 2, even parity
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


ZENITHAL = ('TAN', 'SIN', 'ARC', 'STG', 'ZEA')


def projection_of(header):
    """(projection code, has SIP terms) of a celestial WCS header, e.g. ('TAN', False) for RA---TAN / DEC--TAN."""
    c1, c2 = str(header['CTYPE1']), str(header['CTYPE2'])
    sip = c1.endswith('-SIP') and c2.endswith('-SIP')
    if sip:
        c1, c2 = c1[:-4], c2[:-4]
    if not (c1.startswith('RA--') and c2.startswith('DEC-') and c1[-3:] == c2[-3:]):
        raise NotImplementedError('unsupported CTYPE pair %r / %r' % (header['CTYPE1'], header['CTYPE2']))
    return c1[-3:], sip


def is_plain_tan(header):
    """The headers the fused kernels take (reference wcs.py:50-52: its fast path)."""
    return header['CTYPE1'] == 'RA---TAN' and header['CTYPE2'] == 'DEC--TAN' and header.get('LATPOLE', 0.0) == 0.0


def zenithal_params(header, width, height, startX=0, startY=0, corner=True):
    """The amt_zenithal_wcs block of a zenithal (+ SIP) header: what :func:`zenithal_pix2world` evaluates, for the device
    generator ``amt_directions_zenithal``."""
    from .._native import SIP_MAX, ZenithalWcs
    proj, sip = projection_of(header)
    if proj not in ZENITHAL:
        raise NotImplementedError('projection %s is not supported (zenithal projections only: %s)' % (proj, ', '.join(ZENITHAL)))
    if header.get('LATPOLE', 0.0) not in (0.0, 90.0, header['CRVAL2']):
        raise NotImplementedError('LATPOLE = %r' % header.get('LATPOLE'))
    w = ZenithalWcs()
    w.width, w.height, w.corner, w.projection = int(width), int(height), 1 if corner else 0, ZENITHAL.index(proj)
    w.cd[:] = [header['CD1_1'], header['CD1_2'], header['CD2_1'], header['CD2_2']]
    w.crpix[:] = [header['CRPIX1'], header['CRPIX2']]
    w.rot[:] = list(euler_matrix_rzxz(np.deg2rad(header['CRVAL1'] + 90), np.deg2rad(90 - header['CRVAL2']),
                                      np.deg2rad(-(header.get('LONPOLE', 180.0) - 90))).ravel())
    w.start_x, w.start_y = float(startX), float(startY)
    if sip:
        for prefix, name in (('A', 'sip_a'), ('B', 'sip_b')):
            order = int(header.get(prefix + '_ORDER', 0))
            if order >= SIP_MAX:
                raise NotImplementedError('SIP order %d' % order)
            setattr(w, 'sip_order_' + prefix.lower(), order)
            table = getattr(w, name)
            for p_ in range(order + 1):
                for q_ in range(order + 1 - p_):
                    table[p_][q_] = float(header.get('%s_%d_%d' % (prefix, p_, q_)) or 0.0)
    return w


def zenithal_directions_device(header, width, height, startX=0, startY=0, corner=True):
    """:func:`zenithal_pix2world` on the device (``amt_directions_zenithal``): a torch tensor (height[+1], width[+1], 3) that
    :class:`auromat_amd.mapping.astrometry.DirectionArrayMapping` takes as it is — a 12 Mpx frame in a fraction of a
    millisecond where the NumPy generator takes seconds."""
    from .._native import Context
    ctx = Context.current()
    w = zenithal_params(header, width, height, startX, startY, corner)
    out = ctx.empty((w.height + w.corner, w.width + w.corner, 3))
    ctx.call('amt_directions_zenithal', C.byref(w), ptr(out))
    return out


def zenithal_pix2world(header, width, height, startX=0, startY=0, corner=True):
    """
    Unit direction vectors (J2000 / ICRS cartesian) of a pixel rectangle for the zenithal projections TAN, SIN (without
    its slant parameters), ARC, STG and ZEA, with SIP distortion polynomials when the CTYPEs end in -SIP — the headers the
    reference hands to ``astropy.wcs.WCS(header).all_pix2world`` (wcs.py:54-56; astropy / wcslib are absent here:
    restated from Calabretta & Greisen 2002, A&A 395, sections 5.1.1-5.1.7, and the SIP convention of Shupe et al.
    2005).  Host NumPy, row block by row block; the result feeds :class:`DirectionArrayMapping`
    (``amt_georef_frame_dirs``), like any other camera model.  For a plain TAN header it equals the fused kernels'
    own camera model to rounding.

    :rtype: float64 array (height + 1, width + 1, 3) if corner else (height, width, 3)
    """
    proj, sip = projection_of(header)
    if proj not in ZENITHAL:
        raise NotImplementedError('projection %s is not supported (zenithal projections only: %s)' % (proj, ', '.join(ZENITHAL)))
    if header.get('LATPOLE', 0.0) not in (0.0, 90.0, header['CRVAL2']):
        raise NotImplementedError('LATPOLE = %r' % header.get('LATPOLE'))
    cd = np.array([[header['CD1_1'], header['CD1_2']], [header['CD2_1'], header['CD2_2']]], dtype=np.float64)
    rot = euler_matrix_rzxz(np.deg2rad(header['CRVAL1'] + 90), np.deg2rad(90 - header['CRVAL2']),
                            np.deg2rad(-(header.get('LONPOLE', 180.0) - 90)))
    off = -0.5 if corner else 0.0
    n_rows, n_cols = height + (1 if corner else 0), width + (1 if corner else 0)
    # 0-based pixel coordinates -> offsets from the reference pixel (CRPIX is 1-based: reference wcs.py:93-99)
    u_all = np.arange(n_cols, dtype=np.float64) + (startX + off) - header['CRPIX1'] + 1
    out = np.empty((n_rows, n_cols, 3), dtype=np.float64)

    def sip_poly(prefix, u, v):
        order = int(header.get(prefix + '_ORDER', 0))
        f = np.zeros_like(u)
        for p_ in range(order + 1):
            for q_ in range(order + 1 - p_):
                c = header.get('%s_%d_%d' % (prefix, p_, q_))
                if c:
                    f = f + c * u ** p_ * v ** q_
        return f

    k = 180.0 / np.pi
    block = max(1, (1 << 21) // max(n_cols, 1))
    for r0 in range(0, n_rows, block):
        r1 = min(n_rows, r0 + block)
        v = (np.arange(r0, r1, dtype=np.float64) + (startY + off) - header['CRPIX2'] + 1)[:, None]
        u = np.broadcast_to(u_all[None, :], (r1 - r0, n_cols))
        v = np.broadcast_to(v, (r1 - r0, n_cols))
        if sip:
            u, v = u + sip_poly('A', u, v), v + sip_poly('B', u, v)
        x = cd[0, 0] * u + cd[0, 1] * v                     # intermediate world coordinates, degrees
        y = cd[1, 0] * u + cd[1, 1] * v
        r = np.sqrt(x * x + y * y)
        phi = np.arctan2(x, -y)
        with np.errstate(invalid='ignore'):
            if proj == 'TAN':
                theta = np.arctan2(k, r)
            elif proj == 'SIN':
                theta = np.arccos(r / k)
            elif proj == 'ARC':
                theta = np.deg2rad(90.0 - r)
            elif proj == 'STG':
                theta = np.deg2rad(90.0) - 2 * np.arctan(r / (2 * k))
            else:                                               # ZEA
                theta = np.deg2rad(90.0) - 2 * np.arcsin(r / (2 * k))
        ct = np.cos(theta)
        native = np.stack((ct * np.cos(phi), ct * np.sin(phi), np.sin(theta)), axis=-1)
        out[r0:r1] = native.dot(rot.T)
    return out


def check_tan_header(header):
    if not (header['CTYPE1'] == 'RA---TAN' and header['CTYPE2'] == 'DEC--TAN' and header['LATPOLE'] == 0.0):
        raise NotImplementedError('only TAN projections with LATPOLE=0 are supported '
                                  '(the reference uses astropy.wcs for other projections)')


def celestial_rotation(header):
    """Native -> celestial spherical rotation (reference wcs.py:133-139)."""
    euler_z = header['CRVAL1'] + 90
    euler_x = 90 - header['CRVAL2']
    euler_z2 = -(header['LONPOLE'] - 90)
    return euler_matrix_rzxz(np.deg2rad(euler_z), np.deg2rad(euler_x), np.deg2rad(euler_z2))


def fill_wcs_params(params, header, width=None, height=None, startX=0, startY=0):
    """Copy the WCS cards the kernels need into an amt_frame_params block."""
    check_tan_header(header)
    params.width = int(header['IMAGEW'] if width is None else width)
    params.height = int(header['IMAGEH'] if height is None else height)
    params.cd[:] = [header['CD1_1'], header['CD1_2'], header['CD2_1'], header['CD2_2']]
    # a rectangle starting at (startX, startY) is the same as moving the reference pixel
    params.crpix[:] = [header['CRPIX1'] - startX, header['CRPIX2'] - startY]
    params.rot[:] = list(celestial_rotation(header).ravel())
    return params


def _to_radec(st, vec, shape):
    import torch
    flat = vec.reshape(-1, 3)
    x, y, z = (flat[:, i].contiguous() for i in range(3))
    dec, ra = cartesian_to_spherical(x, y, z, with_radius=False)
    dec = torch.rad2deg(dec)
    ra = torch.remainder(torch.rad2deg(ra) - 360, 360)     # wrap into [0,360) (reference wcs.py:148-152)
    return st.result(ra, shape), st.result(dec, shape)


def pix2world(wcsHeader, width, height, startX=0, startY=0, corner=True, ascartesian=False, device=None):
    """
    Calculate RA, Dec coordinates of a given pixel coordinate rectangle (reference wcs.py:18-64).

    Each array element contains the RA,Dec coords of the top left corner of the
    given pixel if corner==True, otherwise the coords of the pixel center.
    If corner==True, an additional row and column exists at the bottom and right.

    If ascartesian=False: tuple(ra, dec), arrays of shape (height+1,width+1) if corner else (height,width).
    If ascartesian=True: array of shape (height[+1],width[+1],3) with x,y,z order.
    With ``device`` given the results stay on that device as torch tensors.
    """
    params = fill_wcs_params(FrameParams(), wcsHeader, width, height, startX, startY)
    corner = 1 if corner else 0
    st = Staged()
    if device is not None:
        st.on_device = True
    shape = (params.height + corner, params.width + corner)
    out = st.out(shape + (3,))
    st.ctx.call('amt_directions_tan', C.byref(params), corner, ptr(out))
    if ascartesian:
        return st.result(out)
    keep = st.on_device
    st.on_device = True
    ra, dec = _to_radec(st, out, shape)
    st.on_device = keep
    return st.result(ra), st.result(dec)


def tan_pix2world(header, px, py, origin, ascartesian=False):
    """
    TAN-only equivalent of astropy.wcs.WCS.wcs_pix2world for arbitrary pixel coordinates
    (reference wcs.py:66-157).

    :rtype: tuple (ra,dec) in degrees, or cartesian coordinates in one array (...,3) if ascartesian=True
    """
    assert origin in [0, 1]
    st = Staged(px, py)
    x = st.inp(px)
    y = st.inp(py)
    assert x.shape == y.shape
    shape = tuple(x.shape)
    params = fill_wcs_params(FrameParams(), dict(header, IMAGEW=header.get('IMAGEW', 1),
                                                 IMAGEH=header.get('IMAGEH', 1)))
    out = st.out((x.numel(), 3))
    st.ctx.call('amt_directions_tan_points', C.byref(params), ptr(x.reshape(-1)), ptr(y.reshape(-1)), x.numel(),
                origin, ptr(out))
    if ascartesian:
        return st.result(out, shape + (3,))
    keep = st.on_device
    st.on_device = True
    ra, dec = _to_radec(st, out, shape)
    st.on_device = keep
    return st.result(ra), st.result(dec)
