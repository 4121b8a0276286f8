"""
Mappings that derive their coordinates from a camera position and a WCS solution
(reference auromat/mapping/astrometry.py).

The reference chains five lazily cached NumPy stages per frame (pixel directions -> ellipsoid
intersection -> J2000->GEO -> Bowring -> elevation, astrometry.py:49-212).  Here the first access
to any coordinate property launches ONE fused kernel (``amt_georef_frame``) that produces corner
lat/lon, centre lat/lon and elevation together; MLat/MLT is a second launch of the same kernel
with only the magnetic outputs enabled.
"""
from __future__ import division

import ctypes as C

import numpy as np

from ..coordinates.geodesic import wgs84A, wgs84B
from ..coordinates.transform import date2es, mat_j2000_to_geo, mat_j2000_to_sm
from ..coordinates.wcs import check_tan_header, fill_wcs_params, pix2world
from ..frame import FrameData
from .._native import Context, FrameParams, GeorefOut, RunFrame, ptr
from .mapping import BaseMapping, GenericMapping, inflatedEarthIntersection


def run_frame(wcsHeader, cameraPosGCRS, photoTime, altitude=0.0, img_ptr=None, out=None, img_host_ptr=None):
    """The amt_run_frame of one frame: the WCS cards, the camera and the photo time as the native side takes them."""
    from ..coordinates.transform import julian_date
    check_tan_header(wcsHeader)
    f = RunFrame() if out is None else out
    f.crval[0], f.crval[1] = wcsHeader['CRVAL1'], wcsHeader['CRVAL2']
    f.crpix[0], f.crpix[1] = wcsHeader['CRPIX1'], wcsHeader['CRPIX2']
    f.cd[0], f.cd[1], f.cd[2], f.cd[3] = wcsHeader['CD1_1'], wcsHeader['CD1_2'], wcsHeader['CD2_1'], wcsHeader['CD2_2']
    f.lonpole = wcsHeader['LONPOLE']
    f.cam[0], f.cam[1], f.cam[2] = float(cameraPosGCRS[0]), float(cameraPosGCRS[1]), float(cameraPosGCRS[2])
    f.jd = julian_date(photoTime)
    f.altitude = float(altitude)
    f.img = img_ptr
    f.img_host = img_host_ptr
    return f


def frame_params(wcsHeader, altitude, cameraPosGCRS, photoTime, fastCenterCalculation, magnetic=True):
    """
    The amt_frame_params block of one frame: WCS cards + camera + per-frame rotation matrices (reference wcs.py:133-139,
    transform.py:525-696), computed by the library's host function amt_frame_params_from_wcs — the one the native
    sequence runner uses, so that every path sees the same bits; it equals the Python functions of
    auromat_amd.coordinates.transform, which are pinned to the reference's doubles (tests/test_host_cpu.py).
    magnetic=False skips the J2000->SM matrix (IGRF dipole, four more rotations) when no MLat/MLT
    output is requested.
    """
    from .. import _native
    f = run_frame(wcsHeader, cameraPosGCRS, photoTime, altitude)
    p = FrameParams()
    rc = _native.lib().amt_frame_params_from_wcs(C.byref(f), int(wcsHeader['IMAGEW']), int(wcsHeader['IMAGEH']),
                                                 1 if fastCenterCalculation else 0, float(altitude), 1 if magnetic else 0,
                                                 C.byref(p))
    if rc != 0:
        from ..coordinates.igrf import IGRF_DEFINED_UNTIL_YEAR
        raise ValueError("ERROR: Specified year is greater than IGRF implementation (" + str(IGRF_DEFINED_UNTIL_YEAR) +
                         "), please update coefficients in auromat_amd.coordinates.igrf module")
    return p


def frame_params_python(wcsHeader, altitude, cameraPosGCRS, photoTime, fastCenterCalculation, magnetic=True):
    """:func:`frame_params` through the Python functions (what the native host function is checked against)."""
    p = fill_wcs_params(FrameParams(), wcsHeader)
    p.fast_center = 1 if fastCenterCalculation else 0
    p.cam[:] = [float(v) for v in cameraPosGCRS]
    p.a, p.b = wgs84A + altitude, wgs84B + altitude          # mapping.py:1498-1501
    p.a0, p.b0 = wgs84A, wgs84B                              # transform.py:338
    et = date2es(photoTime)
    p.m_geo[:] = list(mat_j2000_to_geo(et).ravel())
    if magnetic:
        p.m_sm[:] = list(mat_j2000_to_sm(et).ravel())
    return p


def pole_in_view(params, min_elevation=None, magnetic=False):
    """
    Host-side pole test for camera mappings: is the geographic (or, with magnetic=True, the SM) north
    or south pole of the mapping shell imaged by a valid pixel?  The pole point is projected through
    the inverse TAN model; it counts when it falls inside the frame, is the *first* hit of its ray
    and lies above the elevation threshold.  Replaces, for frames with a known camera model, the
    outline-based geodesic test of the reference (geodesic.py:183, mapping.py:705-721) at zero
    per-pixel cost.  Returns +1 (north), -1 (south) or 0.
    """
    rot = np.array(params.rot[:]).reshape(3, 3)
    cd = np.array(params.cd[:]).reshape(2, 2)
    cam = np.array(params.cam[:])
    m = np.array(params.m_sm[:] if magnetic else params.m_geo[:]).reshape(3, 3)
    a, b = params.a, params.b
    scale = np.array([1 / a, 1 / a, 1 / b])
    for sign in (1, -1):
        u = m.T.dot([0.0, 0.0, float(sign)])                 # pole axis in J2000
        pole = u / np.sqrt(np.sum((u * scale) ** 2))          # point of the shell on that axis
        los = pole - cam
        dist = np.sqrt(los.dot(los))
        d = los / dist
        # first intersection of the ray with the shell (same quadratic as intersection.py:58-104)
        ds, os_ = d * scale, -cam * scale
        d_o, d_d, o_o = ds.dot(os_), ds.dot(ds), os_.dot(os_)
        disc = d_o * d_o - o_o * d_d + d_d
        if disc < 0:
            continue
        inside = np.sum((cam * scale) ** 2) < 1
        t = ((d_o + np.sqrt(disc)) if inside else (d_o - np.sqrt(disc))) / d_d
        if abs(t - dist) > 1e-6 * dist:
            continue                                           # the pole is on the far side
        v = rot.T.dot(d)                                       # native (projection) frame
        if v[2] <= 0:
            continue
        k = 180 / np.pi
        px, py = np.linalg.solve(cd, [k * v[1] / v[2], -k * v[0] / v[2]])
        x, y = px + params.crpix[0] - 1, py + params.crpix[1] - 1
        if not (-0.5 <= x <= params.width - 0.5 and -0.5 <= y <= params.height - 0.5):
            continue
        if min_elevation is not None:
            elev = np.rad2deg(np.arcsin(np.clip(-d.dot(pole) / np.sqrt(pole.dot(pole)), -1, 1)))
            if not elev >= min_elevation:
                continue
        return sign
    return 0


def georef_into(fd, params, geo=True, mag=False, bbox_min_elevation=None, dirs=None):
    """Launch the fused kernel writing the requested output groups into `fd` (allocating them).
    `dirs`: device tensor (h+1, w+1, 3) of corner directions in J2000 instead of the TAN camera model."""
    h, w = fd.height, fd.width
    out = GeorefOut()
    ctx = fd.ctx
    if geo:
        fd.lat, fd.lon = ctx.empty((h + 1, w + 1)), ctx.empty((h + 1, w + 1))
        fd.lat_c, fd.lon_c, fd.elev = ctx.empty((h, w)), ctx.empty((h, w)), ctx.empty((h, w))
        out.lat, out.lon, out.lat_c, out.lon_c, out.elev = (t.data_ptr() for t in
                                                            (fd.lat, fd.lon, fd.lat_c, fd.lon_c, fd.elev))
    if mag:
        fd.mlat, fd.mlt = ctx.empty((h + 1, w + 1)), ctx.empty((h + 1, w + 1))
        fd.mlat_c, fd.mlt_c = ctx.empty((h, w)), ctx.empty((h, w))
        out.mlat, out.mlt, out.mlat_c, out.mlt_c = (t.data_ptr() for t in (fd.mlat, fd.mlt, fd.mlat_c, fd.mlt_c))
    if bbox_min_elevation is not None:
        fd.bbox = ctx.empty((8,))
        out.bbox = fd.bbox.data_ptr()
        out.bbox_min_elevation = float(bbox_min_elevation)
    if dirs is not None:
        ctx.call('amt_georef_frame_dirs', C.byref(params), ptr(dirs), C.byref(out))
    else:
        ctx.call('amt_georef_frame', C.byref(params), C.byref(out))
    return fd


class BaseAstrometryMapping(BaseMapping):
    """
    A mapping which calculates its coordinates based on the camera position and its WCS definition
    (reference astrometry.py:18-218).
    """

    def __init__(self, wcsHeader, alti, cameraPosGCRS, photoTime, identifier, metadata={},
                 fastCenterCalculation=False):
        """
        :param alti: mapping altitude in km
        :param fastCenterCalculation: centre coordinates from the mean of the four corner
                                      intersection points instead of an own ray cast per centre
        """
        BaseMapping.__init__(self, alti, cameraPosGCRS, photoTime, identifier, metadata)
        self._wcsHeader = wcsHeader
        self.fastCenterCalculation = fastCenterCalculation
        self._frame = None
        self._img_array = None
        # maskedByElevation(e) of a mapping whose arrays do not exist yet is remembered, not computed: resample() of
        # such a mapping runs the single-pass plan (one kernel: georeferencing, mask, box, binning), and anything else
        # that asks for the arrays materialises them, masks included, on first use (see frame())
        self._lazy_elev = None

    @property
    def wcsHeader(self):
        return self._wcsHeader

    def _params(self):
        return frame_params(self._wcsHeader, self.altitude, self.cameraPosGCRS, self.photoTime,
                            self.fastCenterCalculation)

    def frame(self):
        if self._frame is None:
            ctx = Context.current()
            hdr = self._wcsHeader
            fd = FrameData(ctx, hdr['IMAGEH'], hdr['IMAGEW'])
            georef_into(fd, self._params(), geo=True)
            if self._img_array is not None:
                fd.set_image(self._img_array, lazy=True)
            if not self.fastCenterCalculation:
                # exact centres carry their own misses: reconcile corner and centre masks
                # (sanitize_data, reference mapping.py:1063-1125); fast centres are consistent
                # by construction (astrometry.py:35-40)
                ctx.call('amt_sanitize_masks', ptr(fd.corner_mask_tensor()), ptr(fd.center_mask_tensor()), None,
                         fd.height, fd.width, 0)
            self._frame = fd
            self._apply_lazy_elevation()
        return self._frame

    def _apply_lazy_elevation(self):
        if self._lazy_elev is not None:
            # the remembered maskedByElevation (raises the reference's ValueError when nothing is left)
            e, self._lazy_elev = self._lazy_elev, None
            try:
                self._frame = BaseMapping.maskedByElevation(self, e)._frame
            except ValueError:
                self._frame, self._lazy_elev = None, e
                raise
            self.setDirty()

    def maskedByElevation(self, minElevation=10):
        """
        BaseMapping.maskedByElevation (reference mapping.py:845-864).  While the arrays of this mapping have not been
        asked for, the mask is only remembered: ``resample(mapping.maskedByElevation(e), pxPerDeg=...)`` — the call of the
        reference's user guide — then costs one launch of the fused kernel instead of the array pipeline, and every
        other use computes arrays and masks on first access.  The reference's ``ValueError`` for a threshold that masks
        every pixel is raised at that first use (``resample`` included) instead of here.
        """
        if self._frame is None and self._lazy_elev is None and self._fusable():
            import copy
            m = copy.copy(self)
            m._lazy_elev = float(minElevation)
            m.setDirty()
            return m
        return BaseMapping.maskedByElevation(self, minElevation)

    def _fusable(self):
        img = self._img_array
        return img is not None and getattr(img, 'ndim', 0) == 3 and img.shape[2] == 3 and img.dtype in (np.uint8, np.uint16)

    def _fused_resample(self, pxPerDeg, containsPole=None, magnetic=False, arcsecPerPx=None):
        """
        resample() / resampleMLatMLT() of this mapping through the single-pass plan (FramePipeline.run(fuse=True): the
        binning inside the georeferencing kernel, no per-pixel array is written or read back) -> the result dict of
        resample_frame, or None when that does not apply (arrays already exist, masks other than one
        maskedByElevation, no RGB image).  A frame the plan does not cover takes the two-pass plan inside run().
        """
        if self._frame is not None or not self._fusable():
            return None
        from ..pipeline import fused_class_pipeline
        hdr = self._wcsHeader
        pipe = fused_class_pipeline(hdr['IMAGEW'], hdr['IMAGEH'], self._img_array.dtype, magnetic)
        params = None
        if arcsecPerPx:
            # the reference's own call form (cli/convert.py:176-185, test/mapping_test.py:24-42): px/deg from this frame's
            # bounding box — the box-first plan (FramePipeline.run).  With a pole in view the reference's
            # plateCarreeResolution has no longitude resolution to offer: that case stays with the array route
            params = frame_params(hdr, self.altitude, self.cameraPosGCRS, self.photoTime, self.fastCenterCalculation,
                                  magnetic=bool(magnetic))
            if pole_in_view(params, self._lazy_elev, magnetic=bool(magnetic)):
                return None
        res = pipe.run(hdr, self.altitude, self.cameraPosGCRS, self.photoTime, img=self._img_array,
                       fast=self.fastCenterCalculation, min_elevation=self._lazy_elev, pxPerDeg=pxPerDeg,
                       containsPole=containsPole, magnetic=magnetic, fuse=True, arcsecPerPx=arcsecPerPx, params=params)
        res['plan'] = pipe.last_plan
        return res

    def _mlatmlt_tensors(self, center):
        """
        Overrides BaseMapping: J2000 intersection points go straight to SM (reference
        astrometry.py:170-198) — one more launch of the fused kernel, magnetic outputs only.
        """
        fd = self.frame()
        if fd.mlat is None:
            georef_into(fd, self._params(), geo=False, mag=True)
        return (fd.mlat_c, fd.mlt_c) if center else (fd.mlat, fd.mlt)

    # -- debugging helpers of the reference ---------------------------------------------------
    @property
    def cameraToPixelCornerDirection(self):
        """Direction vector for each pixel corner, (h+1, w+1, 3)."""
        return self._cached('dir_corner', lambda: pixelDirection(self._wcsHeader, corner=True))

    @property
    def cameraToPixelCenterDirection(self):
        """Direction vector for each pixel center, (h, w, 3)."""
        def make():
            if self.fastCenterCalculation:
                return self._calcCenters(self.cameraToPixelCornerDirection)
            return pixelDirection(self._wcsHeader, corner=False)
        return self._cached('dir_center', make)

    @property
    def intersectionInflatedCorner(self):
        """Point of intersection with the inflated earth for each pixel corner, (h+1, w+1, 3)."""
        def make():
            d = self.cameraToPixelCornerDirection
            return inflatedEarthIntersection(d.reshape(-1, 3), self.cameraPosGCRS, self.altitude).reshape(d.shape)
        return self._cached('p_corner', make)

    @property
    def intersectionInflatedCenter(self):
        """Point of intersection with the inflated earth for each pixel center, (h, w, 3)."""
        def make():
            if self.fastCenterCalculation:
                with np.errstate(invalid='ignore'):
                    return self._calcCenters(self.intersectionInflatedCorner)
            d = self.cameraToPixelCenterDirection
            return inflatedEarthIntersection(d.reshape(-1, 3), self.cameraPosGCRS, self.altitude).reshape(d.shape)
        return self._cached('p_center', make)

    @property
    def distance(self):
        """Distance for each pixel center between camera and intersection point, (h, w) in km.  For debugging purposes
        only! (reference astrometry.py:108-116)"""
        from ..utils import vectorLengths
        p = self.intersectionInflatedCenter
        return vectorLengths((p - np.asarray(self.cameraPosGCRS)).reshape(-1, 3)).reshape(p.shape[0], p.shape[1])

    @staticmethod
    def _calcCenters(corners):
        centers = corners[:-1, :-1] + corners[:-1, 1:]
        centers += corners[1:, 1:]
        centers += corners[1:, :-1]
        centers /= 4
        return centers

    @property
    def ra(self):
        """Right ascension for each pixel corner. For debugging purposes only!"""
        ra, _ = pix2world(self._wcsHeader, self._wcsHeader['IMAGEW'], self._wcsHeader['IMAGEH'])
        return ra

    @property
    def dec(self):
        """Declination for each pixel corner. For debugging purposes only!"""
        _, dec = pix2world(self._wcsHeader, self._wcsHeader['IMAGEW'], self._wcsHeader['IMAGEH'])
        return dec

    def createResampled(self, lats, lons, latsCenter, lonsCenter, elevation, img):
        return GenericMapping(lats, lons, latsCenter, lonsCenter, elevation, self.altitude, img,
                              self.cameraPosGCRS, self.photoTime, self.identifier, metadata=self.metadata)


class DirectionArrayMapping(BaseAstrometryMapping):
    """
    A mapping whose camera model is given as an array of line-of-sight unit vectors, one per pixel corner, in
    J2000 / GCRS — the hook for camera models other than a TAN WCS: all-sky calibrations (reference
    miracle.py:196-258 builds such vectors from azimuth / elevation tables), SIP-distorted or non-TAN WCS solutions
    evaluated by another library, THEMIS-style reprojections.  Everything downstream of the direction generator
    is the same kernel (``amt_georef_frame_dirs``: shell intersection, geodetic and geomagnetic coordinates,
    elevation, masks); centres are the mean of their four corner hits (the reference's fast mode, which is also
    what miracle.py:139-160 does).
    """

    def __init__(self, cornerDirections, alti, img, cameraPosGCRS, photoTime, identifier, metadata=None):
        """`cornerDirections`: (h + 1, w + 1, 3) array, or a float64 device tensor of that shape (e.g. from
        ``coordinates.wcs.zenithal_directions_device``), which is used where it lies."""
        img = np.asarray(img)
        on_device = hasattr(cornerDirections, 'is_cuda') and cornerDirections.is_cuda
        d = cornerDirections.contiguous() if on_device else np.ascontiguousarray(cornerDirections, dtype=np.float64)
        assert d.ndim == 3 and d.shape[2] == 3 and img.ndim == 3
        assert tuple(d.shape[:2]) == (img.shape[0] + 1, img.shape[1] + 1), 'one direction per pixel corner'
        hdr = {'IMAGEW': img.shape[1], 'IMAGEH': img.shape[0]}
        BaseAstrometryMapping.__init__(self, hdr, alti, cameraPosGCRS, photoTime, identifier, metadata or {},
                                       fastCenterCalculation=True)
        self._dirs = None if on_device else d
        self._dirs_dev = d if on_device else None
        self._img_array = img

    def _params(self):
        p = FrameParams()
        p.width, p.height, p.fast_center = self._wcsHeader['IMAGEW'], self._wcsHeader['IMAGEH'], 1
        p.cd[:] = [1.0, 0.0, 0.0, 1.0]                     # the camera-model block is not used with directions
        p.rot[:] = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]
        p.cam[:] = [float(v) for v in self.cameraPosGCRS]
        p.a, p.b = wgs84A + self.altitude, wgs84B + self.altitude
        p.a0, p.b0 = wgs84A, wgs84B
        et = date2es(self.photoTime)
        p.m_geo[:] = list(mat_j2000_to_geo(et).ravel())
        p.m_sm[:] = list(mat_j2000_to_sm(et).ravel())
        return p

    def _fused_resample(self, pxPerDeg, containsPole=None, magnetic=False, arcsecPerPx=None):
        """
        The single-pass plan on the direction array (amt_pipe_launch_dirs: shell intersection, coordinates, mask, box and
        binning in one kernel that reads the (h + 1, w + 1, 3) directions once).  Not with arcsecPerPx (the box-first plan is
        built on the camera model) -> None: the array route.  A frame whose box comes near a pole of the grid's coordinates
        is decided from its corner quads by the two-pass plan inside run().
        """
        if self._frame is not None or not self._fusable() or arcsecPerPx:
            return None
        from ..pipeline import fused_class_pipeline
        hdr = self._wcsHeader
        pipe = fused_class_pipeline(hdr['IMAGEW'], hdr['IMAGEH'], self._img_array.dtype, magnetic)
        res = pipe.run(None, self.altitude, self.cameraPosGCRS, self.photoTime, img=self._img_array, fast=True,
                       min_elevation=self._lazy_elev, pxPerDeg=pxPerDeg, containsPole=containsPole, magnetic=magnetic, fuse=True,
                       params=self._params(), dirs=self._dirs_tensor(pipe.ctx))
        res['plan'] = pipe.last_plan
        pipe.forget_inputs()          # (the cached pipeline must not keep this mapping's direction array alive)
        return res

    def _dirs_tensor(self, ctx):
        if self._dirs_dev is None:
            self._dirs_dev = ctx.to_device(self._dirs)
        return self._dirs_dev

    def frame(self):
        if self._frame is None:
            ctx = Context.current()
            fd = FrameData(ctx, self._wcsHeader['IMAGEH'], self._wcsHeader['IMAGEW'])
            georef_into(fd, self._params(), geo=True, dirs=self._dirs_tensor(ctx))
            fd.set_image(self._img_array)
            self._frame = fd
            self._apply_lazy_elevation()
        return self._frame

    def _mlatmlt_tensors(self, center):
        fd = self.frame()
        if fd.mlat is None:
            georef_into(fd, self._params(), geo=False, mag=True, dirs=self._dirs_tensor(fd.ctx))
        return (fd.mlat_c, fd.mlt_c) if center else (fd.mlat, fd.mlt)

    @property
    def cameraToPixelCornerDirection(self):
        if self._dirs is None:
            from .._native import to_host
            self._dirs = to_host(self._dirs_dev)
        return self._dirs

    @property
    def cameraToPixelCenterDirection(self):
        return self._cached('dir_center', lambda: self._calcCenters(self.cameraToPixelCornerDirection.copy()))


def pixelDirection(fitsWcsHeader, corner=True):
    """
    Direction vector in ICRS for each pixel corner or center, given a WCS solution
    (reference astrometry.py:245-269; ICRS is treated as GCRS, the difference is far below a pixel).

    :param dictionary fitsWcsHeader: must also contain IMAGEW, IMAGEH in pixels
    :rtype: unit direction vector array of shape (IMAGEH+1, IMAGEW+1, 3) if corner==True,
            otherwise (IMAGEH, IMAGEW, 3)
    """
    w, h = fitsWcsHeader['IMAGEW'], fitsWcsHeader['IMAGEH']
    dirs = pix2world(fitsWcsHeader, w, h, corner=corner, ascartesian=True)
    assert tuple(dirs.shape) == (h + corner, w + corner, 3)
    return dirs
