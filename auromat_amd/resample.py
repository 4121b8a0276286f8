"""
Resampling of mappings onto a regular latitude/longitude grid (plate carree), relative to either
geodetic or MLat/MLT coordinates — mirror of the reference's auromat/resample.py for the
``method='mean'`` binning.

Host side: the grid definition (global alignment, pole / discontinuity handling, bin edges —
a few hundred scalars per frame, reference resample.py:159-299).  Device side: bin assignment with
the reference's ``searchsorted(..., 'right')`` edge semantics, accumulation and mean
(``amt_bin_frame`` / ``amt_hist2d_accumulate``), reference resample.py:301-351 and
util/histogram.py:57-282.  Pixel data never leaves the GPU; only the small output grid does.
"""
from __future__ import division, print_function

import copy
import ctypes as C
from functools import partial

import numpy as np
import numpy.ma as ma

from .coordinates.transform import rotation_matrix
from .coordinates.geodesic import wgs84A, wgs84B
from .mapping.mapping import (BaseMapping, MappingCollection, convertMappingToSM, convertSMMappingToGeo,
                              wrap_at_180)
from .coordinates.geodesic import angularDistanceOnParallel
from .util.histogram import make_axis
from ._native import Context, host9, ptr, to_host


def plateCarreeResolution(boundingBox, arcsecPerPx):
    """
    Approximates the latitude and longitude resolution of a plate carree projection from a
    spherical resolution for the area given by the bounding box (reference resample.py:36-61).
    The approximation is calculated for the bounding box center.  The longitude part needs
    ``geodesic.angularDistance`` between the box's mid-latitude end points, geographiclib's ``a12`` in the
    reference; :func:`auromat_amd.coordinates.geodesic.angularDistanceOnParallel` evaluates the same quantity
    from Karney's integral formulation to ~1e-14 relative, so ``round(pxPerDeg * 360 + 1)`` and with it the
    grid come out the same.

    The arithmetic runs in the library (``amt_plate_carree_resolution``, csrc/amt_grid.h: the same formulation in C++, host
    code, no GPU; microseconds where the NumPy bisection below takes a millisecond), so that the mapping classes, the
    sequence pipeline's box-first plan and the native sequence runner all derive a frame's px/deg from ONE implementation;
    :func:`plateCarreeResolution_py` is the Python restatement the tests pin it to.

    :type boundingBox: auromat_amd.mapping.mapping.BoundingBox
    :param arcsecPerPx: spherical resolution
    :rtype: tuple (latPxPerDeg, lonPxPerDeg)
    """
    from ._native import lib
    lat_ppd, lon_ppd = C.c_double(), C.c_double()
    rc = lib().amt_plate_carree_resolution(float(boundingBox.latSouth), float(boundingBox.lonWest), float(boundingBox.latNorth),
                                           float(boundingBox.lonEast), float(arcsecPerPx), C.byref(lat_ppd), C.byref(lon_ppd))
    if rc != 0 and rc != -5:
        return plateCarreeResolution_py(boundingBox, arcsecPerPx)      # (raises what the Python restatement raises)
    # (-5 = AMT_EDOMAIN: a box that goes all the way round — the reference's function returns (latPxPerDeg, 0) for it and fails
    # later, resample.py:226-227; the callers here check the longitude resolution before they lay out a grid)
    return lat_ppd.value, lon_ppd.value


def plateCarreeResolution_py(boundingBox, arcsecPerPx):
    """:func:`plateCarreeResolution` in Python (the restatement the library's C++ is checked against)."""
    degPerPx = arcsecPerPx / 3600.0
    latPxPerDeg = 1 / degPerPx
    latMiddle = (boundingBox.latNorth + boundingBox.latSouth) / 2
    lonEast = boundingBox.lonEast
    if boundingBox.lonWest > lonEast:
        lons = lonEast + 360 - boundingBox.lonWest
    else:
        lons = lonEast - boundingBox.lonWest
    # the shortest geodesic between the two end points spans min(lons, 360 - lons) of longitude
    lonMiddleDistance = angularDistanceOnParallel(latMiddle, min(lons, 360 - lons))
    px = lonMiddleDistance / degPerPx
    return latPxPerDeg, px / lons


def resampleMLatMLT(mapping, **kw):
    """Resamples a mapping such that MLat/MLT become regular grids (reference resample.py:63-71).

    See :func:`resample` for parameters.
    """
    global last_plan
    fused = getattr(mapping, '_fused_resample', None)
    if fused is not None and kw.get('method', 'mean') == 'mean' and \
            set(kw) <= {'pxPerDeg', 'containsPole', 'method', 'arcsecPerPx'}:
        # a camera mapping whose arrays nobody has asked for yet: the single-pass plan on the (MLat, SM longitude) grid
        # (arcsecPerPx: box-first, px/deg from the frame's own box in (MLat, SM longitude))
        arcsec = kw.get('arcsecPerPx')
        res = fused(None if arcsec else _px_per_deg(kw.get('pxPerDeg', 25)), kw.get('containsPole'), magnetic=True,
                    arcsecPerPx=arcsec)
        if res is not None:
            last_plan = res['plan']
            img = ma.masked_array(res['img'], mask=np.repeat(res['mask'][:, :, None], res['img'].shape[2], 2))
            elevation = ma.masked_invalid(res['mean'][:, :, -1], copy=False)
            # convertSMMappingToGeo (reference mapping.py:1549-1559) on the grid's arrays as they are: building the SM mapping
            # first would send the grid to the device and back through its properties, for the same numbers
            from .coordinates.transform import smToLatLon
            from .mapping.mapping import GenericMapping
            nc = res['lat'].size
            la, lo = smToLatLon(np.concatenate((res['lat'].ravel(), res['lat_c'].ravel())),
                                np.concatenate((res['lon'].ravel(), res['lon_c'].ravel())), mapping.photoTime)
            return GenericMapping(la[:nc].reshape(res['lat'].shape), lo[:nc].reshape(res['lon'].shape),
                                  la[nc:].reshape(res['lat_c'].shape), lo[nc:].reshape(res['lon_c'].shape), elevation,
                                  mapping.altitude, img, mapping.cameraPosGCRS, mapping.photoTime, mapping.identifier)
    sm = convertMappingToSM(mapping)
    smResampled = resample(sm, **kw)
    return convertSMMappingToGeo(smResampled)


# 'single-pass' / 'two-pass': the plan the last resample() of a camera mapping took (None: the array pipeline)
last_plan = None


def _px_per_deg(pxPerDeg):
    try:
        _, _ = pxPerDeg
    except TypeError:
        assert pxPerDeg is not None
        pxPerDeg = (pxPerDeg, pxPerDeg)
    return tuple(pxPerDeg)


def resample(mappingOrCollection, pxPerDeg=25, arcsecPerPx=None, containsPole=None, method='mean'):
    """
    Returns a new mapping (or collection) where the colors and elevation are resampled into a
    regular latitude/longitude grid (plate carree projection) with y=latitude and x=longitude
    (reference resample.py:73-157).

    With 'mean' binning, holes appear at low elevation angles when the resampling resolution is
    high, because binning does not interpolate empty bins; mask the mapping by elevation
    (e.g. 10deg) first.

    :param mappingOrCollection:
    :param None|number|tuple pxPerDeg: tuple (latPxPerDeg, lonPxPerDeg) or a number if both are the same
    :param None|number arcsecPerPx: spherical resolution, used to approximate pxPerDeg; has precedence
    :param None|bool containsPole: specify True|False to skip the pole check
    :param method: binning: 'mean'; interpolation: 'nearest' (value of the closest pixel centre in the lat/lon
                   plane), 'linear' and 'cubic' (scipy's griddata: the Delaunay triangulation of the pixel centres — Qhull's,
                   triangle for triangle, wherever it is unique —; 'linear': the barycentric sum in the grid centre's
                   triangle; 'cubic': scipy's gradient estimator in scipy's order with its stopping rule and the Clough-Tocher
                   element; both equal to the reference up to summation order, ~1e-13 of a channel's span (and up to the per-channel sweep count of
                   the relaxation, a yes / no decision at scipy's tolerance: see csrc/amt_nearest.hip k_cubic_gs); a cubic overshoots, and an
                   integer image wraps like numpy's cast), all masked outside the mapping's outline.
    :rtype: a subclass of BaseMapping or MappingCollection
    """
    _check_method(method)

    def doResample(mapping, pxPerDeg, arcsecPerPx, containsPole):
        global last_plan
        last_plan = None
        fused = getattr(mapping, '_fused_resample', None)
        if fused is not None and method == 'mean':
            # a camera mapping whose arrays nobody has asked for yet (getMapping(...).maskedByElevation(e), the user
            # guide's flow): georeferencing, mask, bounding box and binning in ONE kernel; with arcsecPerPx (the call form
            # of the reference's CLI and tests) a box pass of the same kernel comes first (box-first plan)
            res = fused(None if arcsecPerPx else _px_per_deg(pxPerDeg), containsPole, arcsecPerPx=arcsecPerPx)
            if res is not None:
                last_plan = res['plan']
                img = ma.masked_array(res['img'], mask=np.repeat(res['mask'][:, :, None], res['img'].shape[2], 2))
                elevation = ma.masked_invalid(res['mean'][:, :, -1], copy=False)
                return mapping.createResampled(res['lat'], res['lon'], res['lat_c'], res['lon_c'], elevation, img)
        if containsPole is None:
            containsPole = mapping.containsPole
        if arcsecPerPx:
            pxPerDeg = plateCarreeResolution(mapping.boundingBox, arcsecPerPx)
        else:
            try:
                _, _ = pxPerDeg
            except TypeError:
                assert pxPerDeg is not None
                pxPerDeg = (pxPerDeg, pxPerDeg)
        res = resample_frame(mapping.frame(), mapping.altitude, mapping.boundingBox, pxPerDeg,
                             mapping.containsDiscontinuity, containsPole, method=method,
                             outline=mapping.outline if (method != 'mean' or containsPole) else None)
        img = ma.masked_array(res['img'], mask=np.repeat(res['mask'][:, :, None], res['img'].shape[2], 2))
        elevation = ma.masked_invalid(res['mean'][:, :, -1], copy=False) if res['has_elev'] else None
        return mapping.createResampled(res['lat'], res['lon'], res['lat_c'], res['lon_c'], elevation, img)

    if isinstance(mappingOrCollection, BaseMapping):
        return doResample(mappingOrCollection, pxPerDeg, arcsecPerPx, containsPole)
    elif isinstance(mappingOrCollection, MappingCollection):
        mappings = [doResample(m, pxPerDeg, arcsecPerPx, containsPole) for m in mappingOrCollection.mappings]
        # (the reference forgets the identifier here and raises TypeError, resample.py:151)
        return MappingCollection(mappings, mappingOrCollection.identifier,
                                 mayOverlap=mappingOrCollection.mayOverlap)
    raise ValueError('First argument must be a mapping or a mapping collection, but is: {}'.
                     format(type(mappingOrCollection)))


def fixedGrid(pxPerDeg, latMin, latMax, lonMin, lonMax):
    """
    Aligns the given bounding box to a fixed plate carree grid as defined by `pxPerDeg`
    (reference resample.py:281-299).

    :param lonMin,lonMax: must NOT contain the discontinuity
    """
    latPxPerDeg, lonPxPerDeg = pxPerDeg
    latSpaceAll = _global_axis(-90, 90, int(round(latPxPerDeg * 180 + 1)))
    lonSpaceAll = _global_axis(-180, 180, int(round(lonPxPerDeg * 360 + 1)))
    # np.argmax(axis > v) / np.argmax(axis >= v) of the reference, as binary searches on the sorted axis
    # (an all-False comparison gives index 0 there, i.e. the searchsorted result n wraps to 0)
    def first_gt(axis, v):
        i = int(np.searchsorted(axis, v, side='right'))
        return i if i < len(axis) else 0

    def first_ge(axis, v):
        i = int(np.searchsorted(axis, v, side='left'))
        return i if i < len(axis) else 0

    latMinInGrid = latSpaceAll[first_gt(latSpaceAll, latMin) - 1]
    latMaxInGrid = latSpaceAll[first_ge(latSpaceAll, latMax)]
    lonMinInGrid = lonSpaceAll[first_gt(lonSpaceAll, lonMin) - 1]
    lonMaxInGrid = lonSpaceAll[first_ge(lonSpaceAll, lonMax)]
    nLat = int(round(latPxPerDeg * (latMaxInGrid - latMinInGrid) + 1))
    nLon = int(round(lonPxPerDeg * (lonMaxInGrid - lonMinInGrid) + 1))
    return nLat, nLon, latMinInGrid, latMaxInGrid, lonMinInGrid, lonMaxInGrid


_GLOBAL_AXES = {}


def _global_axis(lo, hi, n):
    key = (lo, hi, n)
    if key not in _GLOBAL_AXES:
        if len(_GLOBAL_AXES) > 64:
            _GLOBAL_AXES.clear()
        _GLOBAL_AXES[key] = np.linspace(lo, hi, n)
    return _GLOBAL_AXES[key]


_GRID_CACHE = {}


def cached_grid(pxPerDeg, latMin, latMax, lonMin, lonMax):
    """
    The :class:`_Grid` for a bounding box.  Grids only depend on the global nodes the box is rounded out
    to, and consecutive frames of a sequence mostly round to the same nodes, so the arrays (and the device
    axis descriptors attached to them) are kept and re-used.
    """
    nodes = fixedGrid(pxPerDeg, latMin, latMax, lonMin, lonMax)
    key = (tuple(pxPerDeg),) + nodes
    g = _GRID_CACHE.get(key)
    if g is None:
        if len(_GRID_CACHE) > 512:
            _GRID_CACHE.clear()
        g = _GRID_CACHE[key] = _Grid(pxPerDeg, latMin, latMax, lonMin, lonMax, _nodes=nodes)
    return g


class _Grid(object):
    """Output grid of one resampling (reference resample.py:220-241,330-334)."""

    def __init__(self, pxPerDeg, latMin, latMax, lonMin, lonMax, _nodes=None):
        latPxPerDeg, lonPxPerDeg = pxPerDeg
        assert latPxPerDeg > 0 and lonPxPerDeg > 0
        nLat, nLon, latLo, latHi, lonLo, lonHi = _nodes or fixedGrid(pxPerDeg, latMin, latMax, lonMin, lonMax)
        self._axes = {}
        assert nLat > 1, 'nlat={}, latMax={}, latMin={}, pxperdeg={}'.format(nLat, latHi, latLo, pxPerDeg)
        assert nLon > 1, 'nlon={}, lonMax={}, lonMin={}, pxperdeg={}'.format(nLon, lonHi, lonLo, pxPerDeg)
        latSpaceCenter, latStep = np.linspace(latHi, latLo, num=nLat, retstep=True)
        lonSpaceCenter, lonStep = np.linspace(lonLo, lonHi, num=nLon, retstep=True)
        # first and last centre are dropped so that no corner lies outside the determined range
        latSpace = latSpaceCenter[:-1] + latStep / 2
        lonSpace = lonSpaceCenter[:-1] + lonStep / 2
        self.latCenters = latSpaceCenter[1:-1]
        self.lonCenters = lonSpaceCenter[1:-1]
        self.latStep, self.lonStep = latStep, lonStep
        self.lat0, self.lon0 = (float(self.latCenters[0]), float(self.lonCenters[0])) if nLat > 2 and nLon > 2 \
            else (np.nan, np.nan)
        self._latSpace, self._lonSpace = latSpace, lonSpace
        self._corner_grid = self._center_grid = None
        self.nx, self.ny = len(self.lonCenters), len(self.latCenters)
        # histogram ranges; latitude edges ascend, the output is flipped afterwards
        self.xrange = [self.lonCenters[0] - lonStep / 2, self.lonCenters[-1] + lonStep / 2]
        self.yrange = [self.latCenters[-1] + latStep / 2, self.latCenters[0] - latStep / 2]
        self.xedges = np.linspace(self.xrange[0], self.xrange[1], self.nx + 1)
        self.yedges = np.linspace(self.yrange[0], self.yrange[1], self.ny + 1)

    def axes(self, ctx):
        """(amt_axis x, amt_axis y) of this grid on `ctx`'s device (built once, kept alive with the grid)."""
        key = ctx.device.index
        if key not in self._axes:
            self._axes[key] = (make_axis(ctx, self.xedges, uniform=True), make_axis(ctx, self.yedges, uniform=True))
        (xaxis, _), (yaxis, _) = self._axes[key]
        return xaxis, yaxis

    def device_corners(self, ctx):
        """(lat, lon) of the corner grid on the device, longitude-major ((nx+1, ny+1)), kept with the grid."""
        key = ('corners', ctx.device.index)
        if key not in self._axes:
            self._axes[key] = (ctx.to_device(np.ascontiguousarray(self.lat.T, dtype=np.float64)),
                               ctx.to_device(np.ascontiguousarray(self.lon.T, dtype=np.float64)))
        return self._axes[key]

    def device_centers(self, ctx):
        """(latCenters (ny), lonCenters (nx)) on the device, kept with the grid."""
        key = ('centers', ctx.device.index)
        if key not in self._axes:
            self._axes[key] = (ctx.to_device(np.ascontiguousarray(self.latCenters)),
                               ctx.to_device(np.ascontiguousarray(self.lonCenters)))
        return self._axes[key]

    # 2-D coordinate arrays of the output mapping (reference resample.py:239-241), built on first use
    def _corners(self):
        if self._corner_grid is None:
            self._corner_grid = np.dstack(np.meshgrid(self._latSpace, self._lonSpace)).T
        return self._corner_grid

    def _centers(self):
        if self._center_grid is None:
            self._center_grid = np.dstack(np.meshgrid(self.latCenters, self.lonCenters)).T
        return self._center_grid

    lat = property(lambda self: self._corners()[0])
    lon = property(lambda self: self._corners()[1])
    lat_c = property(lambda self: self._centers()[0])
    lon_c = property(lambda self: self._centers()[1])


def _rot_x(angle):
    return rotation_matrix(np.deg2rad(angle), [1, 0, 0])[:3, :3]


def _rotate_pole_dev(ctx, lat_deg, lon_deg, altitude, angle):
    """rotatePole (reference transform.py:301-322) on device tensors in degrees."""
    la, lo = lat_deg.reshape(-1), lon_deg.reshape(-1)
    if not la.is_contiguous():
        la = la.contiguous()
    if not lo.is_contiguous():
        lo = lo.contiguous()
    ola, olo = ctx.empty(la.shape), ctx.empty(lo.shape)
    ctx.call('amt_rotate_pole_deg', host9(_rot_x(angle)), ptr(la), ptr(lo), float(altitude), la.numel(), wgs84A, wgs84B,
             ptr(ola), ptr(olo))
    return ola.reshape(lat_deg.shape), olo.reshape(lon_deg.shape)


def _rotate_pole_host(lat_deg, lon_deg, altitude, angle):
    ctx = Context.current()
    la, lo = _rotate_pole_dev(ctx, ctx.to_device(np.ascontiguousarray(lat_deg)),
                              ctx.to_device(np.ascontiguousarray(lon_deg)), altitude, angle)
    return to_host(la), to_host(lo)


def wrap_at_180_t(t):
    """wrap_at_180 on a torch tensor (astropy Angle.wrap_at(180 deg): into [-180, 180))."""
    import torch
    a = t - torch.floor((t + 180.0) / 360.0) * 360.0
    a = torch.where(a >= 180.0, a - 360.0, a)
    return torch.where(a < -180.0, a + 360.0, a)


def _check_method(method):
    # (reference resample.py:353-360: NotImplementedError for 'median' and for anything it does not know)
    if method not in ('mean', 'nearest', 'linear', 'cubic'):
        raise NotImplementedError("method='%s' is not implemented (use 'mean', 'nearest', 'linear' or 'cubic')" % method)


CUBIC_TOLERANCE = 1e-6      # scipy.interpolate.CloughTocher2DInterpolator(tol=1e-6, maxiter=400): griddata's defaults
CUBIC_MAX_SWEEPS = 400

def cubic_exact(ctx, lat, lon, valid, values, height, width, grid, target_mask, method='cubic', vertices_out=None):
    """
    ``scipy.interpolate.griddata((lat, lon), values, grid centres, method='cubic' | 'linear')`` as the reference calls it
    (resample.py:315-326) on its own triangulation and in its own order: the Delaunay triangulation of the valid pixel centres
    (``amt_delaunay_create``, host: equal to Qhull's wherever that is unique); for 'cubic' scipy's Gauss-Seidel gradient
    estimator over its edges in the order of the points (``amt_cubic_gradients_csr``: tolerance 1e-6, at most 400 sweeps, every
    channel with its own stopping sweep) and the Clough-Tocher element in the triangle of every grid centre
    (``amt_delaunay_locate`` + ``amt_cubic_eval``); for 'linear' the barycentric sum over that triangle's vertices.
    `vertices_out`: a list that receives the (ny * nx, 3) int64 device tensor of each grid centre's triangle as flat pixel
    indices (-1: none).

    :param lat, lon: flat float64 device tensors (height * width) in the coordinates the grid is laid out in
    :param valid: flat bool device tensor: the pixel is a data point
    :param values: (height * width, channels) float64 device tensor
    :param target_mask: (ny, nx) uint8 device tensor, non-zero = the grid centre is not wanted (outside the outline), or None
    :return: ((ny * nx, channels) float64 device tensor, NaN outside the convex hull and where masked; sweeps per channel)
    """
    import os
    import time
    import torch
    from ._native import lib
    L = lib()
    debug = bool(os.environ.get('AMT_CUBIC_DEBUG'))
    marks = []

    def mark(what):
        if debug:
            torch.cuda.synchronize()
            marks.append((what, time.perf_counter()))
    mark('start')
    idx = torch.nonzero(valid.reshape(-1)).reshape(-1)               # row-major pixel order = the reference's point order
    n, nchan = int(idx.numel()), int(values.shape[1])
    out = torch.full((grid.ny * grid.nx, nchan), float('nan'), dtype=torch.float64, device=ctx.device)
    if n < 3:
        raise ValueError("method='%s' needs at least three valid pixels" % method)
    xy = torch.stack((lat.reshape(-1)[idx], lon.reshape(-1)[idx]), dim=1).contiguous()
    xy_host = np.ascontiguousarray(to_host(xy))
    handle = C.c_void_p()
    mark('points to the host')
    rc = L.amt_delaunay_create(xy_host.ctypes.data_as(C.c_void_p), n, C.byref(handle))
    if rc == -3:                                                     # AMT_ENOMEM
        raise MemoryError("method='%s': no memory for the triangulation of %d pixel centres" % (method, n))
    if rc != 0:
        raise ValueError("method='%s': the valid pixel centres cannot be triangulated (all collinear, or a coordinate that is "
                         "not finite)" % method)
    mark('triangulation')
    try:
        if method != 'linear':
            # the vertices' neighbour lists (scipy.spatial.Delaunay.vertex_neighbor_vertices, every list in increasing order), made
            # ON THE DEVICE from the build's own triangle slots (amt_delaunay_slots: no compaction, no list building, no 190 MB of
            # lists over the link — the host spent 0.2 s on those): every finite triangle gives the directed edges v1 -> v2,
            # v2 -> v0, v0 -> v1; an inner edge comes up once in each direction (from its two triangles), a hull edge once and
            # gets its reverse here; sorted by (source, target) they are the CSR the relaxation walks
            d_indptr, d_indices = device_vertex_lists(ctx, L, handle, n)
            rows = torch.div(idx, int(width), rounding_mode='floor')
            row_start = torch.zeros(int(height) + 1, dtype=torch.int64, device=ctx.device)
            row_start[1:] = torch.cumsum(torch.bincount(rows, minlength=int(height)), 0)
            mark('neighbour lists on the device')
        # the grid centres that are wanted, in row-major order (the walk from one to the next is a step or two)
        wanted = np.ones((grid.ny, grid.nx), dtype=bool) if target_mask is None else ~to_host(target_mask).astype(bool)
        sel = np.flatnonzero(wanted.ravel())
        m = int(sel.size)
        if m:
            iy, ix = np.divmod(sel, grid.nx)
            targets = np.ascontiguousarray(np.column_stack((np.asarray(grid.latCenters)[iy], np.asarray(grid.lonCenters)[ix])))
            vertices = np.empty((m, 3), dtype=np.int32)
            centroids = np.empty((m, 3, 2), dtype=np.float64)
            has_nb = np.empty((m, 3), dtype=np.uint8)
            rc = L.amt_delaunay_locate(handle, targets.ctypes.data_as(C.c_void_p), m, vertices.ctypes.data_as(C.c_void_p),
                                       centroids.ctypes.data_as(C.c_void_p), has_nb.ctypes.data_as(C.c_void_p))
            assert rc == 0
            d_t, d_v, d_c, d_h = (ctx.to_device(a, a.dtype) for a in (targets, vertices, centroids, has_nb))
            d_sel = ctx.to_device(sel.astype(np.int64), np.int64)
        mark('point location')
        if vertices_out is not None:
            tri_px = torch.full((grid.ny * grid.nx, 3), -1, dtype=torch.int64, device=ctx.device)
            if m:
                v64 = d_v.to(torch.int64)
                tri_px[d_sel] = torch.where(v64 >= 0, idx[v64.clamp(min=0)], v64)
            vertices_out.append(tri_px)
        sweeps_all = []
        if method == 'linear':
            if m:
                # scipy's LinearNDInterpolator: the barycentric sum over the triangle's vertices
                inside = d_v[:, 0] >= 0
                v64 = d_v.to(torch.int64).clamp(min=0)
                x, y = xy[:, 0][v64], xy[:, 1][v64]                       # (m, 3)
                px, py = d_t[:, 0], d_t[:, 1]
                det = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0])
                w1 = ((px - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (py - y[:, 0])) / det
                w2 = ((x[:, 1] - x[:, 0]) * (py - y[:, 0]) - (px - x[:, 0]) * (y[:, 1] - y[:, 0])) / det
                wts = torch.stack((1.0 - w1 - w2, w1, w2), dim=1)
                part = (wts[:, :, None] * values[idx][v64]).sum(dim=1)
                part[~inside] = float('nan')
                out[d_sel] = part
            return out, sweeps_all
        for c0 in range(0, nchan, 32):                      # (the relaxation kernel takes up to 63 channels, one per lane)
            vals = values[idx][:, c0:c0 + 32].contiguous()
            k = int(vals.shape[1])
            grad = ctx.empty((n, k, 2))
            sweeps = (C.c_int32 * k)()
            ctx.call('amt_cubic_gradients_csr', ptr(xy), n, ptr(d_indptr), ptr(d_indices), ptr(row_start), int(height), ptr(vals),
                     k, CUBIC_TOLERANCE, CUBIC_MAX_SWEEPS, ptr(grad), sweeps)
            sweeps_all.extend(int(v) for v in sweeps)
            mark('relaxation (%d sweeps)' % max(sweeps))
            if m:
                part = ctx.empty((m, k))
                ctx.call('amt_cubic_eval', m, ptr(d_t), ptr(d_v), ptr(d_c), ptr(d_h), ptr(xy), ptr(vals), ptr(grad), k, ptr(part))
                out[d_sel, c0:c0 + k] = part
        mark('element')
    finally:
        mark('values')
        # (giving back the triangulation's ~0.6 GB of host memory takes 40 ms: on a thread of its own — the call holds no lock of
        # the interpreter —, beside whatever the caller does next)
        import threading
        threading.Thread(target=L.amt_delaunay_destroy, args=(handle,), daemon=True).start()
        mark('triangulation handed back')
        if debug:
            print('cubic_exact: ' + ', '.join('%s %.3f s' % (b[0], b[1] - a[1]) for a, b in zip(marks, marks[1:])))
    return out, sweeps_all


def device_vertex_lists(ctx, L, handle, n):
    """(indptr (n + 1) int64, indices int32) on the device: the vertex neighbour lists of a triangulation made by
    amt_delaunay_create*, equal to amt_delaunay_vertex_neighbours' (tests/test_gpu_nearest.py)."""
    import torch
    pv, pd, ns = C.c_void_p(), C.c_void_p(), C.c_int64()
    rc = L.amt_delaunay_slots(handle, C.byref(pv), C.byref(pd), C.byref(ns))
    assert rc == 0 and ns.value > 0, rc
    slots = int(ns.value)
    v_host = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_int32)), shape=(slots, 3))
    dead_host = np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_uint8)), shape=(slots,))
    v = ctx.to_device(v_host, np.int32)
    dead = ctx.to_device(dead_host, np.uint8)
    tri = v[(dead == 0) & (v >= 0).all(dim=1)].to(torch.int64)            # finite triangles (a ghost carries -1)
    del v, dead
    src = torch.cat((tri[:, 1], tri[:, 2], tri[:, 0]))
    dst = torch.cat((tri[:, 2], tri[:, 0], tri[:, 1]))
    del tri
    key = torch.sort((src << 32) | dst).values
    rev = (dst << 32) | src
    del src, dst
    at = torch.searchsorted(key, rev).clamp(max=key.numel() - 1)
    hull = rev[key[at] != rev]                                             # reverses that no triangle supplies: hull edges
    del at, rev
    if hull.numel():
        key = torch.sort(torch.cat((key, hull))).values
    source = key >> 32
    indices = (key & 0xffffffff).to(torch.int32)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=ctx.device)
    indptr[1:] = torch.cumsum(torch.bincount(source, minlength=n), 0)
    return indptr, indices.contiguous()


def outside_outline_mask(ctx, grid, outline):
    """
    (ny, nx) uint8 device mask of the grid cells with a corner outside the outline polygon (reference
    resample.py:246-259: pointsInsidePolygon of the corner grid, a cell is masked if any of its 4 corners is outside).

    :param outline: (n,2) [lat,lon] polygon in the coordinates of the grid (rotated / shifted like the data)
    """
    import torch
    poly = ctx.to_device(np.ascontiguousarray(outline, dtype=np.float64))
    # points in longitude-major order: consecutive points share their y (= longitude), which lets the kernel drop
    # almost every polygon edge per block of points (it skips edges whose y-range misses the block's)
    lat, lon = grid.device_corners(ctx)
    inside = ctx.empty(tuple(lat.shape), torch.uint8)
    ctx.call('amt_points_in_polygon', ptr(lat), ptr(lon), lat.numel(), ptr(poly), int(poly.shape[0]), ptr(inside))
    out = inside.T == 0
    return (out[:-1, :-1] | out[1:, :-1] | out[:-1, 1:] | out[1:, 1:]).to(torch.uint8).contiguous()


def nearest_indices(ctx, lat_c, lon_c, elev, center_mask, height, width, min_elevation, grid, lon_wrap, target_mask):
    """(ny, nx) int64 device tensor: flat index of the pixel centre nearest to every grid centre, -1 = none / masked
    (``amt_nearest_frame``; reference resample.py:323-327, griddata(method='nearest'))."""
    import torch
    xaxis, yaxis = grid.axes(ctx)
    tlat, tlon = grid.device_centers(ctx)
    index = ctx.empty((grid.ny, grid.nx), torch.int64)
    min_el = float('-inf') if min_elevation is None else float(min_elevation)
    ctx.call('amt_nearest_frame', ptr(lat_c), ptr(lon_c), ptr(elev), ptr(center_mask), height, width, min_el,
             C.byref(xaxis), C.byref(yaxis), lon_wrap, ptr(tlat), ptr(tlon), ptr(target_mask), ptr(index))
    return index


def resample_frame(fd, altitude, boundingBox, pxPerDeg, containsDiscontinuity=False, containsPole=False,
                   min_elevation=None, keep_on_device=False, method='mean', outline=None, shard=None):
    """
    ``_resample`` + ``_resampleCenterData(method='mean')`` + the image finalisation of ``resample``
    (reference resample.py:119-136,159-279,301-351) on a device-resident frame.

    :param FrameData fd: centre lat/lon, elevation (optional), image and masks in HBM
    :param min_elevation: fuse ``maskedByElevation(min_elevation)`` into the binning pass (the frame's
                          own centre mask is applied in addition)
    :param method: 'mean' (binning) or 'nearest' (closest pixel centre; needs `outline`)
    :param outline: (n,2) [lat,lon] polygon of the mapping (``BaseMapping.outline``) for the interpolating methods:
                    grid cells with a corner outside it are masked (reference resample.py:246-259)
    :param shard: `fd` holds a band of rows of a frame whose other bands are on other ranks (see
                  :func:`auromat_amd.sequence.resample_frame_sharded`): shard.box combines the reduction of the rotated
                  corners, shard.acc sums the integer accumulators over the ranks before the means are taken
    :return: dict(lat, lon, lat_c, lon_c [grid coordinates, host], mean (ny,nx,C+1), img (ny,nx,C),
                  mask (ny,nx), count (ny,nx) ['mean' only], has_elev)
    """
    import torch
    _check_method(method)
    ctx = fd.ctx
    latMin, latMax = boundingBox.latSouth, boundingBox.latNorth
    lonMin, lonMax = boundingBox.lonWest, boundingBox.lonEast
    lat_c, lon_c = fd.lat_c, fd.lon_c
    lon_wrap = 0
    if containsPole:
        # rotate the pole out of the data by +90 deg about x (reference resample.py:176-201)
        if outline is not None:
            # as the reference: the extent of the rotated outline
            ola, olo = _rotate_pole_host(np.asarray(outline, dtype=np.float64)[:, 0],
                                         np.asarray(outline, dtype=np.float64)[:, 1], altitude, 90)
            latMin, latMax, lonMin, lonMax = ola.min(), ola.max(), olo.min(), olo.max()
        else:
            # (no outline at hand — the frame pipeline: all unmasked corners, the same unless the mask has islands)
            rla, rlo = _rotate_pole_dev(ctx, fd.lat, fd.lon, altitude, 90)
            corner, cmask = fd.corner_mask_tensor(), fd.center_mask_tensor()
            if min_elevation is not None and fd.elev is not None:
                # maskedByElevation(min_elevation) is fused into the binning pass below; the box is that of the corners
                # which survive it (mapping.py:845-864 + the lazy sanitisation, 1161-1213)
                cmask = (cmask.bool() | ~(fd.elev >= float(min_elevation))).to(torch.uint8)
                corner = corner.clone()
                ctx.call('amt_sanitize_masks', ptr(corner), ptr(cmask), None, fd.height, fd.width, 1)
            red = ctx.empty((8,))
            ctx.call('amt_bbox_corners', ptr(rla), ptr(rlo), ptr(corner), ptr(cmask), fd.height, fd.width, ptr(red))
            r = to_host(red)
            if shard is not None:
                r = shard.box(r)
            latMin, latMax, lonMin, lonMax = r[0], r[1], r[2], r[3]
        lat_c, lon_c = _rotate_pole_dev(ctx, fd.lat_c, fd.lon_c, altitude, 90)
    elif containsDiscontinuity:
        # rotate longitudes out of the 180 deg discontinuity (reference resample.py:203-218); the outline's
        # extremes are the box's west/east edge, and the wrap is monotonic on each side
        lonMin, lonMax = wrap_at_180(lonMin + 180), wrap_at_180(lonMax + 180)
        lon_wrap = 1

    grid = cached_grid(pxPerDeg, latMin, latMax, lonMin, lonMax)
    xaxis, yaxis = grid.axes(ctx)
    nch = fd.nchan
    if method in ('nearest', 'linear', 'cubic'):
        assert outline is not None, "method='%s' needs the outline of the mapping" % method
        assert shard is None, "rows of one frame over several ranks: method='mean' only"
        outline = np.array(outline, dtype=np.float64)
        if containsPole:
            outline[:, 0], outline[:, 1] = _rotate_pole_host(outline[:, 0], outline[:, 1], altitude, 90)
        elif containsDiscontinuity:
            outline[:, 1] = wrap_at_180(outline[:, 1] + 180)
        target_mask = outside_outline_mask(ctx, grid, outline)
        mean = ctx.empty((grid.ny, grid.nx, nch + 1))
        img = ctx.empty((grid.ny, grid.nx, max(nch, 1)), torch.uint8 if fd.img_dtype_code != 2 else torch.int16)
        mask = ctx.empty((grid.ny, grid.nx), torch.uint8)
        extra = {}
        index = None
        if method == 'nearest':
            index = nearest_indices(ctx, lat_c, lon_c, fd.elev, fd.center_mask, fd.height, fd.width, min_elevation, grid,
                                    lon_wrap, target_mask)
            ctx.call('amt_nearest_gather', ptr(index), grid.nx * grid.ny, ptr(fd.img), fd.img_dtype_code or 1, nch,
                     ptr(fd.elev), ptr(mean), ptr(img) if nch else None, ptr(mask))
        else:
            tri = ctx.empty((grid.ny, grid.nx, 3), torch.int64)
            min_el = float('-inf') if min_elevation is None else float(min_elevation)
            # scipy's griddata(method='linear' | 'cubic') on its own triangulation, in its own order (cubic_exact): image
            # channels and elevation as float64 channels of the valid pixels, then numpy's rounding and cast of the image
            assert fd.elev is not None, "method='%s' on a frame needs the elevation" % method
            la, lo = lat_c.reshape(-1), lon_c.reshape(-1)
            if lon_wrap:
                lo = wrap_at_180_t(lo + 180)
            valid = ~(torch.isnan(la) | torch.isnan(lo)) & (fd.elev.reshape(-1) >= min_el)
            if fd.center_mask is not None:
                valid &= fd.center_mask.reshape(-1) == 0
            chans = [fd.elev.reshape(-1, 1)]
            if nch:
                pix = fd.img.reshape(-1, nch)
                if fd.img_dtype_code == 2:
                    pix = pix.to(torch.int32) & 0xffff            # uint16 bits kept as int16
                chans.insert(0, pix.to(torch.float64))
            tri_out = []
            vals, sweeps = cubic_exact(ctx, la, lo, valid, torch.cat(chans, dim=1), fd.height, fd.width, grid, target_mask,
                                       method=method, vertices_out=tri_out)
            mean.copy_(vals.reshape(grid.ny, grid.nx, nch + 1))
            empty = torch.isnan(mean[..., 0])
            mask.copy_(empty.to(torch.uint8))
            if nch:
                # np.round + astype of the interpolated floats (reference resample.py:128-136); an overshoot wraps
                rounded = torch.round(torch.nan_to_num(mean[..., :nch], nan=0.0)).to(torch.int64)
                if fd.img_dtype_code == 2:
                    img.copy_((rounded & 0xffff).to(torch.int32).to(torch.int16))
                else:
                    img.copy_((rounded & 0xff).to(torch.uint8))
            tri.copy_(tri_out[0].reshape(grid.ny, grid.nx, 3))
            extra = dict(triangles=tri)
            if method == 'cubic':
                extra['sweeps'] = max(sweeps)
        out = dict(has_elev=fd.elev is not None, grid=grid, contains_pole=bool(containsPole),
                   contains_discontinuity=bool(containsDiscontinuity), altitude=altitude)
        if keep_on_device:
            out.update(mean=mean, img=img, mask=mask, **extra)
            if index is not None:
                out['index'] = index
            return out
        out.update(grid_coordinates(out))
        out.update(mean=to_host(mean), img=to_host(img, dtype=fd.img_dtype if nch else np.uint8),
                   mask=to_host(mask).astype(bool))
        if index is not None:
            out['index'] = to_host(index, dtype=np.int64)
        if extra:
            out.update(triangles=to_host(extra['triangles'], dtype=np.int64))
            if 'sweeps' in extra:
                out['sweeps'] = extra['sweeps']
        return out
    acc = ctx.zeros((nch + 2, grid.nx * grid.ny), torch.int64)
    min_el = float('-inf') if min_elevation is None else float(min_elevation)
    ctx.call('amt_bin_frame', ptr(lat_c), ptr(lon_c), ptr(fd.elev), ptr(fd.img), fd.img_dtype_code, nch,
             ptr(fd.center_mask), fd.height, fd.width, min_el, C.byref(xaxis), C.byref(yaxis), lon_wrap, ptr(acc))
    if shard is not None:
        shard.acc(acc)          # integer counts and sums: the order of the ranks does not matter
    mean = ctx.empty((grid.ny, grid.nx, nch + 1))
    img = ctx.empty((grid.ny, grid.nx, max(nch, 1)), torch.uint8 if fd.img_dtype_code != 2 else torch.int16)
    mask = ctx.empty((grid.ny, grid.nx), torch.uint8)
    count = ctx.empty((grid.ny, grid.nx))
    ctx.call('amt_bin_frame_finalize', ptr(acc), grid.nx, grid.ny, nch, fd.img_dtype_code or 1, ptr(mean),
             ptr(img) if nch else None, ptr(mask), ptr(count))

    out = dict(has_elev=fd.elev is not None, grid=grid, contains_pole=bool(containsPole),
               contains_discontinuity=bool(containsDiscontinuity), altitude=altitude)
    if keep_on_device:
        # sequence mode: the grid is described by `grid` (first centre + step); coordinate arrays on demand
        out.update(mean=mean, img=img, mask=mask, count=count)
        return out
    out.update(grid_coordinates(out))
    out.update(mean=to_host(mean), img=to_host(img, dtype=fd.img_dtype if nch else np.uint8),
               mask=to_host(mask).astype(bool), count=to_host(count))
    return out


def grid_coordinates(res):
    """Corner / centre coordinate arrays of a resample_frame result (reference resample.py:239-241,261-277)."""
    grid = res['grid']
    lat, lon, lat_gc, lon_gc = grid.lat, grid.lon, grid.lat_c, grid.lon_c
    if res['contains_pole']:
        lat, lon = _rotate_pole_host(lat, lon, res['altitude'], -90)            # reference resample.py:262-273
        lat_gc, lon_gc = _rotate_pole_host(lat_gc, lon_gc, res['altitude'], -90)
    elif res['contains_discontinuity']:
        lon = wrap_at_180(lon + 180)                                             # reference resample.py:274-277
        lon_gc = wrap_at_180(lon_gc + 180)
    return dict(lat=lat, lon=lon, lat_c=lat_gc, lon_c=lon_gc)


def _resample(latsCenter, lonsCenter, altitude, data, outlineLatLonFn, boundingBox, pxPerDeg,
              containsDiscontinuity=False, containsPole=False, method='mean'):
    """
    Array-level resampling with the reference's signature (resample.py:159-279): every channel of
    `data` (float, NaN = missing) is binned on its own.

    :param latsCenter, lonsCenter: (h,w), NaN = not mapped
    :param data: float data for each pixel center, (h,w,n) with n>0, or (h,w)
    :param outlineLatLonFn: callable returning (n,2) [lat,lon] points whose min/max bound the data
                            (only used in the pole / discontinuity branches)
    :param pxPerDeg: tuple (latPxPerDeg, lonPxPerDeg)
    :rtype: tuple (lat, lon, latCenter, lonCenter, data)
    """
    _check_method(method)
    ctx = Context.current()
    latMin, latMax = boundingBox.latSouth, boundingBox.latNorth
    lonMin, lonMax = boundingBox.lonWest, boundingBox.lonEast
    lat_c = ctx.to_device(np.asarray(latsCenter, dtype=np.float64))
    lon_c = ctx.to_device(np.asarray(lonsCenter, dtype=np.float64))
    lon_wrap = 0
    outline = None
    if containsPole:
        outline = np.array(outlineLatLonFn(), dtype=np.float64)
        outline[:, 0], outline[:, 1] = _rotate_pole_host(outline[:, 0], outline[:, 1], altitude, 90)
        latMin, latMax, lonMin, lonMax = np.min(outline[:, 0]), np.max(outline[:, 0]), np.min(outline[:, 1]), \
            np.max(outline[:, 1])
        lat_c, lon_c = _rotate_pole_dev(ctx, lat_c, lon_c, altitude, 90)
    elif containsDiscontinuity:
        outline = np.array(outlineLatLonFn(), dtype=np.float64)
        outline[:, 1] = wrap_at_180(outline[:, 1] + 180)
        lonMin, lonMax = np.min(outline[:, 1]), np.max(outline[:, 1])
        lon_wrap = 1
    grid = _Grid(pxPerDeg, latMin, latMax, lonMin, lonMax)
    scalar = np.ndim(data) == 2
    d = np.asarray(data, dtype=np.float64)
    if scalar:
        d = d[..., None]
    if method == 'mean':
        mean = _resampleCenterData(lat_c, lon_c, d, grid, lon_wrap)
    else:
        # nearest pixel centre for every grid centre, then everything outside the outline is masked
        # (reference resample.py:246-259,323-327; the reference rotates / shifts the outline in place, :176-218)
        import torch
        if outline is None:
            outline = np.array(outlineLatLonFn(), dtype=np.float64)
        target_mask = outside_outline_mask(ctx, grid, outline)
        h, w = d.shape[:2]
        flat = ctx.to_device(np.ascontiguousarray(d.reshape(h * w, d.shape[2])))
        if method == 'nearest':
            index = nearest_indices(ctx, lat_c.reshape(-1), lon_c.reshape(-1), None, None, h, w, None, grid, lon_wrap,
                                    target_mask)
            picked = flat[index.clamp(min=0).reshape(-1)]
            picked[index.reshape(-1) < 0] = float('nan')
        elif method == 'cubic':
            # scipy's griddata on its own triangulation, in its own order (cubic_exact); the data points are the pixels with
            # a latitude (reference resample.py:315-321)
            la, lo = lat_c.reshape(-1).contiguous(), lon_c.reshape(-1).contiguous()
            if lon_wrap:
                lo = wrap_at_180_t(lo + 180)
            picked, _ = cubic_exact(ctx, la, lo, ~(torch.isnan(la) | torch.isnan(lo)), flat, h, w, grid, target_mask)
        else:
            # scipy's griddata(method='linear') on its own triangulation (cubic_exact): Qhull's triangle of every grid centre,
            # the barycentric sum of the channels
            la, lo = lat_c.reshape(-1).contiguous(), lon_c.reshape(-1).contiguous()
            if lon_wrap:
                lo = wrap_at_180_t(lo + 180)
            picked, _ = cubic_exact(ctx, la, lo, ~(torch.isnan(la) | torch.isnan(lo)), flat, h, w, grid, target_mask, method='linear')
        mean = to_host(picked.reshape(grid.ny, grid.nx, d.shape[2]))
    lat, lon, lat_gc, lon_gc = grid.lat, grid.lon, grid.lat_c, grid.lon_c
    if containsPole:
        lat, lon = _rotate_pole_host(lat, lon, altitude, -90)
        lat_gc, lon_gc = _rotate_pole_host(lat_gc, lon_gc, altitude, -90)
    elif containsDiscontinuity:
        lon = wrap_at_180(lon + 180)
        lon_gc = wrap_at_180(lon_gc + 180)
    if scalar:
        mean = mean.reshape(mean.shape[0], mean.shape[1])
    return lat, lon, lat_gc, lon_gc, mean


def _resampleCenterData(lat_c, lon_c, centerData, grid, lon_wrap):
    """Binned mean of float channels (reference resample.py:301-368, method='mean')."""
    ctx = Context.current()
    n = lat_c.numel()
    nchan = centerData.shape[2]
    assert nchan <= 8, 'at most 8 channels per call'
    chans = [ctx.to_device(np.ascontiguousarray(centerData[:, :, k])) for k in range(nchan)]
    xaxis, xkeep = make_axis(ctx, grid.xedges, uniform=True)
    yaxis, ykeep = make_axis(ctx, grid.yedges, uniform=True)
    count = ctx.zeros((grid.nx * grid.ny,))
    sums = [ctx.zeros((grid.nx * grid.ny,)) for _ in range(nchan)]
    wptr = (C.c_void_p * nchan)(*[t.data_ptr() for t in chans])
    sptr = (C.c_void_p * nchan)(*[t.data_ptr() for t in sums])
    # pixels without coordinates are outliers of the histogram (NaN latitude, resample.py:315-321)
    ctx.call('amt_hist2d_accumulate', ptr(lon_c.reshape(-1)), ptr(lat_c.reshape(-1)), n, wptr, nchan,
             C.byref(xaxis), C.byref(yaxis), lon_wrap, ptr(count), sptr)
    mean = ctx.empty((grid.ny, grid.nx, nchan))
    ctx.call('amt_hist2d_finalize_mean', ptr(count), sptr, nchan, grid.nx, grid.ny, ptr(mean))
    return to_host(mean)


def ResampleProvider(provider, **kw):
    """
    Wrap the given mapping provider by resampling every returned mapping (reference resample.py:370-394).

    See :func:`resample` for parameters.
    """
    resampleFn = partial(resample, **kw)

    class ResamplingProvider(provider.__class__):
        def get(self, *a, **k):
            return resampleFn(super(ResamplingProvider, self).get(*a, **k))

        def getById(self, *a, **k):
            return resampleFn(super(ResamplingProvider, self).getById(*a, **k))

        def getSequence(self, *a, **k):
            return map(resampleFn, super(ResamplingProvider, self).getSequence(*a, **k))

    wrapped = copy.copy(provider)
    wrapped.__class__ = ResamplingProvider
    return wrapped
