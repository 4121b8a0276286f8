"""
Core classes for georeferenced images — the API surface of the reference's
auromat/mapping/mapping.py (BoundingBox, BaseMapping, GenericMapping, MappingCollection,
inflatedEarthIntersection, SM<->geo mapping conversion) on top of device-resident arrays.

A mapping owns a :class:`auromat_amd.frame.FrameData` (torch tensors in HBM).  The reference's
abstract properties (``lats``, ``lons``, ``latsCenter``, ``lonsCenter``, ``elevation``, ``img``,
``mLatMlt`` ...) are NumPy masked arrays fetched from the device on first access and cached;
the heavy consumers (``maskedByElevation``, ``boundingBox``, ``auromat_amd.resample.resample``)
work on the device tensors directly and never round-trip through the host.
"""
from __future__ import division

import copy
from abc import ABCMeta, abstractmethod, abstractproperty
from collections import namedtuple

import numpy as np
import numpy.ma as ma
from six import add_metaclass

from ..coordinates.geodesic import Location, wgs84A, wgs84B
from ..coordinates.intersection import ellipsoidLineIntersection, sphereLineIntersection
from ..coordinates.transform import (date2es, j2000ToLatLon, j2000ToMLatMLT, mat_geo_to_sm, mltToSmLon,
                                     smToLatLon)
from ..frame import FrameData
from .._native import host9, ptr, to_host

Size = namedtuple('Size', ['width', 'height'])
PixelScales = namedtuple('PixelScales', ['width', 'height', 'diagonal'])
PixelScale = namedtuple('PixelScale', ['mean', 'median', 'min', 'max'])
MappingProperties = namedtuple('MappingProperties',
                               'altitude cameraPosGCRS boundingBox photoTime '
                               'centroid cameraFootpoint identifier')

R_EARTH_KM = 6378.1366  # astropy.constants.R_earth (IAU 2012 nominal), used by earthModel='sphere'


def wrap_at_180(deg):
    """``astropy.coordinates.Angle(deg).wrap_at(180 deg).degree`` — into [-180, 180)."""
    a = np.array(deg, dtype=np.float64, copy=True)
    wraps = (a + 180.0) // 360.0
    a = a - wraps * 360.0
    a = np.where(a >= 180.0, a - 360.0, a)
    a = np.where(a < -180.0, a + 360.0, a)
    return a if a.ndim else float(a)


class BoundingBox(object):
    """
    Describes a geographical bounding box that can span across the discontinuity
    (reference mapping.py:44-287).
    """

    def __init__(self, latSouth, lonWest, latNorth, lonEast):
        assert -180 <= lonWest <= 180, 'Longitude: ' + str(lonWest)
        assert -180 <= lonEast <= 180, 'Longitude: ' + str(lonEast)
        assert -90 <= latSouth <= 90, 'Latitude: ' + str(latSouth)
        assert -90 <= latNorth <= 90, 'Latitude: ' + str(latNorth)
        self._latSouth, self._lonWest, self._latNorth, self._lonEast = latSouth, lonWest, latNorth, lonEast

    latSouth = property(lambda self: self._latSouth)
    lonWest = property(lambda self: self._lonWest)
    latNorth = property(lambda self: self._latNorth)
    lonEast = property(lambda self: self._lonEast)
    topLeft = property(lambda self: Location(self.latNorth, self.lonWest))
    bottomLeft = property(lambda self: Location(self.latSouth, self.lonWest))
    topRight = property(lambda self: Location(self.latNorth, self.lonEast))
    bottomRight = property(lambda self: Location(self.latSouth, self.lonEast))

    @property
    def _minSphericalRectangle(self):
        """
        (center, Size(width, height) in km) of the smallest "rectangle" of great-circle sides around the box — the
        parameters of a stereographic view of it (same values as the reference's property, mapping.py:118-170; sizes are
        meaningful for boxes narrower than 180 degrees of longitude).

        The geometry: of the two parallels that bound the box, the one nearer the equator is the longer and bulges OUT of
        the great circle through its end points, the one nearer the pole is shorter and its great circle bulges out beyond
        it.  So the rectangle is as wide as the equatorward edge's geodesic, and it reaches from that parallel at the box's
        central meridian to the midpoint of the poleward edge's geodesic.  A cap around a pole is a square of twice the
        geodesic from the pole to its bounding parallel.
        """
        from ..coordinates import geodesic
        if self.containsPole:
            pole_lat, rim_lat = (90, self.latSouth) if self.latNorth == 90 else (-90, self.latNorth)
            side = 2 * geodesic.distance(Location(pole_lat, 0), Location(rim_lat, 0))
            return Location(pole_lat, 0), Size(side / 1000, side / 1000)
        east = self.lonEast + (360 if self.lonWest > self.lonEast else 0)
        mid_lon = wrap_at_180((self.lonWest + east) / 2)
        span = {lat: geodesic.distance(Location(lat, self.lonWest), Location(lat, self.lonEast))
                for lat in (self.latSouth, self.latNorth)}
        # (equal spans — a box symmetric about the equator, or a point — count as "north of the equator")
        equatorward = self.latNorth if span[self.latNorth] > span[self.latSouth] else self.latSouth
        poleward = self.latSouth if equatorward == self.latNorth and self.latNorth != self.latSouth else self.latNorth
        on_parallel = Location(equatorward, mid_lon)
        on_geodesic = geodesic.intermediate(Location(poleward, self.lonWest), Location(poleward, self.lonEast), 0.5)
        height = geodesic.distance(on_parallel, on_geodesic)
        return geodesic.intermediate(on_parallel, on_geodesic, 0.5), Size(span[equatorward] / 1000, height / 1000)

    @property
    def center(self):
        """Center of the minimum spherical rectangle that fits the bounding box (:class:`Location`)."""
        return self._minSphericalRectangle[0]

    @property
    def size(self):
        """Width and height in km of the minimum spherical rectangle that fits the bounding box."""
        return self._minSphericalRectangle[1]

    @property
    def containsDiscontinuity(self):
        """Whether the bounding box contains the 180 degree discontinuity."""
        return self.lonWest > self.lonEast or self.containsPole

    @property
    def containsPole(self):
        """Whether the bounding box contains the north and/or south pole."""
        return self.lonWest == -180 and self.lonEast == 180 and (self.latNorth == 90 or self.latSouth == -90)

    @staticmethod
    def minimumBoundingBox(latLons):
        """The smallest box around (lat, lon) pairs."""
        return BoundingBox.mergedBoundingBoxes(BoundingBox(lat, lon, lat, lon) for lat, lon in latLons)

    @staticmethod
    def mergedBoundingBoxes(boundingBoxes):
        """The smallest box around the given boxes: the latitude extremes, and in longitude the complement of the widest
        arc of the circle that no box touches."""
        boxes = list(boundingBoxes)
        south = min(b.latSouth for b in boxes)
        north = max(b.latNorth for b in boxes)
        west, east = BoundingBox._minimumBoundingBoxLons([(b.lonWest, b.lonEast) for b in boxes])
        return BoundingBox(south, west, north, east)

    @staticmethod
    def _minimumBoundingBoxLons(lons):
        """
        (lonWest, lonEast) of the shortest arc that holds every [west, east] arc of `lons` (degrees).  The end points cut
        the circle into pieces; a piece inside one of the arcs is occupied, and the answer starts where the longest free
        piece ends and ends where it starts (the first such piece in ascending order of longitude when several tie, as
        the reference's argmax does, mapping.py:250-277).  An arc runs eastwards from its west end over less than half the
        circle (the reference unwraps each pair that way).
        """
        arcs = np.asarray(lons, dtype=np.float64).reshape(-1, 2)
        # eastwards of `west`, the nearer way round.  Both ends go through radians and back, as the reference's do: the round
        # trip can move an end by an ulp, and whether an arc then still "holds" the piece that starts at its own west end
        # decides ties the reference's way
        west, east = np.rad2deg(np.unwrap(np.deg2rad(arcs), axis=1)).T
        cuts = np.sort(arcs.ravel())
        ends = np.concatenate((cuts[1:], [cuts[0] + 360.0]))                    # piece k = [cuts[k], ends[k]]
        occupied = ((west[:, None] <= cuts[None, :]) & (east[:, None] >= ends[None, :])).any(axis=0)
        free = np.where(occupied, -np.inf, ends - cuts)
        k = int(np.argmax(free))
        return wrap_at_180(ends[k]), wrap_at_180(cuts[k])

    def __eq__(self, obj):
        return isinstance(obj, BoundingBox) and \
            self.latNorth == obj.latNorth and self.latSouth == obj.latSouth and \
            self.lonWest == obj.lonWest and self.lonEast == obj.lonEast

    def __ne__(self, obj):
        return not self == obj

    def __repr__(self):
        return 'BoundingBox(latSouth={0}, lonWest={1}, latNorth={2}, lonEast={3})'.format(
            self.latSouth, self.lonWest, self.latNorth, self.lonEast)


def bounding_box_from_reduction(red):
    """
    BaseMapping.boundingBox decision logic (reference mapping.py:711-741) on the 8 numbers of the
    device reduction [lat_min, lat_max, lon_min, lon_max, lon_min_positive, lon_max_nonpositive,
    n_valid, n_pole_quads].
    """
    lat_min, lat_max, lon_min, lon_max, lon_pos, lon_neg, n_valid, n_pole = [float(v) for v in red]
    if n_valid == 0:
        raise ValueError('The mapping has no valid pixels')
    if n_pole > 0:
        lonWest, lonEast = -180, 180
        if lat_max < 0:
            latSouth, latNorth = -90, lat_max
        else:
            latNorth, latSouth = 90, lat_min
    else:
        if lon_max - lon_min > 180:      # mappings are assumed smaller than 180 deg of longitude
            lonWest, lonEast = lon_pos, lon_neg
        else:
            lonWest, lonEast = lon_min, lon_max
        latNorth, latSouth = lat_max, lat_min
    return BoundingBox(latSouth, lonWest, latNorth, lonEast)


@add_metaclass(ABCMeta)
class BaseMapping(object):
    """
    Base class for all mapping objects: a georeferenced image for a given altitude
    (reference mapping.py:293-929).  The guarantees of the reference hold:

    - lats[y,x] masked <=> lons[y,x] masked; latsCenter[y,x] masked <=> lonsCenter[y,x] masked.
    - lats[y,x] not masked => at least one adjacent centre not masked.
    - latsCenter[y,x] not masked => its four corners not masked.
    - img[y,x] / elevation[y,x] masked <=> latsCenter[y,x] masked.
    """

    def __init__(self, altitude, cameraPosGCRS, photoTime, identifier, metadata=None):
        assert altitude >= 0
        cameraPosGCRS = np.asarray(cameraPosGCRS)
        assert cameraPosGCRS.shape == (3,)
        self._altitude = altitude
        self._cameraPosGCRS = cameraPosGCRS
        self._photoTime = photoTime
        self._identifier = identifier
        self._metadata = metadata
        self._host = {}            # cache of host (NumPy masked) views
        self._boundingBox = None

    # -- device state ------------------------------------------------------------------------
    @abstractmethod
    def frame(self):
        """The device-resident :class:`FrameData` of this mapping (computed / uploaded on first use)."""

    def _cached(self, key, make):
        if key not in self._host:
            self._host[key] = make()
        return self._host[key]

    def _corner_array(self, name):
        fd = self.frame()
        return self._cached(name, lambda: ma.masked_array(fd.host(name), mask=fd.host_mask('corner')))

    def _center_array(self, name):
        fd = self.frame()
        return self._cached(name, lambda: ma.masked_array(fd.host(name), mask=fd.host_mask('center')))

    # -- simple attributes -------------------------------------------------------------------
    altitude = property(lambda self: self._altitude, doc='Mapping altitude in km.')
    cameraPosGCRS = property(lambda self: self._cameraPosGCRS)
    photoTime = property(lambda self: self._photoTime)
    identifier = property(lambda self: self._identifier)

    @property
    def metadata(self):
        return {} if self._metadata is None else self._metadata

    @property
    def cameraFootpoint(self):
        """The camera footpoint in geodetic coordinates (:class:`Location`)."""
        lat, lon = self._cached('footpoint', lambda: j2000ToLatLon([self.cameraPosGCRS], self.photoTime))
        return Location(lat[0], lon[0])

    @property
    def properties(self):
        return MappingProperties(identifier=self.identifier, altitude=self.altitude,
                                 cameraPosGCRS=self.cameraPosGCRS, boundingBox=self.boundingBox,
                                 photoTime=self.photoTime, centroid=self.centroid,
                                 cameraFootpoint=self.cameraFootpoint)

    # -- coordinate arrays (NumPy masked views) ------------------------------------------------
    @property
    def lats(self):
        """Masked array (h+1, w+1): latitude of every pixel corner, degrees."""
        return self._corner_array('lat')

    @property
    def lons(self):
        """Masked array (h+1, w+1): longitude of every pixel corner, degrees."""
        return self._corner_array('lon')

    @property
    def latsCenter(self):
        """Masked array (h, w): latitude of every pixel centre."""
        return self._center_array('lat_c')

    @property
    def lonsCenter(self):
        """Masked array (h, w): longitude of every pixel centre."""
        return self._center_array('lon_c')

    @property
    def elevation(self):
        """Masked array (h, w): elevation in degrees for each pixel centre (or None)."""
        fd = self.frame()
        if fd.elev is None:
            return None

        def make():
            elev = fd.host('elev')
            return ma.masked_array(elev, mask=fd.host_mask('center') | np.isnan(elev))
        return self._cached('elev', make)

    @property
    def img_unmasked(self):
        """Like img but as a normal numpy array."""
        return self._cached('img_unmasked', lambda: self.frame().host_image())

    @property
    def img(self):
        """Masked array of shape (h,w,n) and type uint8/uint16."""
        def make():
            data = self.img_unmasked
            mask = self.frame().host_mask('center')
            return ma.masked_array(data, mask=np.repeat(mask[:, :, None], data.shape[2], 2))
        return self._cached('img', make)

    @property
    def rgb_unmasked(self):
        """(h,w,3) uint8 RGB representation of img (reference DefaultRGBMixin, mapping.py:980-1007)."""
        src = self.img_unmasked
        if src.dtype == np.uint8:
            img = src
        elif src.dtype == np.uint16:
            img = (src * (255 / 65535)).astype(np.uint8)
        else:
            raise NotImplementedError
        if img.shape[2] == 3:
            return img
        elif img.shape[2] == 1:
            return np.repeat(img, 3, 2)
        raise NotImplementedError('Unknown img format')

    @property
    def rgb(self):
        rgbm = self.rgb_unmasked
        mask = self.frame().host_mask('center')
        return ma.masked_array(rgbm, mask=np.repeat(mask[:, :, None], rgbm.shape[2], 2))

    # -- geomagnetic coordinates -----------------------------------------------------------------
    def _mlatmlt_tensors(self, center):
        """
        Generic path (reference mapping.py:540-550): geodetic lat/lon at the mapping altitude ->
        ECEF -> SM -> MLat/MLT, one kernel (amt_latlon_to_mlat_mlt).
        """
        fd = self.frame()
        names = ('mlat_c', 'mlt_c', 'lat_c', 'lon_c') if center else ('mlat', 'mlt', 'lat', 'lon')
        if getattr(fd, names[0]) is None:
            lat, lon = getattr(fd, names[2]), getattr(fd, names[3])
            mlat, mlt = fd.ctx.empty(lat.shape), fd.ctx.empty(lat.shape)
            fd.ctx.call('amt_latlon_to_mlat_mlt', host9(mat_geo_to_sm(date2es(self.photoTime))), ptr(lat), ptr(lon),
                        float(self.altitude), lat.numel(), wgs84A, wgs84B, ptr(mlat), ptr(mlt))
            setattr(fd, names[0], mlat)
            setattr(fd, names[1], mlt)
        return getattr(fd, names[0]), getattr(fd, names[1])

    @property
    def mLatMlt(self):
        """Tuple (mlat, mlt) of masked arrays (h+1, w+1) for the pixel corners: degrees / hours."""
        def make():
            mlat, mlt = self._mlatmlt_tensors(False)
            mask = self.frame().host_mask('corner')
            return ma.masked_array(to_host(mlat), mask), ma.masked_array(to_host(mlt), mask)
        return self._cached('mlatmlt', make)

    @property
    def mLatMltCenter(self):
        """Tuple (mlat, mlt) of masked arrays (h, w) for the pixel centres."""
        def make():
            mlat, mlt = self._mlatmlt_tensors(True)
            mask = self.frame().host_mask('center')
            return ma.masked_array(to_host(mlat), mask), ma.masked_array(to_host(mlt), mask)
        return self._cached('mlatmlt_c', make)

    # -- plate carree ------------------------------------------------------------------------------
    @property
    def isPlateCarree(self):
        return isPlateCarree(self.lats, self.lons)

    def checkPlateCarree(self):
        return checkPlateCarree(self.lats, self.lons)

    # -- bounding box (device reduction) -----------------------------------------------------------
    def _bbox_reduction(self):
        fd = self.frame()
        if fd.bbox is None:
            bbox = fd.ctx.empty((8,))
            fd.ctx.call('amt_bbox_corners', ptr(fd.lat), ptr(fd.lon), ptr(fd.corner_mask_tensor()),
                        ptr(fd.center_mask_tensor()), fd.height, fd.width, ptr(bbox))
            fd.bbox = bbox
        return fd.host_bbox()

    @property
    def boundingBox(self):
        """
        Smallest lat/lon box around the unmasked corners (reference mapping.py:693-743).  The
        reference traces the mask outline on the host and asks geographiclib whether it encloses a
        pole; here one device pass reduces the corner extremes and counts pixels whose corner quad
        winds around a pole; mappings whose longitudes go all around but where no unmasked pixel sees a pole are
        decided by the reference's own rule on the traced outline (see _pole_possible).  In case containsPole is
        True the box spans the full longitude range.
        """
        if self._boundingBox is None:
            red = np.array(self._bbox_reduction(), dtype=np.float64)
            if red[6] > 0:
                # the extremes come from the outline, as in the reference (mapping.py:699-705): identical to the
                # device's reduction over all unmasked corners unless the mask has islands besides its biggest
                # component, which the reference's outline (the biggest contour) leaves out
                outl = self.outline
                lo = outl[:, 1]
                red[0], red[1], red[2], red[3] = outl[:, 0].min(), outl[:, 0].max(), lo.min(), lo.max()
                red[4] = lo[lo > 0].min() if np.any(lo > 0) else np.inf
                red[5] = lo[lo <= 0].max() if np.any(lo <= 0) else -np.inf
            # pole: the reference's rule (does the sampled convex hull of the outline contain or cross a pole,
            # mapping.py:705-721), asked only when the outline's longitudes go all around.  The device's count of
            # pixel quads winding around a pole agrees with it except when the pole sits in a hole / masked part
            # (no pixel sees it) or in an island that is not part of the outline (a pixel sees it, the outline
            # does not) — tools/fuzz_mapping.py found both.
            red[7] = 1 if (red[6] > 0 and self._pole_possible(red) and self._hull_contains_pole()) else 0
            self._boundingBox = bounding_box_from_reduction(red)
        return self._boundingBox

    @staticmethod
    def _pole_possible(red):
        """
        The device counts pixels whose corner quad winds around a pole.  A pole can also lie in a hole or in a masked
        part of the footprint, where no unmasked pixel sees it, while the reference's rule — does the (sampled) convex
        hull of the outline contain or cross a pole, mapping.py:705-721 — still says yes.  That needs longitudes all
        around: this is the cheap test for it (corner longitudes spanning more than 180 deg and no gap of more than
        180 deg around longitude 0, which is what a mapping across the date line has).
        """
        lon_min, lon_max, lon_pos, lon_neg = red[2], red[3], red[4], red[5]
        if lon_max - lon_min <= 180:
            return False
        return not (lon_pos - lon_neg > 180)

    def _hull_contains_pole(self):
        # reference mapping.py:705-715: at most 50 points of the convex hull of the outline
        from ..coordinates.geodesic import containsOrCrossesPole
        hull = self.outlineConvexHull
        pointCount = len(hull)
        if pointCount < 3:
            return False
        indices = np.round(np.linspace(0, pointCount - 1, min(pointCount, 50))).astype(int)
        try:
            return bool(containsOrCrossesPole(hull[indices]))
        except AssertionError:
            # the course deltas do not add up to a multiple of 180 deg: a hull vertex sits (numerically) on a pole,
            # where an azimuth is not defined (the reference's own test notes this case as broken,
            # geodesic_test.py:31-38) — the outline crosses the pole
            return True

    @property
    def containsDiscontinuity(self):
        return self.boundingBox.containsDiscontinuity

    @property
    def containsPole(self):
        return self.boundingBox.containsPole

    @property
    def outline(self):
        """
        The complete outline of this mapping as an (n,2) [lat,lon] array: the traced contour of the unmasked
        corners (reference mapping.py:655-691, utils.py:97-151), clockwise in image coordinates.
        Note that the outline can be concave.
        """
        return self._latlon_at(self._outline_pixels)

    @property
    def outlineConvexHull(self):
        """The convex hull (in pixel space) of the regular outline, as [lat,lon] (reference mapping.py:664-690)."""
        from ..utils import convexHull
        return self._cached('outline_hull', lambda: self._latlon_at(convexHull(self._outline_pixels), cache=None))

    @property
    def _outline_pixels(self):
        def make():
            from ..utils import outline_of_mask_tensor
            fd = self.frame()
            return outline_of_mask_tensor(fd.ctx, fd.corner_mask_tensor(), fd.height + 1, fd.width + 1)
        return self._cached('outline_px', make)

    def _latlon_at(self, xy, cache='outline'):
        def make():
            import torch
            fd = self.frame()
            idx = torch.from_numpy(np.ascontiguousarray(xy[:, 1] * (fd.width + 1) + xy[:, 0])).to(fd.lat.device)
            return np.transpose([to_host(fd.lat.reshape(-1)[idx]), to_host(fd.lon.reshape(-1)[idx])])
        return self._cached(cache, make) if cache else make()

    @property
    def centroid(self):
        """The centroid of the mapping based on the plate-carree projection (reference mapping.py:758-784).

        :rtype: auromat.coordinates.geodesic.Location
        """
        from ..utils import polygonCentroid
        if self.containsPole:
            # TODO rotate away from pole, see resample module
            raise NotImplementedError
        outline = self.outline
        if self.containsDiscontinuity:
            shifted = np.transpose([outline[:, 0], wrap_at_180(outline[:, 1] + 180)])
            lat, lon = polygonCentroid(shifted)
            return Location(lat, float(wrap_at_180(lon + 180)))
        lat, lon = polygonCentroid(outline)
        return Location(lat, lon)

    # -- masking -----------------------------------------------------------------------------------
    def maskedByElevation(self, minElevation=10):
        """
        Return a new mapping with data masked below the given minimum elevation
        (reference mapping.py:845-864 + the lazy sanitisation of mapping.py:1063-1125,1161-1213).
        The new mapping shares the device arrays and only gets new masks.

        :param minElevation: 0 to 90, in degrees
        :raises ValueError: if every pixel would be masked
        """
        import torch
        fd = self.frame()
        assert fd.elev is not None
        ctx = fd.ctx
        center = ctx.empty((fd.height, fd.width), torch.uint8)
        corner = ctx.empty((fd.height + 1, fd.width + 1), torch.uint8)
        n_valid = ctx.zeros((1,), torch.int64)
        if fd.center_mask is None and fd.corner_mask is None:
            ctx.call('amt_mask_by_elevation', ptr(fd.elev), ptr(fd.lat), fd.height, fd.width, float(minElevation),
                     ptr(center), ptr(corner), ptr(n_valid))
            count = int(n_valid.item())
        else:
            # keep what is already masked: centre |= previous centre mask, then the corner rule
            ctx.call('amt_mask_by_elevation', ptr(fd.elev), None, fd.height, fd.width, float(minElevation),
                     ptr(center), None, None)
            corner.copy_(fd.corner_mask_tensor())
            ctx.call('amt_sanitize_masks', ptr(corner), ptr(center), ptr(fd.center_mask_tensor()), fd.height,
                     fd.width, 1)
            count = int((center == 0).sum().item())
        if count == 0:
            raise ValueError('minElevation=' + str(minElevation) + ' would mask all pixels!')
        return self.createMasked(center, _corner_mask=corner)

    def maskedByPolygon(self, polygon):
        """
        Returns a copy of this mapping where the image is masked using the given polygon.  Only those pixels are
        retained where all of its corners are inside the polygon (reference mapping.py:866-917; the point-in-polygon
        test of all corners runs on the device, ``amt_points_in_polygon``).

        .. warning:: If the mapping or the polygon contains the discontinuity and/or poles then this method tries to
                 handle it in a best-effort approach, as the reference does.

        :param array-like polygon: ordered points of an unclosed polygon in [lat,lon] order
        :rtype: BaseMapping
        """
        import torch
        from ..coordinates.geodesic import containsOrCrossesPole
        from ..resample import _rotate_pole_dev, _rotate_pole_host
        polygon = np.array(polygon, dtype=np.float64)
        fd = self.frame()
        ctx = fd.ctx
        lat, lon = fd.lat, fd.lon
        polyBoundingBox = BoundingBox.minimumBoundingBox(polygon)
        if self.containsDiscontinuity or polyBoundingBox.containsDiscontinuity:
            polygon[:, 1] = wrap_at_180(polygon[:, 1] + 180)
            lon = torch.remainder(lon + 360.0, 360.0) - 180.0           # wrap_at(lon + 180, 180)
        elif self.containsPole or containsOrCrossesPole(polygon):
            polygon[:, 0], polygon[:, 1] = _rotate_pole_host(polygon[:, 0], polygon[:, 1], self.altitude, 90)
            lat, lon = _rotate_pole_dev(ctx, lat, lon, self.altitude, 90)
        inside = ctx.empty(tuple(lat.shape), torch.uint8)
        lat, lon, poly = lat.contiguous(), lon.contiguous(), ctx.to_device(polygon)     # named: they outlive the call
        ctx.call('amt_points_in_polygon', ptr(lat), ptr(lon), lat.numel(), ptr(poly), len(polygon), ptr(inside))
        mask = (inside == 0) | (fd.corner_mask_tensor() != 0)
        if bool(mask.all().item()):
            raise ValueError('The given mask would mask all pixels!')
        # we mask every pixel which misses at least one of its four corner coordinates
        centerMask = mask[:-1, :-1] | mask[1:, :-1] | mask[:-1, 1:] | mask[1:, 1:]
        return self.createMasked(centerMask.to(torch.uint8).contiguous())

    @property
    def arcSecPerPx(self):
        """
        Min, max, median, and mean angular sizes of pixels/polygons determined for the width, height, and
        diagonal of 1000 polygons (reference mapping.py:786-843; the geodesic arcs of the sampled polygons are host
        arithmetic, only their 3 x 1000 corner coordinates leave the device).

        :rtype: PixelScales
        """
        def make():
            import torch
            from ..coordinates import geodesic
            fd = self.frame()
            bad = fd.corner_mask_tensor() != 0
            has_nans = bad[:-1, :-1] | bad[:-1, 1:] | bad[1:, 1:] | bad[1:, :-1]
            polys = torch.nonzero(~has_nans.reshape(-1)).reshape(-1)
            polyCount = int(polys.numel())
            sampleCount = min(polyCount, 1000)
            pick = polys[torch.from_numpy(np.round(np.linspace(0, polyCount - 1, sampleCount)).astype(np.int64))
                         .to(polys.device)]
            i, j = pick // fd.width, pick % fd.width
            W1 = fd.width + 1
            corners = torch.stack((i * W1 + j, i * W1 + j + 1, (i + 1) * W1 + j + 1))        # verts 0, 1, 2
            la = to_host(fd.lat.reshape(-1)[corners.reshape(-1)]).reshape(3, -1)
            lo = to_host(fd.lon.reshape(-1)[corners.reshape(-1)]).reshape(3, -1)
            scales = []
            for a, b in ((0, 1), (1, 2), (0, 2)):               # width, height, diagonal
                deg = [geodesic.angularDistance(Location(la[a, k], lo[a, k]), Location(la[b, k], lo[b, k]))
                       for k in range(sampleCount)]
                scales.append(PixelScale(np.mean(deg) * 3600, np.median(deg) * 3600, min(deg) * 3600,
                                         max(deg) * 3600))
            return PixelScales(width=scales[0], height=scales[1], diagonal=scales[2])
        return self._cached('pixel_scales', make)

    def createMasked(self, centerMask, _corner_mask=None):
        """
        Return a copy of this mapping with the given centre mask applied (True/1 = masked) to img,
        latsCenter, lonsCenter and elevation; corners without any unmasked neighbour centre get
        masked as well (reference mapping.py:616-653,1161-1213).
        """
        import torch
        fd = self.frame()
        new = fd.shallow_copy()
        if isinstance(centerMask, torch.Tensor):
            center = centerMask.to(torch.uint8)
        else:
            center = fd.ctx.to_device(np.asarray(centerMask, dtype=np.uint8), np.uint8)
        assert tuple(center.shape) == (fd.height, fd.width)
        if _corner_mask is None:
            _corner_mask = fd.corner_mask_tensor().clone()
            fd.ctx.call('amt_sanitize_masks', ptr(_corner_mask), ptr(center), None, fd.height, fd.width, 1)
        new.center_mask, new.corner_mask, new.bbox = center, _corner_mask, None
        m = copy.copy(self)
        m._frame = new
        m.setDirty()
        return m

    def setDirty(self):
        """Forget every cached host view and derived attribute."""
        self._host = {}
        self._boundingBox = None

    # -- testing aid ---------------------------------------------------------------------------------
    def checkGuarantees(self):
        """Checks the mask guarantees listed in the class docstring (reference mapping.py:362-428)."""
        lats, lons = self.lats, self.lons
        latsCenter, lonsCenter = self.latsCenter, self.lonsCenter
        mlat, mlt = self.mLatMlt
        mlatCenter, mltCenter = self.mLatMltCenter
        img, elevation = self.img, self.elevation
        assert not np.any(np.isnan(lats.compressed()))
        assert not np.any(np.isnan(latsCenter.compressed()))
        assert not np.any(np.isnan(mlat.compressed()))
        assert not np.any(np.isnan(elevation.compressed()))
        gm = ma.getmaskarray
        assert np.array_equal(gm(lats), gm(lons))
        assert np.array_equal(gm(latsCenter), gm(lonsCenter))
        ok = np.zeros((latsCenter.shape[0] + 2, latsCenter.shape[1] + 2), bool)
        ok[1:-1, 1:-1] = ~gm(latsCenter)
        assert np.all(np.logical_or.reduce((gm(lats), ok[1:, 1:], ok[1:, :-1], ok[:-1, :-1], ok[:-1, 1:])))
        cok = ~gm(lats)
        assert np.all(np.logical_or(gm(latsCenter), np.logical_and.reduce(
            (cok[:-1, :-1], cok[1:, :-1], cok[1:, 1:], cok[:-1, 1:]))))
        center_ok = ~gm(latsCenter)
        for d in range(img.shape[2]):
            assert np.all(np.logical_xor(gm(img)[:, :, d], center_ok))
        assert np.all(np.logical_xor(gm(elevation), center_ok))
        assert np.all(np.logical_xor(gm(mlatCenter), center_ok))
        assert np.all(np.logical_xor(gm(mltCenter), center_ok))
        assert np.all(np.logical_xor(gm(mlat), cok))
        assert np.all(np.logical_xor(gm(mlt), cok))

    @abstractmethod
    def createResampled(self, lats, lons, latsCenter, lonsCenter, elevation, img):
        """Returns a new mapping object of the appropriate class for resampled data."""


def checkPlateCarree(lats, lons):
    """
    Checks whether the given 2D coordinate arrays describe a plate carree projection: latitudes
    (longitudes) evenly spaced and monotonically decreasing (increasing) (reference mapping.py:931-965).

    :raise ValueError: when the projection is not plate carree
    """
    if ma.isMaskedArray(lats):
        lats, lons = lats.data, lons.data
    if np.any(np.isnan(lats)):
        raise ValueError('coordinates contains NaNs')
    lons = np.unwrap(np.deg2rad(lons))
    if lons[0, -1] - lons[0, 0] <= 0:
        raise ValueError('longitudes are not monotonically increasing')
    if lats[0, 0] - lats[-1, 0] <= 0:
        raise ValueError('latitudes are not monotonically decreasing')
    eps = 1e-4
    deltaLon = lons[0, 1:] - lons[0, :-1]
    if not np.max(deltaLon) - np.min(deltaLon) < eps:
        raise ValueError('longitudes are not evenly spaced; max delta: {}'.format(np.max(deltaLon) - np.min(deltaLon)))
    deltaLat = lats[:-1, 0] - lats[1:, 0]
    if not np.max(deltaLat) - np.min(deltaLat) < eps:
        raise ValueError('latitudes are not evenly spaced; max delta: {}'.format(np.max(deltaLat) - np.min(deltaLat)))


def isPlateCarree(lats, lons):
    try:
        checkPlateCarree(lats, lons)
    except Exception:
        return False
    return True


class GenericMapping(BaseMapping):
    """
    A mapping consisting of precalculated latitudes/longitudes/elevation values
    (reference mapping.py:1233-1313, including the lazy sanitisation of its ``sanitize_data``
    decorator, mapping.py:1063-1231, which runs as mask stencils on the device).
    """

    def __init__(self, lats, lons, latsCenter, lonsCenter, elev, alti, img, cameraPosGCRS, photoTime,
                 identifier, metadata=None):
        """
        :param ndarray lats, lons: (h+1,w+1) in degrees (NaN or masked = missing)
        :param ndarray latsCenter, lonsCenter: (h,w) in degrees
        :param ndarray elev: (h,w) in degrees; can also be None
        :param number alti: the altitude in km onto which the image was mapped (e.g. 110)
        :param ndarray img: uint8 or uint16 array of shape (h,w) for grayscale or (h,w,3) for RGB
        :param array-like cameraPosGCRS: [x,y,z] in km
        :param datetime.datetime photoTime:
        """
        h, w = img.shape[0], img.shape[1]
        assert lats.shape == lons.shape == (h + 1, w + 1)
        assert latsCenter.shape == lonsCenter.shape == (h, w)
        assert elev is None or elev.shape == (h, w)
        if img.ndim == 2:
            img = img[:, :, None]
        assert img.ndim == 3
        assert img.dtype in [np.uint8, np.uint16]
        BaseMapping.__init__(self, alti, cameraPosGCRS, photoTime, identifier, metadata)
        self._inputs = (lats, lons, latsCenter, lonsCenter, elev, img)
        self._frame = None

    def frame(self):
        if self._frame is None:
            lats, lons, latsCenter, lonsCenter, elev, img = self._inputs
            filled = [np.asarray(ma.filled(a.astype(np.float64), np.nan)) if ma.isMA(a)
                      else np.asarray(a, dtype=np.float64) for a in (lats, lons, latsCenter, lonsCenter)]
            corner_mask = np.isnan(filled[0]) | np.isnan(filled[1])
            center_mask = np.isnan(filled[2]) | np.isnan(filled[3])
            img_mask = ma.getmaskarray(img)[:, :, 0] if ma.isMA(img) else None
            e = None
            if elev is not None:
                e = np.asarray(ma.filled(elev.astype(np.float64), np.nan)) if ma.isMA(elev) \
                    else np.asarray(elev, dtype=np.float64)
            fd = FrameData.from_host(filled[0], filled[1], filled[2], filled[3], e, ma.getdata(img),
                                     corner_mask=corner_mask, center_mask=center_mask)
            img_mask_t = None if img_mask is None else fd.ctx.to_device(img_mask.astype(np.uint8), np.uint8)
            # sanitize_data (mapping.py:1063-1125): image mask -> centres, corner/centre consistency
            fd.ctx.call('amt_sanitize_masks', ptr(fd.corner_mask), ptr(fd.center_mask), ptr(img_mask_t),
                        fd.height, fd.width, 0)
            self._frame = fd
            self._inputs = None
        return self._frame

    def createResampled(self, lats, lons, latsCenter, lonsCenter, elevation, img):
        return GenericMapping(lats, lons, latsCenter, lonsCenter, elevation, self.altitude, img,
                              self.cameraPosGCRS, self.photoTime, self.identifier, metadata=self.metadata)

    @staticmethod
    def fromMapping(mapping):
        """Create a :class:`GenericMapping` sharing the device arrays of the given mapping."""
        m = GenericMapping.__new__(GenericMapping)
        BaseMapping.__init__(m, mapping.altitude, mapping.cameraPosGCRS, mapping.photoTime, mapping.identifier,
                             mapping.metadata)
        m._inputs = None
        m._frame = mapping.frame()
        return m


class MappingCollection(object):
    def __init__(self, mappings, identifier, mayOverlap=True):
        """
        A collection of mappings for the same photo time (+- a few seconds) (reference mapping.py:1315-1373).
        """
        self._mappings = mappings
        self._identifier = identifier
        self._mayOverlap = mayOverlap

    identifier = property(lambda self: self._identifier)
    mappings = property(lambda self: self._mappings)
    mayOverlap = property(lambda self: self._mayOverlap)
    empty = property(lambda self: len(self.mappings) == 0)

    def maskedByElevation(self, minElevation=10):
        return MappingCollection([m.maskedByElevation(minElevation) for m in self.mappings],
                                 self.identifier, self.mayOverlap)

    @property
    def boundingBox(self):
        return BoundingBox.mergedBoundingBoxes([m.boundingBox for m in self.mappings])

    @property
    def photoTime(self):
        times = sorted(m.photoTime for m in self.mappings)
        return times[len(times) // 2]

    def __len__(self):
        return len(self._mappings)


@add_metaclass(ABCMeta)
class BaseMappingProvider(object):
    """Base class for all mapping providers (reference mapping.py:1376-1445)."""

    def __init__(self, maxTimeOffset):
        """:param maxTimeOffset: in seconds"""
        self.maxTimeOffset = maxTimeOffset

    @abstractproperty
    def range(self):
        """The dates of the first and last available mappings: datetime tuple (from, to)."""

    @abstractmethod
    def contains(self, date):
        """True if there is a mapping for the given date within +-maxTimeOffset."""

    def containsAny(self, dates):
        """True if there is a mapping for at least one of the given dates within +-maxTimeOffset."""
        return any(self.contains(date) for date in dates)

    @abstractmethod
    def get(self, date):
        """The mapping closest to the given date within +-maxTimeOffset; ValueError when there is none."""

    @abstractmethod
    def getById(self, identifier):
        """The mapping with the given identifier; ValueError when there is none."""

    @abstractmethod
    def getSequence(self, dateBegin=None, dateEnd=None):
        """Generator of mappings ordered by date for the given (inclusive) date range; all of them by default."""


def MaskByElevationProvider(provider, *args, **kw):
    """Wrap the given mapping provider by masking every returned mapping by elevation (mapping.py:1447-1472)."""
    def mask(m):
        return m.maskedByElevation(*args, **kw)

    class MaskingProvider(provider.__class__):
        def get(self, *a, **k):
            return mask(super(MaskingProvider, self).get(*a, **k))

        def getById(self, *a, **k):
            return mask(super(MaskingProvider, self).getById(*a, **k))

        def getSequence(self, *a, **k):
            return map(mask, super(MaskingProvider, self).getSequence(*a, **k))

    wrapped = copy.copy(provider)
    wrapped.__class__ = MaskingProvider
    return wrapped


def inflatedEarthIntersection(cameraToPixelDirection, cameraPos, earthInflation=110, earthModel='wgs84'):
    """
    Return the intersection points with an inflated earth when shooting rays originating at
    `cameraPos` and going in the direction `cameraToPixelDirection` (reference mapping.py:1474-1510).

    :param cameraToPixelDirection: direction vectors from camera to pixel/sky location, shape (n,3)
    :param cameraPos: xyz J2000 coordinates in km
    :param earthInflation: in km, how much to expand the earth when intersecting
    :param earthModel: 'wgs84' (ellipsoid a+h, b+h) or 'sphere' (R_earth + h)
    """
    shape = cameraToPixelDirection.shape
    assert (len(shape) == 1 and shape[0] == 3) or (len(shape) == 2 and shape[1] == 3)
    if earthModel == 'wgs84':
        return ellipsoidLineIntersection(wgs84A + earthInflation, wgs84B + earthInflation, cameraPos,
                                         cameraToPixelDirection)
    elif earthModel == 'sphere':
        return sphereLineIntersection(R_EARTH_KM + earthInflation, cameraPos, cameraToPixelDirection)
    raise ValueError('unsupported earth model: ' + earthModel)


class _SMMapping(GenericMapping):
    @property
    def cameraFootpoint(self):
        mlat, mlt = j2000ToMLatMLT([self.cameraPosGCRS], self.photoTime)
        return Location(mlat[0], mltToSmLon(mlt)[0])


def convertMappingToSM(mapping):
    """
    Return a new mapping with the coordinates transformed to solar magnetic latitudes and
    longitudes (reference mapping.py:1519-1547).  Device arrays are re-used; nothing is copied to the host.
    """
    fd = mapping.frame()
    mlat, mlt = mapping._mlatmlt_tensors(False)
    mlat_c, mlt_c = mapping._mlatmlt_tensors(True)
    new = fd.shallow_copy()
    new.lat, new.lat_c = mlat, mlat_c
    new.lon = (mlt - 12) / (24 / 360)          # mltToSmLon (transform.py:388-401)
    new.lon_c = (mlt_c - 12) / (24 / 360)
    new.mlat = new.mlt = new.mlat_c = new.mlt_c = None
    new.bbox = None
    # masks follow the geodetic masks (astrometry.py:181-182, mapping.py:1540-1546)
    new.corner_mask = fd.corner_mask_tensor()
    new.center_mask = fd.center_mask_tensor()
    sm = _SMMapping.__new__(_SMMapping)
    BaseMapping.__init__(sm, mapping.altitude, mapping.cameraPosGCRS, mapping.photoTime, mapping.identifier)
    sm._inputs = None
    sm._frame = new
    return sm


def convertSMMappingToGeo(mapping):
    """Inverse operation to :func:`convertMappingToSM` (reference mapping.py:1549-1559)."""
    smlats, smlons = mapping.lats.data, mapping.lons.data
    smlatsCenter, smlonsCenter = mapping.latsCenter.data, mapping.lonsCenter.data
    # corners and centres in ONE call (per point the same arithmetic: one round trip to the device instead of two)
    nc = smlats.size
    la, lo = smToLatLon(np.concatenate((smlats.ravel(), smlatsCenter.ravel())),
                        np.concatenate((smlons.ravel(), smlonsCenter.ravel())), mapping.photoTime)
    lats, lons = la[:nc].reshape(smlats.shape), lo[:nc].reshape(smlons.shape)
    latsCenter, lonsCenter = la[nc:].reshape(smlatsCenter.shape), lo[nc:].reshape(smlonsCenter.shape)
    return GenericMapping(lats, lons, latsCenter, lonsCenter, mapping.elevation, mapping.altitude,
                          mapping.img, mapping.cameraPosGCRS, mapping.photoTime, mapping.identifier)
