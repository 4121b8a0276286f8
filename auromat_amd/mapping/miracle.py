"""
FMI MIRACLE all-sky camera mappings (reference auromat/mapping/miracle.py) on the MI355X path: the equidistant
fisheye calibration -> azimuth / elevation -> line of sight in GEO -> shell intersection -> geodetic coordinates
runs as one kernel per point family (``amt_georef_allsky``: corners, centres); masks, bounding box, MLat/MLT and
resampling are the common :class:`BaseMapping` machinery.

The provider's folder scanning and the image decoding are host I/O; :func:`getMapping` does the same file-name
and ``cal.txt`` handling as the reference and decodes the image with Pillow.
"""
import ctypes as C
import datetime
import os
from collections import namedtuple

import numpy as np

from .._native import AllSkyParams, Context, ptr, to_host
from ..coordinates.geodesic import wgs84A, wgs84B
from ..coordinates.transform import Y, Z, geodetic2EcefZero, latLonToJ2000, rotation_matrix
from ..frame import FrameData
from .mapping import BaseMapping, BoundingBox, GenericMapping

fileDateTimeFormat = '%y%m%d_%H%M%S'

# NOTE: xc, yc, k are relative to a 512x512 image; xc is vertical, yc is horizontal (reference miracle.py:28-35)
CalibrationData = namedtuple('CalibrationData', ['station', 'validFrom', 'validTo', 'lat', 'lon', 'xc', 'yc',
                                                 'k', 'rotation', 'boundingBoxSimple'])

_REF_SIZE = 512


def allsky_params(calData, size, altitude, center_offset=0.5):
    """The ``amt_allsky_params`` of a calibration for a `size` x `size` image (reference miracle.py:139-140,
    249-252,314-326)."""
    p = AllSkyParams()
    p.size = int(size)
    scale = size / _REF_SIZE
    p.xc, p.yc, p.k = calData.xc * scale, calData.yc * scale, calData.k * scale
    p.rotation = float(calData.rotation)
    p.center_offset = float(center_offset)
    matLat = rotation_matrix(np.deg2rad(90 - calData.lat), Y)[:3, :3]
    matLon = rotation_matrix(np.deg2rad(-calData.lon), Z)[:3, :3]
    p.to_geo[:] = list(np.dot(matLon, matLat).ravel())     # latitude first, then longitude
    x, y, z = geodetic2EcefZero(np.deg2rad(calData.lat), np.deg2rad(calData.lon))
    p.station[:] = [float(x), float(y), float(z)]
    p.a, p.b = wgs84A + altitude, wgs84B + altitude
    p.a0, p.b0 = wgs84A, wgs84B
    return p


class MIRACLEMapping(BaseMapping):
    """
    A mapping defined using an image and calibration data from FMI MIRACLE (reference miracle.py:120-352).

    :param CalibrationData calData:
    :param image: (n,n,3) or (n,n) uint8 array, or the path of an image file (a caption below a square image is
                  cut off as in the reference, miracle.py:159-163)
    :param photoTime: datetime object
    :param alti: the altitude in km onto which the image is mapped; always 110 for simple=True
    :param bool simple: constant lat-lon grid from the calibration's bounding box instead of intersections
    :param center_offset: index offset of pixel centres (see ``amt_allsky_params.center_offset``)
    """

    def __init__(self, calData, image, photoTime, alti, simple=False, center_offset=0.5):
        station = calData.station.decode() if isinstance(calData.station, bytes) else calData.station
        identifier = station + '.' + photoTime.strftime('%Y.%m.%d.%H.%M.%S')
        x, y, z = geodetic2EcefZero(np.deg2rad(calData.lat), np.deg2rad(calData.lon))
        self.cameraPosGEO = [float(x), float(y), float(z)]
        cameraPosGCRS = latLonToJ2000(calData.lat, calData.lon, 0, photoTime)
        alti = 110 if simple or alti is None else alti
        BaseMapping.__init__(self, alti, np.asarray(cameraPosGCRS, dtype=np.float64).reshape(3), photoTime, identifier)
        self._calData = calData
        self._simple = simple
        self._image = image
        self._center_offset = center_offset
        self._frame = None
        self._azel = {}

    def _image_array(self):
        if isinstance(self._image, str):
            self._image = loadImage(self._image)
        rgb = np.asarray(self._image)
        if rgb.ndim == 2:
            rgb = np.repeat(rgb[:, :, None], 3, 2)
        if rgb.shape[0] != rgb.shape[1]:
            rgb = rgb[:rgb.shape[1], :]                    # caption below the image
            assert rgb.shape == (rgb.shape[1], rgb.shape[1], 3)
        self._image = np.ascontiguousarray(rgb)
        return self._image

    def _params(self):
        return allsky_params(self._calData, self._image_array().shape[0], self.altitude, self._center_offset)

    def frame(self):
        if self._frame is None:
            img = self._image_array()
            n = img.shape[0]
            ctx = Context.current()
            fd = FrameData(ctx, n, n)
            p = self._params()
            fd.elev = ctx.empty((n, n))
            if self._simple:
                lats, lons, latsCenter, lonsCenter = self._simple_grid(n)
                for name, a in (('lat', lats), ('lon', lons), ('lat_c', latsCenter), ('lon_c', lonsCenter)):
                    setattr(fd, name, ctx.to_device(np.ascontiguousarray(a, dtype=np.float64)))
                ctx.call('amt_georef_allsky', C.byref(p), 0, None, ptr(fd.elev), None, None, None)
            else:
                fd.lat, fd.lon = ctx.empty((n + 1, n + 1)), ctx.empty((n + 1, n + 1))
                fd.lat_c, fd.lon_c = ctx.empty((n, n)), ctx.empty((n, n))
                ctx.call('amt_georef_allsky', C.byref(p), 1, None, None, None, ptr(fd.lat), ptr(fd.lon))
                ctx.call('amt_georef_allsky', C.byref(p), 0, None, ptr(fd.elev), None, ptr(fd.lat_c), ptr(fd.lon_c))
            fd.set_image(img)
            # sanitize_data (reference mapping.py:1063-1125) on the NaN masks
            ctx.call('amt_sanitize_masks', ptr(fd.corner_mask_tensor()), ptr(fd.center_mask_tensor()), None, n, n, 0)
            self._frame = fd
        return self._frame

    def _simple_grid(self, n):
        # reference miracle.py:198-212; rows run north -> south, columns west -> east
        bb = self._calData.boundingBoxSimple
        deltaLat = (bb.latNorth - bb.latSouth) / n
        deltaLon = (bb.lonEast - bb.lonWest) / n
        latC = np.linspace(bb.latNorth - deltaLat / 2, bb.latSouth + deltaLat / 2, n)
        lonC = np.linspace(bb.lonWest + deltaLon / 2, bb.lonEast - deltaLon / 2, n)
        lat = np.linspace(bb.latNorth, bb.latSouth, n + 1)
        lon = np.linspace(bb.lonWest, bb.lonEast, n + 1)
        return (np.repeat(lat[:, None], n + 1, 1), np.repeat(lon[None, :], n + 1, 0),
                np.repeat(latC[:, None], n, 1), np.repeat(lonC[None, :], n, 0))

    # -- azimuth / elevation / direction tables (reference miracle.py:214-312) --------------------------------
    def calculateAzEl(self, center=True):
        """Azimuth in [0,360) and elevation in degrees of every pixel centre or corner, as NumPy arrays."""
        key = bool(center)
        if key not in self._azel:
            n = self._image_array().shape[0]
            ctx = Context.current()
            m = n if center else n + 1
            az, el = ctx.empty((m, m)), ctx.empty((m, m))
            ctx.call('amt_georef_allsky', C.byref(self._params()), 0 if center else 1, ptr(az), ptr(el), None, None,
                     None)
            self._azel[key] = (to_host(az), to_host(el))
        return self._azel[key]

    azElCenter = property(lambda self: self.calculateAzEl(True))
    azElCorner = property(lambda self: self.calculateAzEl(False))
    azimuthCenter = property(lambda self: self.calculateAzEl(True)[0])
    azimuthCorner = property(lambda self: self.calculateAzEl(False)[0])
    elevationCorner = property(lambda self: self.calculateAzEl(False)[1])

    def _directions(self, corner):
        n = self._image_array().shape[0] + (1 if corner else 0)
        ctx = Context.current()
        d = ctx.empty((n, n, 3))
        ctx.call('amt_georef_allsky', C.byref(self._params()), 1 if corner else 0, None, None, ptr(d), None, None)
        return to_host(d)

    @property
    def cameraToPixelCornerDirection(self):
        """Direction vector (GEO) for each pixel corner."""
        return self._cached('dir_corner', lambda: self._directions(True))

    @property
    def cameraToPixelCenterDirection(self):
        """Direction vector (GEO) for each pixel center."""
        return self._cached('dir_center', lambda: self._directions(False))

    def createResampled(self, lats, lons, latsCenter, lonsCenter, elevation, img):
        return GenericMapping(lats, lons, latsCenter, lonsCenter, elevation, self.altitude, img,
                              self.cameraPosGCRS, self.photoTime, self.identifier)


def loadImage(imagePath):
    """RGB image of shape (height,width,3) in its native range (reference util/image.py:17-39); needs Pillow."""
    try:
        from PIL import Image
    except ImportError:
        raise ImportError('decoding ' + imagePath + ' needs Pillow; pass the image as an array instead')
    with Image.open(imagePath) as im:
        return np.asarray(im.convert('RGB'))


def getMapping(imagePath, alti=110, simple=False, image=None):
    """
    Mapping of a MIRACLE image named like ``SOD120304_171900_557_1000.jpg`` with the ``cal.txt`` next to it
    (reference miracle.py:354-366), masked below 0.1 deg of elevation.

    :param image: optional already decoded image array (the file is not opened then)
    """
    filename = os.path.basename(imagePath)
    station = filename[:3]
    date = datetime.datetime.strptime(filename[3:16], fileDateTimeFormat)
    calData = getCalibrationData(os.path.join(os.path.dirname(imagePath), 'cal.txt'), station, date)
    mapping = MIRACLEMapping(calData, imagePath if image is None else image, date, alti, simple=simple)
    return mapping.maskedByElevation(0.1)      # .1 to account for rounding errors


def getCalibrationData(path, station, date):
    """The calibration row of `station` valid at `date` (reference miracle.py:368-409)."""
    with open(path) as fp:
        for line in fp:
            tok = line.split()
            if not tok or tok[0].startswith('#') or tok[0] != station:
                continue
            lat, lon, from_, to = (float(v) for v in tok[1:5])
            xc, yc, k, rotation, latP, latM, lonM, lonP = (float(v) for v in tok[5:13])
            fromDateY = int(from_)
            fromDateM = int((from_ - fromDateY) * 12 + 1)
            toDateY = int(to)
            toDateM = int((to - toDateY) * 12 + 1)
            fromDate = datetime.datetime(fromDateY, fromDateM, 1)
            toDate = datetime.datetime(toDateY, toDateM + 1, 1)     # easier than using end of month
            if not fromDate <= date <= toDate:
                continue
            bbSimple = BoundingBox(latSouth=lat + latM, lonWest=lon + lonM, latNorth=lat + latP, lonEast=lon + lonP)
            return CalibrationData(station=station, validFrom=fromDate, validTo=toDate, lat=lat, lon=lon,
                                   xc=xc, yc=yc, k=k, rotation=rotation, boundingBoxSimple=bbSimple)
    raise ValueError('No MIRACLE calibration data found for ' + station + ' station')


__all__ = ['CalibrationData', 'MIRACLEMapping', 'getMapping', 'getCalibrationData', 'allsky_params']
