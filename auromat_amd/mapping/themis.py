"""
THEMIS all-sky imager mappings (reference auromat/mapping/themis.py): the altitude reprojection of the L2
calibration's coordinate tables (``reproject``, themis.py:224-253) as one kernel, and the mapping class over
precomputed arrays with the IDL-style brightness scaling.  Reading the CDF files (spacepy / NASA CDF) and the
download cache are host I/O outside the accelerated path.
"""
import numpy as np
import numpy.ma as ma

from .._native import Context, ptr, to_host
from ..coordinates.geodesic import wgs84A, wgs84B
from .mapping import GenericMapping


def reproject(latLonASI, latsRef, lonsRef, heightRef, heightNew):
    """
    Lines of sight from the imager through the coordinates given for the reference height, intersected with the
    shell at the new height (reference themis.py:224-253).

    :param latLonASI: tuple of latitude,longitude of ASI
    :param latsRef: latitudes of pixel corners for reference height in degrees
    :param lonsRef: longitudes of pixel corners for reference height in degrees
    :param heightRef: reference height in km (above ground)
    :param heightNew: new height in km (above ground)
    :rtype: tuple of reprojected latitudes and longitudes
    """
    latASI, lonASI = latLonASI
    latsRef = np.ascontiguousarray(ma.filled(latsRef, np.nan), dtype=np.float64)
    lonsRef = np.ascontiguousarray(ma.filled(lonsRef, np.nan), dtype=np.float64)
    assert latsRef.shape == lonsRef.shape
    ctx = Context.current()
    la, lo = ctx.to_device(latsRef), ctx.to_device(lonsRef)
    outLa, outLo = ctx.empty(latsRef.shape), ctx.empty(latsRef.shape)
    ctx.call('amt_reproject_altitude', float(latASI), float(lonASI), ptr(la), ptr(lo), latsRef.size,
             float(heightRef), float(heightNew), wgs84A, wgs84B, ptr(outLa), ptr(outLo))
    return to_host(outLa), to_host(outLo)


def bytscl(array, max_=None, min_=None, top=255):
    """IDL BYTSCL for floats (reference themis.py:206-222)."""
    if max_ is None:
        max_ = np.nanmax(array)
    if min_ is None:
        min_ = np.nanmin(array)
    return np.maximum(np.minimum(((top + 0.9999) * (array - min_) / (max_ - min_)).astype(np.int16), top), 0)


class ThemisMapping(GenericMapping):
    """
    A mapping over the coordinate arrays of a THEMIS imager (reference themis.py:110-204); `img` is the (h,w)
    grayscale frame.  ``rgb`` applies the brightness scaling of thm_asi_create_mosaic.pro.
    """

    def __init__(self, lats, lons, latsCenter, lonsCenter, elev, alti, img, cameraPosGCRS, photoTime,
                 station, minBrightness=None, maxBrightness=None):
        assert img.ndim == 2
        identifier = station + '.' + photoTime.strftime('%Y.%m.%d.%H.%M.%S')
        GenericMapping.__init__(self, lats, lons, latsCenter, lonsCenter, elev, alti, img, cameraPosGCRS, photoTime,
                                identifier)
        self.station = station
        self.minBrightness = minBrightness
        self.maxBrightness = maxBrightness

    def brightness_scaled(self, img):
        if self.minBrightness is not None or self.maxBrightness is not None:
            return bytscl(img, min_=self.minBrightness, max_=self.maxBrightness, top=255)
        med = np.median(ma.compressed(img[img > 1]))
        return np.minimum(img / med * 64, 255)

    @property
    def rgb(self):
        return np.require(np.repeat(self.brightness_scaled(self.img), 3, 2), dtype=np.uint8)

    @property
    def rgb_unmasked(self):
        return np.require(np.repeat(self.brightness_scaled(self.img_unmasked), 3, 2), dtype=np.uint8)

    def createResampled(self, lats, lons, latsCenter, lonsCenter, elevation, img):
        return ThemisMapping(lats, lons, latsCenter, lonsCenter, elevation, self.altitude, img[:, :, 0],
                             self.cameraPosGCRS, self.photoTime, self.station, self.minBrightness, self.maxBrightness)


__all__ = ['ThemisMapping', 'reproject', 'bytscl']
