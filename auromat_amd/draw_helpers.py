"""
Pixel polygons of a mapping for drawing (reference auromat/draw_helpers.py:34-94): vertices in lat/lon and one colour
per polygon.  The matplotlib / basemap drawing itself is not part of this package; this is the data it consumes.
"""
import numpy as np
import numpy.ma as ma

from ._native import ptr, to_host


class ColorMode:
    matplotlib = 'matplotlib'


def createPolygonsAndColors(latDeg, lonDeg, rgb, colorMode=None):
    """
    Returns polygons (in lat/lon coords) and a color for each polygon (reference draw_helpers.py:34-63; a pure
    re-arrangement of arrays the caller already holds on the host).

    :param latDeg: latitude for each pixel corner (h+1,w+1)
    :param lonDeg: longitude for each pixel corner (h+1,w+1)
    :param rgb: RGB array of (h,w,3) shape
    :param colorMode: 'matplotlib' normalizes colors to [0,1]
    :rtype: verts of shape (h*w,4,2), colors of shape (h*w,3)
    """
    latLonDeg = ma.dstack((latDeg, lonDeg))
    verts = ma.concatenate((latLonDeg[0:-1, 0:-1], latLonDeg[0:-1, 1:], latLonDeg[1:, 1:], latLonDeg[1:, 0:-1]), axis=2)
    verts = verts.reshape(rgb.shape[0] * rgb.shape[1], 4, 2)
    if colorMode == ColorMode.matplotlib:
        rgb = _normalizeImage(rgb) / 255
    return verts, rgb.reshape(-1, 3)


def _normalizeImage(rgb):
    """(masked) image -> float [0,255], masked entries NaN (reference util/image.py:74-94)"""
    if rgb.dtype == np.uint16:
        rgb = rgb * (255 / 65535)
    elif rgb.dtype != np.uint8:
        raise NotImplementedError('Image format ' + str(rgb.dtype) + ' not supported')
    rgb = np.require(rgb, np.float64)
    if ma.isMaskedArray(rgb):
        rgb = rgb.filled(np.nan)
    return rgb


def filterNanPolygons(verts, colors):
    """Drops the polygons without colour (masked or NaN), reference draw_helpers.py:65-82."""
    if ma.isMaskedArray(colors):
        hasNans = ma.getmaskarray(colors)[:, 0]
    else:
        hasNans = np.isnan(colors[:, 0])
    verts, colors = verts[~hasNans], colors[~hasNans]
    if ma.isMaskedArray(verts):
        verts = verts.data
    if ma.isMaskedArray(colors):
        colors = colors.data
    return verts, colors


def generatePolygonsFromMapping(mapping, colorMode=None, coordsFn=None):
    """
    (verts (n,4,2), colors (n,3)) of the unmasked pixels of a mapping, row-major (reference draw_helpers.py:84-94).
    With the default coordinates the gather runs on the device (``amt_pixel_polygons``) and only the n polygons
    travel to the host.

    :param colorMode: 'matplotlib' normalizes colors to [0,1]
    :param coordsFn: optional function mapping -> (lats, lons) corner arrays (e.g. MLat/MLT); host path
    """
    if coordsFn is not None:
        lats, lons = coordsFn(mapping)
        return filterNanPolygons(*createPolygonsAndColors(lats, lons, mapping.rgb, colorMode))
    import torch
    fd = mapping.frame()
    ctx = fd.ctx
    index = torch.nonzero(fd.center_mask_tensor().reshape(-1) == 0).reshape(-1).contiguous()
    n = int(index.numel())
    verts = ctx.empty((n, 4, 2))
    as_float = colorMode == ColorMode.matplotlib
    colors = ctx.empty((n, 3), torch.float64 if as_float else torch.uint8)
    ctx.call('amt_pixel_polygons', ptr(fd.lat), ptr(fd.lon), ptr(fd.img), fd.img_dtype_code, fd.nchan, fd.height,
             fd.width, ptr(index), n, ptr(verts), None if as_float else ptr(colors), ptr(colors) if as_float else None)
    return to_host(verts), to_host(colors, dtype=np.float64 if as_float else np.uint8)


__all__ = ['ColorMode', 'createPolygonsAndColors', 'filterNanPolygons', 'generatePolygonsFromMapping']
