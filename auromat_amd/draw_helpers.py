"""
Pixel polygons of a mapping for drawing — the data the reference's matplotlib / basemap code consumes (reference
auromat/draw_helpers.py:34-94; the drawing itself is not part of this package).  One polygon per pixel: its four corners in
(lat, lon), clockwise from the upper left, and the pixel's colour.  Every entry point runs ``amt_pixel_polygons``
(csrc/amt_masks.hip) on the device; there is no host implementation beside it.
"""
import numpy as np
import numpy.ma as ma

from ._native import Context, ptr, to_host


class ColorMode:
    matplotlib = 'matplotlib'


def _gather(ctx, lat, lon, img, img_code, nchan, height, width, index, as_float):
    """amt_pixel_polygons for the flat pixel indices `index` (device int64) -> host (verts (n, 4, 2), colours (n, 3))."""
    import torch
    n = int(index.numel())
    verts = ctx.empty((n, 4, 2))
    colors = ctx.empty((n, 3), torch.float64 if as_float else torch.uint8)
    ctx.call('amt_pixel_polygons', ptr(lat), ptr(lon), ptr(img), img_code, nchan, height, width, ptr(index), n, ptr(verts),
             None if as_float else ptr(colors), ptr(colors) if as_float else None)
    return to_host(verts), to_host(colors, dtype=np.float64 if as_float else np.uint8)


def createPolygonsAndColors(latDeg, lonDeg, rgb, colorMode=None):
    """
    Polygons (in lat/lon) and a colour for EVERY pixel of the arrays given, masked or not (reference draw_helpers.py:34-63).

    :param latDeg: latitude for each pixel corner (h+1,w+1), masked or NaN where missing
    :param lonDeg: longitude for each pixel corner (h+1,w+1)
    :param rgb: uint8 RGB array of (h,w,3) shape (a mapping's ``rgb``), optionally masked
    :param colorMode: 'matplotlib' gives colours as floats in [0,1]
    :rtype: verts (h*w,4,2) float64 with NaN at missing corners, colors (h*w,3) — masked like `rgb` when that is masked
    """
    import torch
    h, w = rgb.shape[:2]
    assert ma.getdata(rgb).dtype == np.uint8 and rgb.shape[2] == 3 and np.shape(latDeg) == (h + 1, w + 1), 'uint8 RGB and corner arrays'
    ctx = Context.current()
    corners = [ctx.to_device(np.ascontiguousarray(ma.filled(ma.masked_invalid(a), np.nan), dtype=np.float64)) for a in (latDeg, lonDeg)]
    img = ctx.to_device(np.ascontiguousarray(ma.getdata(rgb)), np.uint8)
    every = torch.arange(h * w, dtype=torch.int64, device=ctx.device)
    verts, colors = _gather(ctx, corners[0], corners[1], img, 1, 3, h, w, every, colorMode == ColorMode.matplotlib)
    if ma.isMaskedArray(rgb):
        colors = ma.masked_array(colors, mask=ma.getmaskarray(rgb).reshape(-1, 3))
    return verts, colors


def filterNanPolygons(verts, colors):
    """The polygons that have a colour: rows whose colour is masked or NaN are dropped (reference draw_helpers.py:65-82)."""
    missing = ma.getmaskarray(colors)[:, 0] if ma.isMaskedArray(colors) else np.isnan(np.asarray(colors, dtype=np.float64)[:, 0])
    keep = ~missing
    return ma.getdata(verts)[keep], ma.getdata(colors)[keep]


def generatePolygonsFromMapping(mapping, colorMode=None, coordsFn=None):
    """
    (verts (n,4,2), colors (n,3)) of the unmasked pixels of a mapping, row-major (reference draw_helpers.py:84-94): the gather
    runs on the mapping's device arrays and only the n polygons travel to the host.

    :param colorMode: 'matplotlib' gives colours as floats in [0,1]
    :param coordsFn: optional function mapping -> (lats, lons) corner arrays to use instead of lats / lons (e.g. MLat / MLT)
    """
    import torch
    fd = mapping.frame()
    ctx = fd.ctx
    lat, lon = fd.lat, fd.lon
    if coordsFn is not None:
        lat, lon = (ctx.to_device(np.ascontiguousarray(ma.filled(ma.masked_invalid(a), np.nan), dtype=np.float64))
                    for a in coordsFn(mapping))
    shown = torch.nonzero(fd.center_mask_tensor().reshape(-1) == 0).reshape(-1).contiguous()
    return _gather(ctx, lat, lon, fd.img, fd.img_dtype_code, fd.nchan, fd.height, fd.width, shown, colorMode == ColorMode.matplotlib)


__all__ = ['ColorMode', 'createPolygonsAndColors', 'filterNanPolygons', 'generatePolygonsFromMapping']
