"""
Device-resident frame pipeline: georeference -> (mask by elevation) -> bounding box -> grid ->
binned mean, i.e. what ``resample(getMapping(...).maskedByElevation(e), pxPerDeg=...)`` does in the
reference (spacecraft.py:380-426, mapping.py:845-864, resample.py:73-157), as three kernel launches
per frame on pre-allocated HBM buffers:

  1. amt_georef_frame   corner/centre lat, lon, elevation (+ MLat/MLT) and the bounding-box reduction
                        of the corners that survive the elevation mask
  2. amt_bin_frame      bin assignment + LDS-privatised integer accumulation (the elevation mask is
                        applied on the fly, no mask array is materialised)
  3. amt_bin_frame_finalize  mean / rounding / flip into the output layout

Between 1 and 2 the host reads the 8 bounding-box doubles and lays out the grid
(reference resample.py:220-241,281-299).  This is the path bench.py times and that
``auromat_amd.sequence`` shards over GPUs; the mapping classes give the same results lazily.
"""
import ctypes as C

import numpy as np

from .frame import FrameData
from .mapping.astrometry import frame_params, pole_in_view
from .mapping.mapping import bounding_box_from_reduction
from .resample import resample_frame
from ._native import Context, GeorefOut, ptr, to_host


class FramePipeline(object):
    def __init__(self, width, height, nchan=3, img_dtype=np.uint16, device=None, with_mag=False):
        import torch
        self.ctx = ctx = Context.current(device)
        self.width, self.height = int(width), int(height)
        h, w = self.height, self.width
        fd = self.fd = FrameData(ctx, h, w)
        fd.lat, fd.lon = ctx.empty((h + 1, w + 1)), ctx.empty((h + 1, w + 1))
        fd.lat_c, fd.lon_c, fd.elev = ctx.empty((h, w)), ctx.empty((h, w)), ctx.empty((h, w))
        if with_mag:
            fd.mlat, fd.mlt = ctx.empty((h + 1, w + 1)), ctx.empty((h + 1, w + 1))
            fd.mlat_c, fd.mlt_c = ctx.empty((h, w)), ctx.empty((h, w))
        fd.bbox = ctx.empty((8,))
        fd.img_dtype = np.dtype(img_dtype)
        fd.img = ctx.empty((h, w, nchan), torch.uint8 if fd.img_dtype == np.uint8 else torch.int16)
        self.with_mag = with_mag
        self._bbox_host = torch.empty(8, dtype=torch.float64, pin_memory=True)
        self._bbox_event = torch.cuda.Event()
        self.params = None
        self.altitude = None
        self.min_elevation = None
        self.events = None

    # -- inputs ---------------------------------------------------------------------------------
    def set_image(self, img):
        """Copy an (h, w, c) host image (or device tensor of the same bytes) into the frame buffer."""
        t = self.ctx.to_device(img, self.fd.img_dtype)
        self.fd.img.copy_(t.reshape(self.fd.img.shape))

    # -- stages ---------------------------------------------------------------------------------
    def georef(self, wcsHeader, altitude, cameraPosGCRS, photoTime, fast=True, min_elevation=10.0, params=None):
        """Stage 1.  `params` (an amt_frame_params made by :func:`frame_params`) skips the host set-up."""
        assert wcsHeader is None or (wcsHeader['IMAGEW'], wcsHeader['IMAGEH']) == (self.width, self.height)
        p = params if params is not None else frame_params(wcsHeader, altitude, cameraPosGCRS, photoTime, fast,
                                                           magnetic=self.with_mag)
        fd = self.fd
        Context.current(self.ctx.device)      # enqueue on whatever stream torch has current now
        out = GeorefOut()
        out.lat, out.lon, out.lat_c, out.lon_c, out.elev = (t.data_ptr() for t in
                                                            (fd.lat, fd.lon, fd.lat_c, fd.lon_c, fd.elev))
        if self.with_mag:
            out.mlat, out.mlt, out.mlat_c, out.mlt_c = (t.data_ptr() for t in
                                                        (fd.mlat, fd.mlt, fd.mlat_c, fd.mlt_c))
        out.bbox = fd.bbox.data_ptr()
        out.bbox_min_elevation = float('-inf') if min_elevation is None else float(min_elevation)
        self.ctx.call('amt_georef_frame', C.byref(p), C.byref(out))
        # the 8 reduction doubles travel to pinned host memory right behind the kernel; the event lets
        # bounding_box() wait for exactly this point while later launches keep the GPU busy
        self._bbox_host.copy_(fd.bbox, non_blocking=True)
        self._bbox_event.record()
        self.params, self.altitude, self.min_elevation = p, altitude, min_elevation
        return fd

    def bounding_box(self):
        """Waits for the fused reduction of the last georef() -> BoundingBox; ValueError if nothing is valid."""
        self._bbox_event.synchronize()
        red = self._bbox_host.numpy().copy()
        if red[6] == 0:
            raise ValueError('minElevation=' + str(self.min_elevation) + ' would mask all pixels!')
        # pole containment from the camera model (the fused kernel does not count pole quads)
        red[7] = 1.0 if pole_in_view(self.params, self.min_elevation) else 0.0
        return bounding_box_from_reduction(red)

    def resample(self, pxPerDeg=10, containsPole=None, magnetic=False, keep_on_device=False):
        """Stages 2 + 3.  magnetic=True bins on the (MLat, SM longitude) grid (resampleMLatMLT)."""
        try:
            _, _ = pxPerDeg
        except TypeError:
            pxPerDeg = (pxPerDeg, pxPerDeg)
        fd = self.fd
        Context.current(self.ctx.device)
        if magnetic:
            assert self.with_mag
            sm = fd.shallow_copy()
            sm.lat, sm.lat_c = fd.mlat, fd.mlat_c
            sm.lon, sm.lon_c = (fd.mlt - 12) / (24 / 360), (fd.mlt_c - 12) / (24 / 360)
            red = self.ctx.empty((8,))
            # corners of centres that pass the elevation threshold, in SM coordinates
            import torch
            cmask = (~(fd.elev >= (float('-inf') if self.min_elevation is None else self.min_elevation))).to(torch.uint8)
            corner = torch.isnan(fd.lat).to(torch.uint8)
            self.ctx.call('amt_sanitize_masks', ptr(corner), ptr(cmask), None, fd.height, fd.width, 1)
            self.ctx.call('amt_bbox_corners', ptr(sm.lat), ptr(sm.lon), ptr(corner), ptr(cmask), fd.height, fd.width,
                          ptr(red))
            bb = bounding_box_from_reduction(to_host(red))
            fd = sm
        else:
            bb = self.bounding_box()
        pole = bb.containsPole if containsPole is None else containsPole
        return resample_frame(fd, self.altitude, bb, pxPerDeg, bb.containsDiscontinuity, pole,
                              min_elevation=self.min_elevation, keep_on_device=keep_on_device)

    def run(self, wcsHeader, altitude, cameraPosGCRS, photoTime, img=None, fast=True, min_elevation=10.0,
            pxPerDeg=10, containsPole=None, magnetic=False, params=None, keep_on_device=False):
        """One frame end to end; returns the dict of :func:`auromat_amd.resample.resample_frame`."""
        if img is not None:
            self.set_image(img)
        self.georef(wcsHeader, altitude, cameraPosGCRS, photoTime, fast, min_elevation, params=params)
        return self.resample(pxPerDeg, containsPole, magnetic, keep_on_device=keep_on_device)

    def host_arrays(self):
        """Raw (NaN = missing) coordinate arrays of the last frame as NumPy arrays."""
        names = ['lat', 'lon', 'lat_c', 'lon_c', 'elev'] + (['mlat', 'mlt', 'mlat_c', 'mlt_c'] if self.with_mag else [])
        return {k: self.fd.host(k) for k in names}
