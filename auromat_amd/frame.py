"""
Device-resident state of one mapping: the raw coordinate / elevation arrays (NaN = missing), the
image and the two masks, as torch tensors in HBM.  The mapping classes are thin views over this;
kernels consume it directly so that georeferencing -> masking -> resampling never leaves the GPU.
"""
import copy

import numpy as np

from ._native import Context, ptr, to_host


class FrameData(object):
    COORDS = ('lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlt', 'mlat_c', 'mlt_c')

    def __init__(self, ctx, height, width):
        self.ctx = ctx
        self.height = int(height)
        self.width = int(width)
        for k in self.COORDS:
            setattr(self, k, None)
        self._img = None           # (H, W, C) uint8 tensor, or int16 tensor holding uint16 bits (see the `img` property)
        self._img_host = None      # a host image that has not been uploaded yet (set_image(lazy=True))
        self.img_dtype = None      # numpy dtype of the image
        self.corner_mask = None    # (H+1, W+1) uint8, 1 = masked; None = "NaN latitude"
        self.center_mask = None    # (H, W) uint8
        self.bbox = None           # 8 doubles on the device, see amt_georef_out.bbox

    # -- construction -------------------------------------------------------------------------
    @classmethod
    def from_host(cls, lats, lons, lats_c, lons_c, elev, img, corner_mask=None, center_mask=None, device=None):
        ctx = Context.current(device)
        h, w = lats_c.shape
        fd = cls(ctx, h, w)
        fd.lat = ctx.to_device(np.asarray(lats, dtype=np.float64))
        fd.lon = ctx.to_device(np.asarray(lons, dtype=np.float64))
        fd.lat_c = ctx.to_device(np.asarray(lats_c, dtype=np.float64))
        fd.lon_c = ctx.to_device(np.asarray(lons_c, dtype=np.float64))
        if elev is not None:
            fd.elev = ctx.to_device(np.asarray(elev, dtype=np.float64))
        if img is not None:
            fd.set_image(img)
        if corner_mask is not None:
            fd.corner_mask = ctx.to_device(np.asarray(corner_mask, dtype=np.uint8), np.uint8)
        if center_mask is not None:
            fd.center_mask = ctx.to_device(np.asarray(center_mask, dtype=np.uint8), np.uint8)
        return fd

    def set_image(self, img, lazy=False):
        """`lazy`: the image crosses PCIe when a kernel first needs it (a mapping that is only asked for its coordinate
        arrays never uploads its 36-72 MB of pixels)."""
        img = np.asarray(img)
        if img.ndim == 2:
            img = img[:, :, None]
        assert img.dtype in (np.uint8, np.uint16), 'image must be uint8 or uint16'
        assert img.shape[:2] == (self.height, self.width) and img.shape[2] <= 4
        self.img_dtype = img.dtype
        if lazy:
            self._img, self._img_host = None, img
        else:
            self._img, self._img_host = self.ctx.to_device(img, img.dtype), None

    @property
    def img(self):
        if self._img is None and self._img_host is not None:
            self._img, self._img_host = self.ctx.to_device(self._img_host, self._img_host.dtype), None
        return self._img

    @img.setter
    def img(self, tensor):
        self._img, self._img_host = tensor, None

    @property
    def nchan(self):
        if self._img is None and self._img_host is not None:
            return int(self._img_host.shape[2])
        return 0 if self._img is None else int(self._img.shape[2])

    @property
    def img_dtype_code(self):
        return 0 if (self._img is None and self._img_host is None) else (1 if self.img_dtype == np.uint8 else 2)

    def shallow_copy(self):
        return copy.copy(self)

    # -- masks ----------------------------------------------------------------------------------
    def corner_mask_tensor(self):
        import torch
        if self.corner_mask is None:
            self.corner_mask = torch.isnan(self.lat).to(torch.uint8)
        return self.corner_mask

    def center_mask_tensor(self):
        import torch
        if self.center_mask is None:
            self.center_mask = torch.isnan(self.lat_c).to(torch.uint8)
        return self.center_mask

    # -- host views -----------------------------------------------------------------------------
    def host(self, name):
        t = getattr(self, name)
        return None if t is None else to_host(t)

    def host_image(self):
        if self._img is None and self._img_host is not None:
            return self._img_host               # never uploaded: the caller's own array
        if self._img is None:
            return None
        return to_host(self._img, dtype=self.img_dtype)

    def host_mask(self, which):
        t = self.corner_mask_tensor() if which == 'corner' else self.center_mask_tensor()
        return to_host(t).astype(bool)

    def host_bbox(self):
        return None if self.bbox is None else to_host(self.bbox)


_UNSET = object()


def _coord_property(name):
    def get(self):
        v = self._plain.get(name, _UNSET)
        return self._compacted(name) if v is _UNSET else v

    def put(self, value):
        # (shallow copies share the dictionaries: copy on write)
        self._plain = dict(self._plain)
        self._plain[name] = value

    return property(get, put, doc='%s: contiguous rows, compacted from the strip-padded buffer when first asked for' % name)


class PaddedFrameData(FrameData):
    """
    FrameData whose coordinate arrays live in STRIP-PADDED rows (include/auromat_hip.h amt_georef_out.row_layout): the layout the
    row kernel writes fastest, for buffers a pipeline owns.  `fd.lat` & co. still are C-contiguous (rows, cols) tensors — what the
    reference's properties return (astrometry.py:118-152) and every other kernel reads —: they are compacted from the padded
    buffer (amt_unpad_rows, one pass over the array on the current stream) when somebody asks, and kept until the buffers are
    written again (:meth:`touch`).  A pipeline that only bins (the single-pass plan) never asks.
    """

    def __init__(self, ctx, height, width):
        self._plain = {}
        FrameData.__init__(self, ctx, height, width)
        self._plain = {}        # names assigned as ordinary tensors (a shallow copy's swapped arrays, None for "not kept")
        self._padded = {}       # name -> (rows, pitch) float64 tensor
        self._cache = {}        # name -> contiguous tensor compacted from _padded[name]
        self.pitch = int(ctx._lib.amt_padded_pitch(self.width))

    def alloc_padded(self, name):
        """The padded buffer of array `name` (allocated on first use) -> tensor (rows, pitch)."""
        t = self._padded.get(name)
        if t is None:
            rows = self.height + (1 if name in ('lat', 'lon', 'mlat', 'mlt') else 0)
            t = self._padded[name] = self.ctx.empty((rows, self.pitch))
            self._plain = {k: v for k, v in self._plain.items() if k != name}
        return t

    def padded(self, name):
        return self._padded.get(name)

    def has(self, name):
        """Is array `name` kept (without compacting it)?"""
        v = self._plain.get(name, _UNSET)
        return (name in self._padded) if v is _UNSET else v is not None

    def touch(self):
        """The padded buffers have been (or are being) written again: compacted copies are stale."""
        self._cache = {}

    def _compacted(self, name):
        src = self._padded.get(name)
        if src is None:
            return None
        t = self._cache.get(name)
        if t is None:
            cols = self.width + (1 if name in ('lat', 'lon', 'mlat', 'mlt') else 0)
            t = self.ctx.empty((src.shape[0], cols))
            Context.current(self.ctx.device)
            self.ctx.call('amt_unpad_rows', ptr(src), int(src.shape[0]), cols, self.width, ptr(t))
            self._cache = dict(self._cache)
            self._cache[name] = t
        return t


for _k in FrameData.COORDS:
    setattr(PaddedFrameData, _k, _coord_property(_k))
del _k


def has_array(fd, name):
    """Does the frame keep array `name`?  (Never compacts a padded buffer.)"""
    return fd.has(name) if isinstance(fd, PaddedFrameData) else getattr(fd, name) is not None


__all__ = ['FrameData', 'PaddedFrameData', 'has_array', 'ptr']
