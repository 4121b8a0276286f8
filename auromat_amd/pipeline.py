"""
Device-resident frame pipeline: georeference -> (mask by elevation) -> bounding box -> grid ->
binned mean, i.e. what ``resample(getMapping(...).maskedByElevation(e), pxPerDeg=...)`` does in the
reference (spacecraft.py:380-426, mapping.py:845-864, resample.py:73-157) on pre-allocated HBM buffers.

Two execution plans give identical results:

* two-pass (always available):
    1. amt_georef_frame        corner/centre lat, lon, elevation (+ MLat/MLT) and the bounding-box reduction
                               of the corners that survive the elevation mask
    2. amt_bin_frame           re-reads centre lat/lon/elevation + image, bins (elevation mask on the fly)
    3. amt_bin_frame_finalize  mean / rounding / flip into the output layout
  The host reads the 8 bounding-box doubles between 1 and 2 and lays out the grid
  (reference resample.py:220-241,281-299).

* single-pass (``fuse=True``; geodetic or MLat/MLT grids, also across the 180 deg discontinuity; not with a pole
  in view): the binning happens inside the georeferencing kernel, so nothing is read back.  Because the grid
  depends on the bounding box the kernel itself produces, an estimate of the box comes first — a cheap pre-pass
  (every 16th corner ray) or, in a sequence, the exact box of the previous frame — the kernel bins into a
  *superset* grid aligned to the same global nodes, and the exact window is cropped by the finalise kernel once
  the exact box is known.  If the exact box is not inside the superset the two-pass plan runs instead.  The
  orchestration is native code (include/auromat_hip.h: amt_pipe_*); :class:`FramePipeline` binds it.

:class:`SequencePipeline` software-pipelines a sequence of frames over several frame buffers (what bench.py times
and ``auromat_amd.sequence`` shards over GPUs); the mapping classes give the same results lazily.
"""
import ctypes as C
import os

import numpy as np

from .frame import FrameData, PaddedFrameData, has_array
from .mapping.astrometry import frame_params, pole_in_view, run_frame
from .mapping.mapping import bounding_box_from_reduction, wrap_at_180
from .resample import cached_grid, grid_coordinates, resample_frame
from ._native import Context, GeorefOut, PipeResult, RunConfig, RunFrame, RunResult, ptr, to_host

NEG_INF = float('-inf')


class EmptyFrame(ValueError):
    """No pixel of the frame is valid (the reference's ValueError of maskedByElevation, mapping.py:858-859)."""


class _GridView(object):
    """
    Output grid as laid out by the native driver (amt_grid): sizes, steps and first centres are available
    at once; everything else (coordinate arrays) comes from the equal :class:`auromat_amd.resample._Grid`,
    built when first asked for.
    """

    def __init__(self, g, pxPerDeg, box):
        self.nx, self.ny = g.nx, g.ny
        self.latStep, self.lonStep = g.lat_step, g.lon_step
        self.lat0, self.lon0 = g.lat_center_first, g.lon_center_first
        self._pxPerDeg, self._box, self._full = pxPerDeg, box, None

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        if self._full is None:
            self._full = cached_grid(self._pxPerDeg, *self._box)
        return getattr(self._full, name)


def earth_rows_of(params, height, min_elevation=None):
    """(r0, r1): the pixel rows of a camera frame's image that a kernel can need (amt_georef_image_rows): outside of them no ray
    reaches the shell or — with `min_elevation` > 0 — no pixel reaches that elevation, and only pixels that survive
    maskedByElevation are ever binned (reference mapping.py:845-864, resample.py:315-321).  Conservative."""
    from . import _native
    r0, r1 = C.c_int32(0), C.c_int32(0)
    me = NEG_INF if min_elevation is None else float(min_elevation)
    _native.lib().amt_georef_image_rows(C.byref(params), me, C.byref(r0), C.byref(r1))
    return max(0, r0.value), min(int(height), r1.value)


def uploaded_rows(rows, height):
    """the rows of a host image :meth:`FramePipeline.set_image` sends for ``rows`` = (r0, r1): the band when it is worth
    it (under nine tenths of the image), else everything"""
    if rows is not None and 0 <= rows[0] < rows[1] <= height and (rows[1] - rows[0]) < 0.9 * height:
        return rows[0], rows[1]
    return 0, int(height)


class FramePipeline(object):
    def __init__(self, width, height, nchan=3, img_dtype=np.uint16, device=None, with_mag=False, alloc_image=True,
                 alloc_coords=True, with_geo=True, padded=False):
        import torch
        self.ctx = ctx = Context.current(device)
        self.width, self.height = int(width), int(height)
        h, w = self.height, self.width
        # padded=True: the per-pixel coordinate arrays are kept in strip-padded rows (amt_georef_out.row_layout: the layout the
        # row kernel writes 8-12 % faster) and compacted into the reference's contiguous arrays when somebody asks for one
        # (PaddedFrameData); for pipelines whose arrays mostly stay where they are — a sequence that bins in the same kernel
        self.padded = bool(padded)
        fd = self.fd = PaddedFrameData(ctx, h, w) if self.padded else FrameData(ctx, h, w)
        self.with_mag = with_mag
        # with_geo=False (needs with_mag): "MLat / MLT only" — the pipeline keeps mlat, mlt, mlat_c, mlt_c and the elevation,
        # which is all resampleMLatMLT consumes (reference resample.py:63-71, mapping.py:1519-1547), and the single-pass
        # launch on the (MLat, SM longitude) grid runs the kernel variant that skips the geodetic half of the arithmetic
        # and its four stores (k_georef_rows<SECOND = 4>).  The geodetic arrays are allocated, and the frame georeferenced
        # again into all nine, only when a frame takes the two-pass plan or somebody asks for them (coordinates()).
        self.with_geo = bool(with_geo) or not with_mag
        self._out = GeorefOut()
        # alloc_coords=False ("grids only"): the single-pass plan then writes no per-pixel coordinate arrays at all (its
        # kernel bins every pixel as soon as its coordinates exist; 480 MB per frame of stores fall away) and the
        # arrays are only allocated, and the frame georeferenced again into them, when a frame has to take the
        # two-pass plan (pole in view, ...) or somebody asks for them
        self._coords_valid = False
        self._kept_valid = False    # the arrays a with_geo=False pipeline keeps hold the last frame
        if alloc_coords:
            self._alloc_coords()
        fd.bbox = ctx.empty((8,))
        fd.img_dtype = np.dtype(img_dtype)
        self._img_torch_dtype = torch.uint8 if fd.img_dtype == np.uint8 else torch.int16
        self._img_shape = (h, w, nchan)
        # alloc_image=False: the caller aliases a device-resident image before the first frame (use_image)
        fd.img = ctx.empty(self._img_shape, self._img_torch_dtype) if alloc_image else None
        self._img_own = fd.img
        self._bbox_host = torch.empty(8, dtype=torch.float64, pin_memory=True)
        self._bbox_event = torch.cuda.Event()
        self._driver = None         # amt_pipe handle of the single-pass plan (created on first use)
        self._fin_stream = None     # the driver's finalise stream as torch sees it (see _finalize_fused)
        self._fused = None          # single-pass launch in flight: its px/deg, later its amt_pipe_result
        self._pole = 0
        self.last_plan = None       # 'single-pass' or 'two-pass': what the last resample() did
        # keep_on_device results of the single-pass plan are produced on the driver's own stream; by default the
        # current stream is ordered behind them at once.  A pipelined caller sets defer_join and calls join()
        # once before it consumes the results, which keeps the stream of big kernels free of wait packets.
        self.defer_join = False
        # the stream that will consume keep_on_device results when it is not the one current at finalisation (a pipelined
        # caller sets it: the outputs are then recorded for it once, at allocation, instead of array by array afterwards)
        self.consumer_stream = None
        # rows of ONE frame sharded over several ranks (auromat_amd.sequence.resample_frame_sharded): an object with
        # box(red) -> red (the 8-number bounding-box reduction combined over the ranks), acc(tensor) (the integer
        # accumulators summed over the ranks, in place) and pole (pole in view of the WHOLE frame); None = whole frames
        self.shard = None
        self.params = None
        self.altitude = None
        self.min_elevation = None
        # corner directions of the last frame when they came from the caller ((h + 1, w + 1, 3) float64 device tensor, J2000;
        # the directions-in form of the pipeline, reference astrometry.py:49-64 with any camera model) instead of the TAN model
        self._dirs = None
        self._out.bbox = fd.bbox.data_ptr()

    def _alloc_coords(self, full=False):
        """Allocate the per-pixel arrays the pipeline keeps (`full`: the geodetic ones of a with_geo=False pipeline too)."""
        fd, ctx = self.fd, self.ctx
        h, w = self.height, self.width

        def alloc(*names):
            for k in names:
                if self.padded:
                    fd.alloc_padded(k)
                else:
                    setattr(fd, k, ctx.empty((h + 1, w + 1) if k in ('lat', 'lon', 'mlat', 'mlt') else (h, w)))

        if not has_array(fd, 'elev'):
            alloc('elev')
            if self.with_mag:
                alloc('mlat', 'mlt', 'mlat_c', 'mlt_c')
        if (full or self.with_geo) and not has_array(fd, 'lat'):
            alloc('lat', 'lon', 'lat_c', 'lon_c')
        self._set_out(full)

    def _set_out(self, full=False):
        """Output pointers of the next launch: every array the pipeline holds, the geodetic ones of a with_geo=False
        pipeline only with `full` (a NULL output is not computed: amt_georef_out)."""
        fd, out = self.fd, self._out
        out.row_layout = 1 if self.padded else 0          # AMT_ROWS_STRIP_PADDED / AMT_ROWS_CONTIGUOUS
        for k in FrameData.COORDS:
            t = fd.padded(k) if self.padded else getattr(fd, k)
            if k in ('lat', 'lon', 'lat_c', 'lon_c') and not (full or self.with_geo):
                t = None
            setattr(out, k, None if t is None else t.data_ptr())

    def _has_all(self):
        return has_array(self.fd, 'elev') and has_array(self.fd, 'lat')

    def _written(self):
        """A launch is about to write the coordinate buffers: contiguous copies of padded arrays are stale."""
        if self.padded:
            self.fd.touch()

    def coordinates(self):
        """Make sure the per-pixel coordinate arrays of the last frame exist (grids-only pipelines compute them on
        demand: one more run of the georeferencing kernel) -> the FrameData."""
        if not self._coords_valid:
            self._alloc_coords(full=True)
            Context.current(self.ctx.device)
            self._out.bbox_min_elevation = NEG_INF if self.min_elevation is None else float(self.min_elevation)
            self._written()
            if self._dirs is not None:
                self.ctx.call('amt_georef_frame_dirs', C.byref(self.params), ptr(self._dirs), C.byref(self._out))
            else:
                self.ctx.call('amt_georef_frame', C.byref(self.params), C.byref(self._out))
            self._set_out()
            self.fd.corner_mask = self.fd.center_mask = None
            self._coords_valid = self._kept_valid = True
        return self.fd

    def __del__(self):
        if getattr(self, '_driver', None):
            try:
                self.ctx._lib.amt_pipe_destroy(self._driver)
            except Exception:
                pass
            self._driver = None

    # -- inputs ---------------------------------------------------------------------------------
    def set_image(self, img, rows=None):
        """Copy an (h, w, c) host image (or device tensor of the same bytes) into the frame buffer.  A pinned host
        tensor of the buffer's dtype (uint16 images as their int16 bits) is copied asynchronously on the current
        stream.  `rows` = (r0, r1), host images only: only that band of rows crosses the link — the rows a camera frame's
        kernels can need (:meth:`earth_rows`); the buffer's other rows are UNDEFINED afterwards (they keep whatever an earlier
        frame left there): the buffer is the binning kernel's input, not a copy of the image for anybody else."""
        import torch
        if self.fd.img is not self._img_own or self._img_own is None:
            # the buffer aliased a caller's device image (use_image): copy into a buffer of our own
            if self._img_own is None:
                self._img_own = self.ctx.empty(self._img_shape, self._img_torch_dtype)
            self.fd.img = self._img_own
        band = uploaded_rows(rows, self.height) != (0, self.height)
        if isinstance(img, torch.Tensor) and not img.is_cuda and img.is_pinned() and img.dtype == self.fd.img.dtype:
            src = img.reshape(self.fd.img.shape)
            if band:
                self.fd.img[rows[0]:rows[1]].copy_(src[rows[0]:rows[1]], non_blocking=True)
            else:
                self.fd.img.copy_(src, non_blocking=True)
            return
        if isinstance(img, torch.Tensor):
            t = self.ctx.to_device(img, self.fd.img_dtype)
            self.fd.img.copy_(t.reshape(self.fd.img.shape))
            return
        a = np.ascontiguousarray(img, dtype=self.fd.img_dtype)
        assert a.size == self.fd.img.numel(), 'image of the wrong size'
        Context.current(self.ctx.device)
        if band:
            a = a.reshape(self._img_shape)
            self.ctx.upload(a[rows[0]:rows[1]], self.fd.img[rows[0]:rows[1]])
            return
        self.ctx.upload(a, self.fd.img)          # through page-locked staging pieces (see Context.upload)

    def earth_rows(self, params, min_elevation=None):
        """(r0, r1): the pixel rows of a camera frame's image a kernel can need (:func:`earth_rows_of`)."""
        return earth_rows_of(params, self.height, min_elevation)

    def forget_inputs(self):
        """Drop what the pipeline holds of its caller's last frame (a 288 MB direction array of a DirectionArrayMapping, the
        parameters): a cached pipeline must not keep them alive.  The frame's arrays can then no longer be computed on demand."""
        self._dirs = None
        self._coords_valid = self._kept_valid = False

    def use_image(self, img):
        """Use a device-resident image of the buffer's layout ((h, w, c) uint8, or uint16 bits as int16) in place:
        nothing is copied; the caller keeps the tensor alive and unchanged until the frame's results exist."""
        assert img.is_cuda and img.dtype == self._img_torch_dtype and img.is_contiguous() and \
            img.numel() == int(np.prod(self._img_shape)), 'device image of the wrong layout'
        self.fd.img = img.view(self._img_shape)

    def is_resident_image(self, img):
        import torch
        return isinstance(img, torch.Tensor) and img.is_cuda and img.dtype == self._img_torch_dtype and \
            img.is_contiguous() and img.numel() == int(np.prod(self._img_shape))

    # -- single-pass plan (native driver, include/auromat_hip.h amt_pipe_*) ----------------------
    def _pipe(self):
        if self._driver is None:
            handle = C.c_void_p()
            self.ctx.call('amt_pipe_create', C.byref(handle))
            self._driver = handle
        return self._driver

    def _pcall(self, name, *args):
        # amt_pipe_* take the driver handle (not the context) as their first argument
        self.ctx.check(getattr(self.ctx._lib, name)(self._pipe(), *args))

    def _finalize_stream(self):
        """The driver's finalise stream, wrapped for torch: tensors that the finalise kernel writes are allocated for
        it (amt_pipe_finalize_stream)."""
        import torch
        if self._fin_stream is None:
            handle = C.c_void_p()
            self._pcall('amt_pipe_finalize_stream', C.byref(handle))
            self._fin_stream = torch.cuda.ExternalStream(handle.value, device=self.ctx.device)
        return self._fin_stream

    def join(self):
        """Order the current stream behind the single-pass driver's finalise kernels (see amt_pipe_join)."""
        if self._driver is not None:
            Context.current(self.ctx.device)
            self._pcall('amt_pipe_join')

    def start_coarse(self, params, min_elevation, magnetic=False, hint=None, dirs=None):
        """
        Enqueue the coarse bounding-box pre-pass for `params` (asynchronous, on the driver's own stream), or, with
        `hint` (the 8 reduction numbers of a neighbouring frame's exact box), skip it: amt_pipe_coarse_hint.
        `dirs`: the frame's corner directions come from the caller (amt_pipe_coarse_dirs samples that array).
        """
        if hint is not None:
            self._pcall('amt_pipe_coarse_hint', (C.c_double * 8)(*hint), 1 if magnetic else 0)
        elif dirs is not None:
            self._pcall('amt_pipe_coarse_dirs', C.byref(params), ptr(dirs),
                        NEG_INF if min_elevation is None else float(min_elevation), 1 if magnetic else 0)
        else:
            self._pcall('amt_pipe_coarse', C.byref(params),
                        NEG_INF if min_elevation is None else float(min_elevation), 1 if magnetic else 0)

    def box_first(self, params, min_elevation, magnetic=False):
        """
        The box pass of the box-first plan (amt_pipe_launch_box): the frame kernel without any output array, image or
        binning -> the exact 8-number bounding-box reduction of the corners that survive maskedByElevation, in (lat, lon) or,
        with `magnetic`, (MLat, SM longitude); [7] = 1 when a pole of those coordinates is in view (camera model).  One
        host synchronisation.  Raises EmptyFrame when no pixel is valid.
        """
        Context.current(self.ctx.device)
        self._pcall('amt_pipe_launch_box', C.byref(params), NEG_INF if min_elevation is None else float(min_elevation),
                    1 if magnetic else 0)
        res = PipeResult()
        self._pcall('amt_pipe_wait', C.byref(res))
        if res.status == 2 or res.bbox[6] == 0:
            self.last_plan = 'empty'
            raise EmptyFrame('minElevation=' + str(min_elevation) + ' would mask all pixels!')
        return list(res.bbox)

    def resolution_from_box(self, params, min_elevation, arcsecPerPx, magnetic=False):
        """
        `resample(mapping, arcsecPerPx=R)` derives px/deg from the mapping's own bounding box (reference resample.py:36-61,
        104-107): the box pass, plateCarreeResolution on its BoundingBox -> ((latPxPerDeg, lonPxPerDeg), reduction).  The
        reduction is what the following single-pass launch takes as its estimate (`hint`), so that launch needs no
        pre-pass and its superset grid always holds the exact one.
        """
        from .resample import plateCarreeResolution
        red = self.box_first(params, min_elevation, magnetic)
        return plateCarreeResolution(bounding_box_from_reduction(red), arcsecPerPx), red

    def _wait_fused(self):
        """amt_pipe_wait once per launch -> the amt_pipe_result."""
        f = self._fused
        if f['result'] is None:
            res = PipeResult()
            self._pcall('amt_pipe_wait', C.byref(res))
            f['result'] = res
        return f['result']

    # -- stages ---------------------------------------------------------------------------------
    def georef(self, wcsHeader, altitude, cameraPosGCRS, photoTime, fast=True, min_elevation=10.0, params=None,
               fuse_pxPerDeg=None, coarse_started=False, fuse_magnetic=False, dirs=None, pole_in_view=-1):
        """
        Stage 1.  `params` (an amt_frame_params made by :func:`frame_params`) skips the host set-up.
        `fuse_pxPerDeg` = (latPxPerDeg, lonPxPerDeg) selects the single-pass plan for that resolution, on the
        geodetic grid or, with `fuse_magnetic`, on the (MLat, SM longitude) grid of resampleMLatMLT.
        `dirs`: (h + 1, w + 1, 3) float64 device tensor of corner directions (J2000) that replaces the TAN camera model
        (`params` then carries camera position, shell and rotations only; fast centres); `pole_in_view` 0 / 1 is the caller's
        knowledge about a pole of the grid's coordinates in such a frame, -1 = unknown (amt_pipe_launch_dirs).
        """
        assert wcsHeader is None or (wcsHeader['IMAGEW'], wcsHeader['IMAGEH']) == (self.width, self.height)
        p = params if params is not None else frame_params(wcsHeader, altitude, cameraPosGCRS, photoTime, fast,
                                                           magnetic=self.with_mag)
        fd = self.fd
        Context.current(self.ctx.device)      # enqueue on whatever stream torch has current now
        out = self._out
        min_elev = NEG_INF if min_elevation is None else float(min_elevation)
        self.params, self.altitude, self.min_elevation = p, altitude, min_elevation
        if dirs is not None:
            assert dirs.is_cuda and dirs.is_contiguous() and tuple(dirs.shape) == (self.height + 1, self.width + 1, 3), \
                'one direction per pixel corner, (h + 1, w + 1, 3) float64 on the device'
        self._dirs = dirs
        self._written()
        if fuse_pxPerDeg is not None and fd.nchan == 3 and (self.with_mag or not fuse_magnetic):
            # coarse pre-pass (unless already enqueued), superset grid, fused kernel, bbox copy: all in the driver
            mag = 1 if fuse_magnetic else 0
            if not coarse_started:
                self.start_coarse(p, min_elevation, mag, dirs=dirs)
            out.altitude = float(altitude)
            if dirs is not None:
                self._pcall('amt_pipe_launch_dirs', C.byref(p), ptr(dirs), C.byref(out), fd.img.data_ptr(),
                            fd.img_dtype_code, min_elev, float(fuse_pxPerDeg[0]), float(fuse_pxPerDeg[1]), int(pole_in_view), mag)
            else:
                self._pcall('amt_pipe_launch', C.byref(p), C.byref(out), fd.img.data_ptr(),
                            fd.img_dtype_code, min_elev, float(fuse_pxPerDeg[0]), float(fuse_pxPerDeg[1]), -1, mag)
            self._fused = dict(pxPerDeg=tuple(fuse_pxPerDeg), magnetic=bool(mag), result=None, pole_in_view=int(pole_in_view))
            fd.corner_mask = fd.center_mask = None
            self._coords_valid = has_array(fd, 'lat') and self.with_geo
            self._kept_valid = has_array(fd, 'elev')
            return fd
        self._alloc_coords(full=True)
        self._coords_valid = self._kept_valid = True
        self._fused = None
        self._pole = None                                # decided lazily in bounding_box()
        out.bbox_min_elevation = min_elev
        if dirs is not None:
            self.ctx.call('amt_georef_frame_dirs', C.byref(p), ptr(dirs), C.byref(out))
        else:
            self.ctx.call('amt_georef_frame', C.byref(p), C.byref(out))
        self._set_out()
        fd.corner_mask = fd.center_mask = None
        # the 8 reduction doubles travel to pinned host memory right behind the kernel; the event lets
        # bounding_box() wait for exactly this point while later launches keep the GPU busy
        self._bbox_host.copy_(fd.bbox, non_blocking=True)
        self._bbox_event.record()
        return fd

    @staticmethod
    def georef_many(pipes, params, altitudes, min_elevation, fuse_pxPerDeg, fuse_magnetic=False, dirs=None, pole_in_view=-1):
        """
        The single-pass launch of :meth:`georef` for up to AMT_PIPE_MAX_BATCH pipelines at once (one frame each, coarse
        pre-pass already started): ONE launch of the big kernel covers all frames (amt_pipe_launch_many).  `dirs`: one
        (h + 1, w + 1, 3) direction tensor per frame instead of the camera model (amt_pipe_launch_dirs_many).
        """
        n = len(pipes)
        ctx = pipes[0].ctx
        Context.current(ctx.device)
        min_elev = NEG_INF if min_elevation is None else float(min_elevation)
        mag = 1 if fuse_magnetic else 0
        if not isinstance(altitudes, (list, tuple)):
            altitudes = [altitudes] * n
        for q, altitude in zip(pipes, altitudes):
            q._out.altitude = float(altitude)
            q._written()
        handles = (C.c_void_p * n)(*[q._pipe() for q in pipes])
        pp = (C.c_void_p * n)(*[C.addressof(p) for p in params])
        oo = (C.c_void_p * n)(*[C.addressof(q._out) for q in pipes])
        ii = (C.c_void_p * n)(*[q.fd.img.data_ptr() for q in pipes])
        # one resolution for all frames, or a list with one (latPxPerDeg, lonPxPerDeg) per frame (arcsecPerPx: every frame's
        # px/deg follows from its own bounding box)
        per_frame = isinstance(fuse_pxPerDeg, list)
        ppd = [tuple(v) for v in fuse_pxPerDeg] if per_frame else [tuple(fuse_pxPerDeg)] * n
        if dirs is not None:
            assert not per_frame and len(dirs) == n
            for d, q in zip(dirs, pipes):
                assert d.is_cuda and d.is_contiguous() and tuple(d.shape) == (q.height + 1, q.width + 1, 3)
            dd = (C.c_void_p * n)(*[d.data_ptr() for d in dirs])
            ctx.check(ctx._lib.amt_pipe_launch_dirs_many(handles, n, pp, dd, oo, ii, pipes[0].fd.img_dtype_code, min_elev,
                                                         float(ppd[0][0]), float(ppd[0][1]), int(pole_in_view), mag))
        elif per_frame:
            la, lo = (C.c_double * n)(*[float(v[0]) for v in ppd]), (C.c_double * n)(*[float(v[1]) for v in ppd])
            ctx.check(ctx._lib.amt_pipe_launch_many_res(handles, n, pp, oo, ii, pipes[0].fd.img_dtype_code, min_elev, la, lo, -1, mag))
        else:
            ctx.check(ctx._lib.amt_pipe_launch_many(handles, n, pp, oo, ii, pipes[0].fd.img_dtype_code, min_elev,
                                                    float(ppd[0][0]), float(ppd[0][1]), -1, mag))
        for i, (q, p, altitude, v) in enumerate(zip(pipes, params, altitudes, ppd)):
            q.params, q.altitude, q.min_elevation = p, altitude, min_elevation
            q._dirs = None if dirs is None else dirs[i]
            q._fused = dict(pxPerDeg=v, magnetic=bool(mag), result=None, pole_in_view=int(pole_in_view))
            q.fd.corner_mask = q.fd.center_mask = None
            q._coords_valid = has_array(q.fd, 'lat') and q.with_geo
            q._kept_valid = has_array(q.fd, 'elev')

    def bounding_box(self):
        """Waits for the fused reduction of the last georef() -> BoundingBox; ValueError if nothing is valid."""
        if self._dirs is not None:
            # caller-supplied directions: no camera model to project a pole through — the corner quads decide
            # (amt_bbox_corners counts those that wind around a pole, reference geodesic.py:183 / mapping.py:705-721)
            if self._fused is not None:
                self._wait_fused()
            self.coordinates()
            red = self._reduce_bbox(self.fd.lat, self.fd.lon)
        elif self._fused is not None and not self._fused['magnetic'] and not \
                (self._wait_fused().fused and self._wait_fused().bbox[7]):
            red = np.array(self._wait_fused().bbox[:])
        elif self._fused is not None:
            # the driver reduced the box over (MLat, SM longitude) or, pole in view, over the rotated corners; the
            # geodetic one comes from the corner arrays
            self.coordinates()
            red = self._reduce_bbox(self.fd.lat, self.fd.lon)
            red[7] = 1.0 if pole_in_view(self.params, self.min_elevation) else 0.0
        else:
            self._bbox_event.synchronize()
            red = self._bbox_host.numpy().copy()
            if self.shard is not None:
                red = self.shard.box(red)
                self._pole = bool(self.shard.pole)
            if self._pole is None:
                # pole containment from the camera model (the kernel does not count pole quads)
                self._pole = pole_in_view(self.params, self.min_elevation)
            red[7] = 1.0 if self._pole else 0.0
        if red[6] == 0:
            self.last_plan = 'empty'
            raise EmptyFrame('minElevation=' + str(self.min_elevation) + ' would mask all pixels!')
        return bounding_box_from_reduction(red)

    def _reduce_bbox(self, lat, lon):
        """8-number bounding-box reduction (see amt_bbox_corners) of corner arrays `lat`, `lon` of this frame over
        the corners that survive maskedByElevation(min_elevation); host array."""
        import torch
        fd = self.fd
        thr = NEG_INF if self.min_elevation is None else self.min_elevation
        cmask = (~(fd.elev >= thr)).to(torch.uint8)
        corner = torch.isnan(fd.lat).to(torch.uint8)
        red = self.ctx.empty((8,))
        # exact centres: a centre also needs its four corners (mapping.py:1093-1101); masking by elevation first
        # and reconciling once gives the same masks as the reference's sanitize -> mask -> sanitize order
        self.ctx.call('amt_sanitize_masks', ptr(corner), ptr(cmask), None, fd.height, fd.width,
                      1 if self.params.fast_center else 0)
        self.ctx.call('amt_bbox_corners', ptr(lat), ptr(lon), ptr(corner), ptr(cmask), fd.height, fd.width, ptr(red))
        return to_host(red)

    def fused_ready(self, pxPerDeg, magnetic):
        """The amt_pipe_result of the single-pass launch in flight when the driver can finalise it for this resolution
        (None: whatever resolution it was launched with) and grid (waits for the frame's bounding box), else None (no such
        launch, or the frame needs the general path)."""
        if self._fused is None or self._fused['magnetic'] != bool(magnetic) or \
                (pxPerDeg is not None and self._fused['pxPerDeg'] != tuple(pxPerDeg)):
            return None
        res = self._wait_fused()
        return res if res.status == 0 else None

    def _fused_nbytes(self, res):
        """Bytes of one frame's outputs (mean | count | image | mask), padded to 256."""
        n = res.grid.ny * res.grid.nx
        return (40 * n + 3 * n * (2 if self.fd.img_dtype_code == 2 else 1) + n + 255) & ~255

    def _fused_alloc(self, nbytes, cur):
        """Output memory for the finalise stream -> (bytes, the same as float64, the same as the image's type)."""
        # The finalise kernel runs on the driver's own stream and does not wait for the current one.  The caching
        # allocator hands out memory per stream: a block freed on the current stream (say the accumulators of a
        # two-pass frame before this one) may go out again at once while kernels queued there still use it — fine
        # for consumers on that stream, fatal for a kernel on another one that writes right away (found by
        # tools/fuzz_sequence.py: frames that fell back to the two-pass plan inside a single-pass sequence came out
        # wrong now and then).  So the outputs are allocated FOR the finalise stream, and the streams that read them
        # are recorded on them.
        import torch
        with torch.cuda.stream(self._finalize_stream()):
            buf = torch.empty(nbytes, dtype=torch.uint8, device=self.ctx.device)
        buf.record_stream(cur)
        if self.consumer_stream is not None and self.consumer_stream != cur:
            buf.record_stream(self.consumer_stream)
        return buf, buf.view(torch.float64), (buf.view(torch.int16) if self.fd.img_dtype_code == 2 else buf)

    def _fused_outputs(self, res, pxPerDeg, mem=None, offset=0):
        """Grid description and output arrays (one allocation for the finalise stream; `mem` = a _fused_alloc shared by
        the frames of a launch, this frame's part starting at byte `offset`) of a frame amt_pipe_wait has laid out ->
        (out dict without the arrays, packed, mean, count, img, mask)."""
        import torch
        ctx, fd = self.ctx, self.fd
        g = res.grid
        b = res.bbox
        wrapped = bool(res.lon_wrapped)
        pole = bool(b[7])                                       # a pole plan: grid in rotated coordinates
        if wrapped:
            # straddles the 180 deg discontinuity: the grid is laid out for longitudes shifted by 180 deg
            box = (b[0], b[1], wrap_at_180(b[4] + 180), wrap_at_180(b[5] + 180))
        else:
            box = (b[0], b[1], b[2], b[3])
        grid = _GridView(g, pxPerDeg, box)
        if mem is None:
            mem = self._fused_alloc(self._fused_nbytes(res), torch.cuda.current_stream(ctx.device))
        buf, b64, bimg = mem
        # (one allocation for the four arrays: mean | count | image | mask)
        n = g.ny * g.nx
        o8 = offset // 8
        packed = b64[o8:o8 + 5 * n]
        mean = b64[o8:o8 + 4 * n].view(g.ny, g.nx, 4)
        count = b64[o8 + 4 * n:o8 + 5 * n].view(g.ny, g.nx)
        if fd.img_dtype_code == 2:
            o16 = offset // 2 + 20 * n
            img = bimg[o16:o16 + 3 * n].view(g.ny, g.nx, 3)
            om = offset + 46 * n
        else:
            img = buf[offset + 40 * n:offset + 43 * n].view(g.ny, g.nx, 3)
            om = offset + 43 * n
        mask = buf[om:om + n].view(g.ny, g.nx)
        # (BoundingBox.containsDiscontinuity is true for every box with a pole in it, reference mapping.py:200-206)
        out = dict(has_elev=True, grid=grid, contains_pole=pole, contains_discontinuity=wrapped or pole,
                   altitude=self.altitude)
        # (the frame's bytes as they lie in the allocation: mean | count | image | mask — one copy takes all four to the host)
        out['_block'] = buf[offset:om + n]
        return out, packed, mean, count, img, mask

    def _fused_wrap(self, out, packed, mean, count, img, mask, keep_on_device):
        fd = self.fd
        block = out.pop('_block')
        if keep_on_device:
            # `packed`: mean and count as they lie in memory, one after the other — the payload of this frame in the
            # gather's wire format (auromat_amd.sequence.pack_results) without a copy per array
            out.update(mean=mean, img=img, mask=mask, count=count, packed=packed)
            return out
        out.update(grid_coordinates(out))
        # ONE device-to-host copy (and one synchronisation) for the four arrays, views on the host
        host = block.cpu().numpy()
        ny, nx = mean.shape[0], mean.shape[1]
        n = ny * nx
        isz = fd.img_dtype.itemsize
        out.update(mean=host[:32 * n].view(np.float64).reshape(ny, nx, 4), count=host[32 * n:40 * n].view(np.float64).reshape(ny, nx),
                   img=host[40 * n:40 * n + 3 * isz * n].view(fd.img_dtype).reshape(ny, nx, 3),
                   mask=host[40 * n + 3 * isz * n:40 * n + 3 * isz * n + n].astype(bool).reshape(ny, nx))
        return out

    def _finalize_fused(self, res, pxPerDeg, keep_on_device):
        """Crop the superset accumulators to the exact grid laid out by amt_pipe_wait."""
        out, packed, mean, count, img, mask = self._fused_outputs(res, pxPerDeg)
        self._pcall('amt_pipe_finalize', ptr(mean), ptr(img), ptr(mask), ptr(count))
        if not (keep_on_device and self.defer_join):
            self.join()
        return self._fused_wrap(out, packed, mean, count, img, mask, keep_on_device)

    @staticmethod
    def finalize_many(pipes, results, pxPerDeg, keep_on_device):
        """:meth:`_finalize_fused` for the frames of one launch with ONE call and ONE kernel (amt_pipe_finalize_many):
        `results` are their amt_pipe_results (all status 0) -> list of result dicts."""
        import torch
        n = len(pipes)
        ctx = pipes[0].ctx
        # the outputs of the launch's frames are ONE allocation when their drivers finalise on the same stream (they do:
        # the tail stream belongs to the context), each frame's part 256-byte aligned
        sizes = [q._fused_nbytes(r) for q, r in zip(pipes, results)]
        fin = pipes[0]._finalize_stream().cuda_stream
        if all(q._finalize_stream().cuda_stream == fin and q.consumer_stream == pipes[0].consumer_stream for q in pipes):
            mem = pipes[0]._fused_alloc(sum(sizes), torch.cuda.current_stream(ctx.device))
            offs = [sum(sizes[:i]) for i in range(n)]
            outs = [q._fused_outputs(r, pxPerDeg or q._fused['pxPerDeg'], mem, o) for q, r, o in zip(pipes, results, offs)]
        else:
            outs = [q._fused_outputs(r, pxPerDeg or q._fused['pxPerDeg']) for q, r in zip(pipes, results)]
        handles = (C.c_void_p * n)(*[q._pipe() for q in pipes])
        arr = lambda k: (C.c_void_p * n)(*[o[k].data_ptr() for o in outs])
        ctx.check(ctx._lib.amt_pipe_finalize_many(handles, n, arr(2), arr(4), arr(5), arr(3)))
        done = []
        for q, o in zip(pipes, outs):
            if not (keep_on_device and q.defer_join):
                q.join()
            q.last_plan = 'single-pass'
            done.append(q._fused_wrap(*o, keep_on_device=keep_on_device))
        return done

    def resample(self, pxPerDeg=10, containsPole=None, magnetic=False, keep_on_device=False):
        """Stages 2 + 3.  magnetic=True bins on the (MLat, SM longitude) grid (resampleMLatMLT)."""
        try:
            _, _ = pxPerDeg
        except TypeError:
            pxPerDeg = (pxPerDeg, pxPerDeg)
        fd = self.fd
        Context.current(self.ctx.device)
        if self._fused is not None and self._fused['pxPerDeg'] == tuple(pxPerDeg) and \
                self._fused['magnetic'] == bool(magnetic):
            res = self._wait_fused()
            # (a direction-array frame whose caller left the pole open and whose box reaches within 5 deg of a pole was handed
            # back by the pole guard of amt_pipe_wait, not for a poor estimate: launching it again cannot help — the driver
            # refuses to fuse that box again —, its corner quads decide, below)
            pole_guard = self._dirs is not None and self._fused.get('pole_in_view', -1) < 0 and \
                (res.bbox[0] <= -85.0 or res.bbox[1] >= 85.0)
            if res.status == 1 and res.fused and res.bbox[6] > 0 and res.edge_pixels <= 16384 and not self._fused.get('retried') \
                    and self.shard is None and not pole_guard:
                # handed back although the launch was fused (the exact box does not fit the superset grid of a poor estimate,
                # the date line judged differently by the pre-pass): once more with the exact box — in the coordinates of
                # the plan, res.bbox[7] — as the estimate, instead of five per-pixel arrays and a separate binning pass
                self.start_coarse(self.params, self.min_elevation, magnetic, hint=list(res.bbox))
                self.georef(None, self.altitude, None, None, bool(self.params.fast_center), self.min_elevation,
                            params=self.params, fuse_pxPerDeg=tuple(pxPerDeg), coarse_started=True, fuse_magnetic=bool(magnetic),
                            dirs=self._dirs, pole_in_view=self._fused.get('pole_in_view', -1))
                self._fused['retried'] = True
                res = self._wait_fused()
            # status 0: the driver could finalise the frame; a caller's containsPole must agree with its decision
            if res.status == 0 and (containsPole is None or bool(containsPole) == bool(res.bbox[7])):
                self.last_plan = 'single-pass'
                return self._finalize_fused(res, tuple(pxPerDeg), keep_on_device)
            if res.status == 2 or res.bbox[6] == 0:
                # no pixel above the threshold: known from the kernel's own reduction — a grids-only pipeline does not
                # allocate (and compute) five per-pixel arrays only to find that out again
                self.last_plan = 'empty'
                raise EmptyFrame('minElevation=' + str(self.min_elevation) + ' would mask all pixels!')
        self.coordinates()          # (grids-only pipelines: the arrays the two-pass plan reads are computed now)
        if not self.params.fast_center and fd.center_mask is None:
            # exact centres carry their own misses: a centre also needs its four corners and a corner a centre
            # (sanitize_data, reference mapping.py:1063-1125) — the separate binning pass takes the reconciled
            # centre mask; fast centres are consistent by construction (astrometry.py:35-40).  The kernel's own
            # bounding box and the single-pass plan apply the same rule per pixel.
            self.ctx.call('amt_sanitize_masks', ptr(fd.corner_mask_tensor()), ptr(fd.center_mask_tensor()), None,
                          fd.height, fd.width, 0)
        if magnetic:
            assert self.with_mag
            sm = fd.shallow_copy()
            sm.lat, sm.lat_c = fd.mlat, fd.mlat_c
            sm.lon, sm.lon_c = (fd.mlt - 12) / (24 / 360), (fd.mlt_c - 12) / (24 / 360)
            # corners of centres that pass the elevation threshold, in SM coordinates
            red = self._reduce_bbox(sm.lat, sm.lon)
            if red[6] == 0:
                self.last_plan = 'empty'
                raise EmptyFrame('minElevation=' + str(self.min_elevation) + ' would mask all pixels!')
            bb = bounding_box_from_reduction(red)
            fd = sm
        else:
            bb = self.bounding_box()
        pole = bb.containsPole if containsPole is None else containsPole
        self.last_plan = 'two-pass'
        return resample_frame(fd, self.altitude, bb, pxPerDeg, bb.containsDiscontinuity, pole,
                              min_elevation=self.min_elevation, keep_on_device=keep_on_device, shard=self.shard)

    def run(self, wcsHeader, altitude, cameraPosGCRS, photoTime, img=None, fast=True, min_elevation=10.0,
            pxPerDeg=10, containsPole=None, magnetic=False, params=None, keep_on_device=False, fuse=False,
            arcsecPerPx=None, dirs=None):
        """One frame end to end; returns the dict of :func:`auromat_amd.resample.resample_frame`.  `arcsecPerPx` (has
        precedence over pxPerDeg, like the reference's resample()): the box-first plan — a box pass, px/deg from the frame's
        own bounding box, then the single-pass launch (``fuse``) or the two-pass plan; the px/deg pair used is in the
        result as 'pxPerDeg'."""
        coarse_started = False
        if img is not None:
            host_array = not hasattr(img, 'is_cuda')
            if host_array and dirs is None and (params is not None or wcsHeader is not None):
                # a host image of a camera frame: the estimate's pre-pass (or the box pass) runs while the image crosses the
                # link, and only the rows a ray can hit cross it (43 % of the bench frame are sky)
                if params is None:
                    params = frame_params(wcsHeader, altitude, cameraPosGCRS, photoTime, fast, magnetic=self.with_mag)
                if fuse and not arcsecPerPx and self.fd.nchan == 3 and (self.with_mag or not magnetic):
                    self.start_coarse(params, min_elevation, bool(magnetic))
                    coarse_started = True
                self.set_image(img, rows=self.earth_rows(params, min_elevation))
            else:
                self.set_image(img)
        assert not (arcsecPerPx and dirs is not None), 'arcsecPerPx: the box-first plan is built on the camera model'
        if arcsecPerPx:
            if params is None:
                params = frame_params(wcsHeader, altitude, cameraPosGCRS, photoTime, fast, magnetic=self.with_mag)
            pxPerDeg, red = self.resolution_from_box(params, min_elevation, arcsecPerPx, magnetic)
            # (a pole in view: the reference's plateCarreeResolution gives no longitude resolution for a box that goes all
            # around, resample.py:47-61 — nothing to bin into, there as here)
            if not pxPerDeg[1] > 0:
                # (the reference fails on its `assert nLon > 1`, resample.py:226-227; raised explicitly: python -O strips asserts)
                raise AssertionError('arcsecPerPx with a pole in view: plateCarreeResolution yields lonPxPerDeg = 0 (as the reference)')
            if fuse and not red[7]:
                self.start_coarse(params, min_elevation, magnetic, hint=red)
                coarse_started = True
        try:
            _, _ = pxPerDeg
        except TypeError:
            pxPerDeg = (pxPerDeg, pxPerDeg)
        self.georef(wcsHeader, altitude, cameraPosGCRS, photoTime, fast, min_elevation, params=params,
                    fuse_pxPerDeg=pxPerDeg if fuse else None, fuse_magnetic=bool(magnetic), coarse_started=coarse_started,
                    dirs=dirs, pole_in_view=-1 if containsPole is None else int(bool(containsPole)))
        res = self.resample(pxPerDeg, containsPole, magnetic, keep_on_device=keep_on_device)
        res['pxPerDeg'] = tuple(pxPerDeg)
        return res

    def host_arrays(self, kept_only=False):
        """Raw (NaN = missing) coordinate arrays of the last frame as NumPy arrays.  `kept_only`: only the arrays the
        pipeline's own launches write (a with_geo=False pipeline: elevation, MLat, MLT), without computing the others."""
        geo = ['lat', 'lon', 'lat_c', 'lon_c']
        names = geo + ['elev'] + (['mlat', 'mlt', 'mlat_c', 'mlt_c'] if self.with_mag else [])
        if kept_only and not self.with_geo and self._kept_valid:
            names = names[4:]
        else:
            self.coordinates()
        return {k: self.fd.host(k) for k in names}


_CLASS_PIPES = {}


def fused_class_pipeline(width, height, img_dtype, magnetic=False):
    """The frame pipeline behind the mapping classes' fused resample() (one per frame size / image type / grid kind and
    device, grids only: its per-pixel arrays are only allocated when a frame has to take the two-pass plan)."""
    ctx = Context.current()
    key = (str(ctx.device), int(width), int(height), np.dtype(img_dtype).str, bool(magnetic))
    pipe = _CLASS_PIPES.get(key)
    if pipe is None:
        if len(_CLASS_PIPES) >= 4:
            _CLASS_PIPES.clear()                # (frames of many sizes: do not hoard their buffers)
        pipe = _CLASS_PIPES[key] = FramePipeline(width, height, img_dtype=img_dtype, with_mag=bool(magnetic), alloc_coords=False)
    return pipe


def _close(a, b):
    """Are two amt_frame_params neighbours in a sequence: same frame size, camera model within 1 % in scale, camera
    within 100 km, boresight and Earth rotation within about half a degree, shell within 30 km?"""
    if (a.width, a.height, a.fast_center) != (b.width, b.height, b.fast_center):
        return False
    if abs(a.a - b.a) > 30.0 or abs(a.b - b.b) > 30.0:
        return False
    # (separately solved frames of one sequence differ in the sixth digit of their CD matrix: the plate scale within 1 %
    # and the reference pixel within 5 px move the box by far less than the superset's margin)
    cd_tol = 0.01 * max(abs(v) for v in a.cd)
    for x, y, tol in ((a.cam, b.cam, 100.0), (a.rot, b.rot, 0.01), (a.m_geo, b.m_geo, 0.01), (a.m_sm, b.m_sm, 0.01),
                      (a.cd, b.cd, cd_tol), (a.crpix, b.crpix, 5.0)):
        for u, v in zip(x, y):
            if abs(u - v) > tol:
                return False
    return True


_STREAMS = {}


def _shared_stream(device, role):
    """The stream of a given role (main / alt / bin / copy) on `device`, shared by every SequencePipeline of the process.
    A GPU serves a process through a handful of hardware queues and HIP maps streams onto them in turn: a pipeline that
    made streams of its own could end up with its main stream on the queue of the driver's helper streams, whose
    kernels WAIT for events — the big kernels then queue behind those waits (measured: the bench variant that came
    second paid 58 us per frame for it, whatever it computed).  With one set of streams per process the mapping is the
    same for every pipeline."""
    import torch
    key = (str(device), role)
    if key not in _STREAMS:
        if not _STREAMS:
            # AMT_STREAM_SHIFT=n (A/B runs): n streams made first, which moves the ones that follow to other hardware queues
            _STREAMS[('shift', '')] = [torch.cuda.Stream(device=device) for _ in range(int(os.environ.get('AMT_STREAM_SHIFT', '0')))]
        _STREAMS[key] = torch.cuda.Stream(device=device)
    return _STREAMS[key]


def _steady(a, b, c, n_ab, n_bc):
    """Frames a, b (n_ab frames apart) and c (n_bc frames after b): same frame size, shell and camera model as `_close`
    asks, c within 400 km of b, and the camera has moved from b to c as it did from a to b (20 % + 5 km)?"""
    if n_ab <= 0 or n_bc <= 0 or n_bc > 16:
        return False
    if (b.width, b.height, b.fast_center) != (c.width, c.height, c.fast_center):
        return False
    if abs(b.a - c.a) > 30.0 or abs(b.b - c.b) > 30.0:
        return False
    for x, y, tol in ((b.cam, c.cam, 400.0), (b.rot, c.rot, 0.05), (b.m_geo, c.m_geo, 0.05), (b.m_sm, c.m_sm, 0.05),
                      (b.crpix, c.crpix, 5.0)):
        for u, v in zip(x, y):
            if abs(u - v) > tol:
                return False
    # the CD matrix turns with the camera's roll (5e-4 per element over 20 s of the real ISS029 sequence): same plate scale
    # within 1 %, and the elements where the pace of a -> b puts them
    scale_b = abs(b.cd[0] * b.cd[3] - b.cd[1] * b.cd[2]) ** 0.5
    scale_c = abs(c.cd[0] * c.cd[3] - c.cd[1] * c.cd[2]) ** 0.5
    if not (scale_b > 0 and abs(scale_c - scale_b) <= 0.01 * scale_b):
        return False
    for i in range(4):
        step = (b.cd[i] - a.cd[i]) / n_ab
        if abs((c.cd[i] - b.cd[i]) - step * n_bc) > 0.3 * abs(step * n_bc) + 0.01 * scale_b:
            return False
    for i in range(3):
        step = (b.cam[i] - a.cam[i]) / n_ab
        if abs((c.cam[i] - b.cam[i]) - step * n_bc) > 0.2 * abs(step * n_bc) + 5.0:
            return False
    for i in range(9):
        step = (b.rot[i] - a.rot[i]) / n_ab
        if abs((c.rot[i] - b.rot[i]) - step * n_bc) > 0.3 * abs(step * n_bc) + 2e-3:
            return False
    return True


_RUN_RESULT_DTYPE = []


def _run_result_dtype():
    """np.dtype of amt_run_result (built once: the conversion of the nested ctypes structure takes ~60 us, which a
    20-frame call would pay twice)."""
    if not _RUN_RESULT_DTYPE:
        _RUN_RESULT_DTYPE.append(np.dtype(RunResult))
    return _RUN_RESULT_DTYPE[0]


class NativeResults(object):
    """
    The results of one :meth:`SequencePipeline.process` call through the native runner (amt_run_process): a read-only
    sequence of the usual per-frame result dicts (None for a frame without a valid pixel), built when first asked for —
    the arrays are views into two arenas the call filled (mean | count of all frames back to back, which is the payload of
    the gather's wire format as it stands; rounded image | mask per frame), so a caller that only gathers the grids
    (:func:`auromat_amd.sequence.gather_device`) touches no per-frame Python object at all.
    """

    def __init__(self, seq, records, grids, images, fallbacks, keep_on_device):
        self._seq, self._rec, self._grids, self._images = seq, records, grids, images
        self._fallbacks, self._keep = fallbacks, keep_on_device
        self._cache = {}
        self._table = np.frombuffer(records, dtype=_run_result_dtype()) if len(records) else None

    def __len__(self):
        return len(self._rec)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if i not in self._cache:
            self._cache[i] = self._build(i)
        return self._cache[i]

    def __add__(self, other):
        return list(self) + list(other)

    def __mul__(self, n):
        return list(self) * n

    def _build(self, i):
        import torch
        r = self._rec[i]
        if r.status in (2, 4):
            return None
        if r.status != 0:
            return self._fallbacks[i]
        seq = self._seq
        fd = seq.pipes[0].fd
        b = r.bbox
        wrapped, pole = bool(r.lon_wrapped), bool(r.contains_pole)
        if wrapped:
            box = (b[0], b[1], wrap_at_180(b[4] + 180), wrap_at_180(b[5] + 180))
        else:
            box = (b[0], b[1], b[2], b[3])
        ppd = (r.lat_px_per_deg, r.lon_px_per_deg)
        grid = _GridView(r.grid, ppd, box)
        ny, nx = r.ny, r.nx
        n = ny * nx
        o = r.grid_offset
        packed = self._grids[o:o + 5 * n]
        mean = self._grids[o:o + 4 * n].view(ny, nx, 4)
        count = self._grids[o + 4 * n:o + 5 * n].view(ny, nx)
        ob = r.image_offset
        if fd.img_dtype != np.uint8:
            img = self._images[ob:ob + 6 * n].view(torch.int16).view(ny, nx, 3)
            om = ob + 6 * n
        else:
            img = self._images[ob:ob + 3 * n].view(ny, nx, 3)
            om = ob + 3 * n
        mask = self._images[om:om + n].view(ny, nx)
        out = dict(has_elev=True, grid=grid, contains_pole=pole, contains_discontinuity=wrapped or pole,
                   altitude=r.altitude, magnetic=seq.magnetic, pxPerDeg=ppd)
        if self._keep:
            out.update(mean=mean, img=img, mask=mask, count=count, packed=packed)
            return out
        out.update(grid_coordinates(out))
        out.update(mean=to_host(mean), img=to_host(img, dtype=fd.img_dtype), mask=to_host(mask).astype(bool),
                   count=to_host(count))
        return out

    # ---- what the gather needs, without per-frame objects (auromat_amd.sequence) -----------------------------------
    def payload(self):
        """(device tensor, length): mean | count of all frames back to back, or None when a frame took another path."""
        if self._table is None or not self._keep or np.any(self._table['status'] == 1):
            return None
        t = self._table
        used = int((5 * t['ny'].astype(np.int64) * t['nx']).sum())
        return self._grids[:used], used

    def descriptors(self, indices):
        """The (n, 12) descriptor table of auromat_amd.sequence.describe_results."""
        t = self._table
        d = np.zeros((len(self), 12), dtype=np.float64)
        if t is None:
            return d
        ok = t['status'] == 0
        g = t['grid']
        d[:, 0], d[:, 1] = t['ny'], t['nx']
        d[:, 2] = np.where(ok, 4, 0)
        d[:, 3], d[:, 4] = g['lat_center_first'], g['lon_center_first']
        d[:, 5], d[:, 6] = g['lat_step'], g['lon_step']
        d[:, 7] = indices
        d[:, 8] = t['contains_pole']
        d[:, 9] = (t['lon_wrapped'] != 0) | (t['contains_pole'] != 0)
        d[:, 10] = t['altitude']
        d[:, 11] = 1.0 if self._seq.magnetic else 0.0
        d[~ok, :7] = 0
        d[~ok, 8:] = 0
        return d


class SequencePipeline(object):
    """
    Software-pipelined processing of a sequence of equally sized frames on one GPU — what the reference does
    with ``map`` over ``getMappingSequence`` followed by ``resample`` (spacecraft.py:326-332, cli/convert.py:185-220).

    Two frame buffers alternate.  While the GPU georeferences + bins frame k+1, the host finishes frame k
    (waits for its bounding box, lays out its grid, enqueues the crop/finalise) and prepares frame k+2 (rotation
    matrices, coarse bounding-box pre-pass).  plan='single-pass' uses the native frame driver (amt_pipe_*) and
    falls back per frame to the two-pass plan where it does not apply; plan='two-pass' bins with a separate
    kernel, optionally on a second stream beside the next frame's ray casting.
    """

    def __init__(self, width, height, nchan=3, img_dtype=np.uint16, device=None, altitude=110, fast=True,
                 min_elevation=10.0, pxPerDeg=10, plan='single-pass', bin_stream=True, shared_image=None,
                 magnetic=False, batch=3, own_image_buffers=True, keep_coordinates=True, launch_streams=1,
                 geodetic_arrays=None, arcsecPerPx=None, padded=None):
        import torch
        assert plan in ('single-pass', 'two-pass')
        try:
            _, _ = pxPerDeg
        except TypeError:
            pxPerDeg = (pxPerDeg, pxPerDeg)
        self.pxPerDeg = tuple(pxPerDeg)
        # arcsecPerPx (has precedence over pxPerDeg, like the reference's resample(); what `auromat-convert --resample
        # --resolution R` passes, cli/convert.py:176-185): every frame's px/deg follows from its own bounding box
        # (plateCarreeResolution, resample.py:36-61) — the box-first plan: a box pass of the frame kernel two batches ahead of
        # the frame's single-pass launch, which takes the exact box as its estimate (process -> _process_box_first)
        self.arcsecPerPx = float(arcsecPerPx) if arcsecPerPx else None
        if self.arcsecPerPx:
            assert plan == 'single-pass' and nchan == 3, 'arcsecPerPx: the box-first plan is a single-pass plan (RGB images)'
            self.pxPerDeg = None
        self.altitude, self.fast, self.min_elevation = altitude, fast, min_elevation
        self.single_pass = plan == 'single-pass' and nchan == 3
        self.nchan = nchan
        self.magnetic = bool(magnetic)          # grids in (MLat, SM longitude): resampleMLatMLT
        # single-pass plan: `batch` frames share one launch of the big kernel (amt_pipe_launch_many; batch frames in
        # flight + batch being prepared = 2 * batch buffers).  Two-pass plan: one frame per launch and four buffers — a
        # buffer's binning pass, which runs beside the NEXT frame's big kernel, must be over before the buffer's next frame
        # starts, and with four that pass was enqueued two frames earlier.  The 14-17 us between two launches on a stream are
        # paid once per batch and the end of a launch is better filled: 0.207 -> 0.200 -> 0.198 ms per frame for
        # 1, 2, 3 frames per launch.
        self.batch = max(1, min(int(batch), 3)) if self.single_pass else 1
        # own_image_buffers=False: every frame of process() brings a device-resident image that is used in place
        # (no per-slot image buffer is allocated).  keep_coordinates=False ("grids only", what a convert run needs):
        # the single-pass plan writes no per-pixel latitude / longitude / elevation arrays, see FramePipeline.
        # geodetic_arrays (magnetic sequences): None / False = the buffers keep what resampleMLatMLT consumes — MLat, MLT
        # (corners and centres) and the elevation — and the single-pass kernel skips the geodetic half of its work
        # (FramePipeline(with_geo=False)); True = all nine per-pixel arrays, for callers that ask for both.  The grids are the
        # same bit for bit either way.
        self.geodetic_arrays = (not self.magnetic) or bool(geodetic_arrays) or not self.single_pass
        # padded (default: on for the single-pass plan, AMT_PADDED_ROWS=0 switches it off for A/B runs): the buffers' per-pixel
        # arrays in strip-padded rows, which the fused kernel writes 8-12 % faster; `pipes[i].fd.lat` & co. stay what they
        # were, contiguous arrays, compacted on request (FramePipeline(padded=True)).  The two-pass plan's binning kernel reads
        # contiguous rows: its buffers are not padded.
        if padded is None:
            padded = os.environ.get('AMT_PADDED_ROWS', '1') != '0'
        self.padded = bool(padded) and self.single_pass
        self.pipes = [FramePipeline(width, height, nchan, img_dtype, device, with_mag=self.magnetic,
                                    alloc_image=own_image_buffers and (shared_image is None or i == 0),
                                    alloc_coords=keep_coordinates or not self.single_pass, with_geo=self.geodetic_arrays,
                                    padded=self.padded)
                      for i in range((3 if self.arcsecPerPx else 2) * self.batch if self.single_pass else 4)]
        self.ctx = self.pipes[0].ctx
        if shared_image is not None:
            # every frame shows the same image (synthetic benchmarks): upload it once, both buffers alias it
            self.pipes[0].set_image(shared_image)
            for q in self.pipes[1:]:
                q.fd.img = self.pipes[0].fd.img
        for q in self.pipes:
            q.defer_join = True             # joined once per process() call
        self.s_main = _shared_stream(self.ctx.device, 'main')
        # launch_streams=2 (single-pass plan): the batches alternate between two streams.  A buffer set is used by every
        # second batch, i.e. always on the same stream, so every ordering between its users stays stream order; what
        # falls away is the order between CONSECUTIVE big kernels, which do not depend on each other: the next one
        # starts while the last waves of the previous one drain.
        self.s_alt = _shared_stream(self.ctx.device, 'alt') if (launch_streams == 2 and self.single_pass) else None
        # two-pass plan: the binning kernel is memory bound and the ray casting FP64 bound, so frame k's binning
        # runs beside frame k+1's ray casting on a second stream
        self.s_bin = _shared_stream(self.ctx.device, 'bin') if (bin_stream and not self.single_pass) else self.s_main
        self._geo_done = [torch.cuda.Event() for _ in self.pipes]
        self._bin_done = [None for _ in self.pipes]
        # per-frame images: uploaded on a copy stream of their own (created on first use: every extra stream competes
        # for the few hardware queues), so that frame k+1's 72 MB cross PCIe while frame k is being computed
        self.s_copy = None
        # events after which nothing enqueued so far reads this buffer's image any more: the single-pass launch on
        # s_main, and the separate binning kernel of a two-pass frame (plan='two-pass', or the per-frame fall-back of
        # the single-pass plan: pole in view, box misjudged, ...) wherever that runs.  An upload into the buffer waits
        # for all of them.
        self._img_busy = [[] for _ in self.pipes]
        self.plans = []                     # plan taken by each frame of the last process() call
        self.use_hints = True               # sequence coherence instead of the coarse pre-pass where possible
        self._hint = None                   # (exact bbox reduction, amt_frame_params, index) of the latest finished frame
        self._hint_prev = None              # ... and of the one finished before it (for the extrapolation, see _box_hint)
        self._frames_done = 0               # frames of earlier process() calls (hints count frames across calls)
        self.hinted = 0                     # frames of the last process() call that needed no pre-pass
        self.uploaded_bytes = 0             # image bytes the last process() call sent over the link (host images)
        # the frame loop in the library (amt_run_process) where it applies: device-resident images, no on_batch hook;
        # AMT_SEQ_NATIVE=0 keeps the Python loop (A/B runs)
        self.native = os.environ.get('AMT_SEQ_NATIVE', '1') != '0'
        self._run = None
        self._run_hints = True
        self._arena_cells = 1 << 16         # grid cells per frame the arenas of the next call are sized for

    def _stream_of(self, k):
        """The stream the batch of frame k is launched on (and its buffers are used on)."""
        if self.s_alt is None:
            return self.s_main
        return self.s_alt if (k // self.batch) % 2 else self.s_main

    def _box_hint(self, k, p):
        """
        Estimate of frame k's bounding-box reduction from frames that are already finished, or None (then the coarse
        pre-pass runs).  The latest finished frame's exact box as it is when that frame is a neighbour of this one
        (`_close`: camera within 100 km); else, in a steady sequence — the two latest finished frames are neighbours
        of each other and the camera has kept its pace — their boxes extrapolated linearly to this frame: a frame is
        prepared two batches ahead of the latest finished one, 20 s of orbit at the ISS's 3 s cadence, which moves
        the box by more than the superset's margin, but smoothly.  A poor estimate costs time (the frame falls back
        to the two-pass plan), never correctness.
        """
        last, prev = self._hint, self._hint_prev
        if last is None:
            return None
        if _close(last[1], p):
            return last[0]
        if prev is None or not _close(prev[1], last[1]) or not _steady(prev[1], last[1], p, last[2] - prev[2], k - last[2]):
            return None
        if bool(prev[0][7]) != bool(last[0][7]) or (last[0][3] - last[0][2] > 180) != (prev[0][3] - prev[0][2] > 180):
            return None                     # a pole or the date line came into view between the two
        f = (k - last[2]) / float(last[2] - prev[2])
        est = [b + f * (b - a) for a, b in zip(prev[0][:6], last[0][:6])] + list(last[0][6:])
        est[0], est[1] = max(est[0], -90.0), min(est[1], 90.0)
        for i in (2, 3, 4, 5):
            est[i] = min(max(est[i], -180.0), 180.0)
        return est

    def _prepare(self, k, frame):
        hdr, cam, t, img = frame[:4]
        alt = frame[4] if len(frame) > 4 and frame[4] is not None else self.altitude     # per-frame shell
        p = hdr if not isinstance(hdr, dict) else frame_params(hdr, alt, cam, t, self.fast, magnetic=self.magnetic)
        if self.single_pass:
            # the superset grid needs an estimate of the frame's bounding box: the exact box of the latest finished
            # frame when this one is its neighbour in the sequence (no kernel at all), else a coarse pre-pass (a
            # tiny kernel on the driver's own stream, which has to find room on a busy GPU)
            hint = self._box_hint(self._frames_done + k, p) if self.use_hints else None
            self.pipes[k % len(self.pipes)].start_coarse(p, self.min_elevation, self.magnetic, hint)
            self.hinted += hint is not None
        return p, cam, t, img, alt

    def _launch(self, k0, prepared, pxPerDeg=None):
        """Launch the frames k0, k0+1, ... (one batch; prepared = their _prepare results).  `pxPerDeg`: a list with one
        resolution per frame instead of the pipeline's."""
        import torch
        nb = len(self.pipes)
        qs = [self.pipes[(k0 + i) % nb] for i in range(len(prepared))]
        two_streams = self.s_bin is not self.s_main
        s_main = self._stream_of(k0)
        with torch.cuda.stream(s_main):
            for i, (q, (p, cam, t, img, alt)) in enumerate(zip(qs, prepared)):
                slot = (k0 + i) % nb
                if two_streams and self._bin_done[slot] is not None:
                    s_main.wait_event(self._bin_done[slot])     # the buffer's previous frame is still being binned
                if img is None:
                    continue
                if q.is_resident_image(img):
                    q.use_image(img)                                 # in place: nothing is overwritten
                    continue
                if self.s_copy is None:
                    self.s_copy = _shared_stream(self.ctx.device, 'copy')
                with torch.cuda.stream(self.s_copy):
                    for busy in self._img_busy[slot]:
                        self.s_copy.wait_event(busy)                 # the buffer's previous image is still being read
                    self._img_busy[slot] = []
                    # a host image: only the rows a ray of this frame can hit cross the link (no kernel reads the others)
                    on_host = not getattr(img, 'is_cuda', False)
                    rows = q.earth_rows(p, self.min_elevation) if (on_host and p is not None) else None
                    q.set_image(img, rows=rows)
                    if on_host:
                        r0, r1 = uploaded_rows(rows, q.height)
                        self.uploaded_bytes += (r1 - r0) * q.width * q.fd.nchan * q.fd.img_dtype.itemsize
                    uploaded = torch.cuda.Event()
                    uploaded.record(self.s_copy)
                s_main.wait_event(uploaded)
            if self.single_pass:
                FramePipeline.georef_many(qs, [pr[0] for pr in prepared], [pr[4] for pr in prepared],
                                          self.min_elevation, pxPerDeg if pxPerDeg is not None else self.pxPerDeg, self.magnetic)
                # a later upload into a buffer of the pipeline's own must wait until this launch has read it; frames that
                # read the caller's device-resident images need no event (each one is a packet on the stream of big
                # kernels: 5 us between two launches)
                own = [i for i, q in enumerate(qs) if q.fd.img is q._img_own]
                if own:
                    done = torch.cuda.Event()
                    done.record(s_main)
                    for i in own:
                        self._img_busy[(k0 + i) % nb].append(done)
            else:
                for i, (q, (p, cam, t, img, alt)) in enumerate(zip(qs, prepared)):
                    q.georef(None, alt, cam, t, self.fast, self.min_elevation, params=p)
                    if two_streams:
                        self._geo_done[(k0 + i) % nb].record(s_main)

    def _finish(self, k, keep_on_device):
        import torch
        slot = k % len(self.pipes)
        q = self.pipes[slot]
        two_streams = self.s_bin is not self.s_main
        s_bin = self.s_bin if two_streams else self._stream_of(k)
        with torch.cuda.stream(s_bin):
            if two_streams:
                s_bin.wait_event(self._geo_done[slot])
            try:
                res = q.resample(self.pxPerDeg if self.pxPerDeg is not None else q._fused['pxPerDeg'], magnetic=self.magnetic,
                                 keep_on_device=keep_on_device)
            except EmptyFrame:
                # no pixel of this frame is valid (looking off the limb, or min_elevation masks everything; the
                # reference raises ValueError there, mapping.py:858-859): the sequence goes on, the frame's place in
                # the results holds None
                res = None
            if res is not None:
                res['magnetic'] = self.magnetic
            if q.last_plan == 'two-pass':
                # the separate binning kernel reads the buffer's image (and coordinate arrays) on this stream
                if two_streams or q.fd.img is q._img_own:
                    ev = torch.cuda.Event()
                    ev.record(s_bin)
                    if q.fd.img is q._img_own:
                        self._img_busy[slot].append(ev)
                    if two_streams:
                        self._bin_done[slot] = ev
        self.plans.append(q.last_plan)
        if q.last_plan == 'single-pass':
            self._hint_prev = self._hint
            self._hint = (list(q._fused['result'].bbox), q.params, self._frames_done + k)
        else:
            self._hint = self._hint_prev = None     # the next frame gets a real pre-pass
        return res

    def _finish_batch(self, k0, n, keep_on_device):
        """The frames k0 .. k0+n-1 of one launch: those the driver can finalise get ONE finalise call and kernel
        (amt_pipe_finalize_many), the others (general path, no valid pixel) go one by one through :meth:`_finish`."""
        import torch
        nb = len(self.pipes)
        if not self.single_pass or os.environ.get('AMT_SEQ_FINISH_MANY') == '0':      # (the switch: A/B runs)
            return [self._finish(k0 + i, keep_on_device) for i in range(n)]
        qs = [self.pipes[(k0 + i) % nb] for i in range(n)]
        ready = [q.fused_ready(self.pxPerDeg, self.magnetic) for q in qs]
        idx = [i for i in range(n) if ready[i] is not None]
        done = {}
        if idx:
            with torch.cuda.stream(self._stream_of(k0)):
                res = FramePipeline.finalize_many([qs[i] for i in idx], [ready[i] for i in idx], self.pxPerDeg, keep_on_device)
            done = dict(zip(idx, res))
        out = []
        for i in range(n):
            if i not in done:
                out.append(self._finish(k0 + i, keep_on_device))
                continue
            r = done[i]
            r['magnetic'] = self.magnetic
            self.plans.append('single-pass')
            self._hint_prev = self._hint
            self._hint = (list(ready[i].bbox), qs[i].params, self._frames_done + k0 + i)
            out.append(r)
        return out

    # ---- the frame loop in the library (amt_run_*, include/auromat_hip.h "native sequence runner") --------------------
    def _native_applies(self, frames):
        """Device-resident images, header dicts, one launch stream; the single-pass plan or, for RGB frames with fast centres on
        a geodetic grid, the two-pass plan: what amt_run_process covers."""
        q = self.pipes[0]
        two_pass_ok = (not self.single_pass and self.nchan == 3 and self.fast and not self.magnetic and
                       os.environ.get('AMT_SEQ_NATIVE_TWO_PASS', '1') != '0')
        if not ((self.single_pass or two_pass_ok) and self.s_alt is None and frames):
            return False
        # (the checks of FramePipeline.is_resident_image, inlined: this runs for every frame of every call)
        dt, numel = q._img_torch_dtype, int(np.prod(q._img_shape))
        for f in frames:
            img = f[3]
            # a device-resident image, or one in page-locked host memory (a pinned torch tensor: amt_run_frame.img_host — the
            # runner uploads the rows of it that can be binned, one batch ahead of the frame's launch)
            if not (type(f[0]) is dict and hasattr(img, 'is_cuda') and img.dtype == dt and img.numel() == numel and
                    img.is_contiguous() and (img.is_cuda or img.is_pinned())):
                return False
        return True

    def _runner(self):
        if self._run is None:
            nb = len(self.pipes)
            slots = (GeorefOut * nb)()
            for i, q in enumerate(self.pipes):
                C.memmove(C.byref(slots[i]), C.byref(q._out), C.sizeof(GeorefOut))
            q = self.pipes[0]
            cfg = RunConfig(width=q.width, height=q.height, img_dtype=1 if q.fd.img_dtype == np.uint8 else 2, fast_center=1 if self.fast else 0,
                            magnetic=1 if self.magnetic else 0, batch=self.batch, use_hints=1 if self.use_hints else 0,
                            n_slots=nb, two_pass=0 if self.single_pass else 1, altitude=float(self.altitude),
                            min_elevation=NEG_INF if self.min_elevation is None else float(self.min_elevation),
                            lat_px_per_deg=float(self.pxPerDeg[0]) if self.pxPerDeg else 0.0,
                            lon_px_per_deg=float(self.pxPerDeg[1]) if self.pxPerDeg else 0.0, slots=slots,
                            arcsec_per_px=self.arcsecPerPx or 0.0)
            handle = C.c_void_p()
            self.ctx.call('amt_run_create', C.byref(cfg), C.byref(handle))
            self._run = handle
            self._run_hints = self.use_hints
        return self._run

    def _hint_native_reset(self):
        self.ctx.check(self.ctx._lib.amt_run_reset_hints(self._run))

    def __del__(self):
        if getattr(self, '_run', None):
            try:
                self.ctx._lib.amt_run_destroy(self._run)
            except Exception:
                pass
            self._run = None

    def _process_native(self, frames, keep_on_device):
        import torch
        ctx = self.ctx
        n = len(frames)
        if self._run is not None and self._run_hints != self.use_hints:
            ctx._lib.amt_run_destroy(self._run)           # (use_hints is part of the runner's configuration)
            self._run = None
        run = self._runner()
        for q in self.pipes:
            q._written()
        if self._hint is not None or self._hint_prev is not None:
            # the Python loop ran in between: its hints are not the runner's
            self._hint = self._hint_prev = None
        rec = (RunResult * n)()
        one = RunFrame()
        lib = ctx._lib
        done_total = 0
        grids = images = None
        arenas = []
        # the big kernels run on the stream all pipelines of the process share (see _shared_stream), behind whatever the
        # caller's stream has queued (its images); the caller's stream is ordered behind the call's work at the end
        caller = torch.cuda.current_stream(ctx.device)
        bpp = 3 if self.pipes[0].fd.img_dtype == np.uint8 else 6
        self.s_main.wait_stream(caller)
        with torch.cuda.stream(self.s_main):
            Context.current(ctx.device)                     # the library enqueues on torch's current stream
            while done_total < n:
                m = n - done_total
                cells = self._arena_cells * m
                g = torch.empty(5 * cells, dtype=torch.float64, device=ctx.device)
                im = torch.empty((bpp + 1) * cells + 256 * m, dtype=torch.uint8, device=ctx.device)
                out = C.cast(C.byref(rec[done_total]), C.POINTER(RunResult))
                ctx.check(lib.amt_run_begin(run, g.data_ptr(), g.numel(), im.data_ptr(), im.numel(), out, m))
                try:
                    # frame by frame: the library launches as soon as a batch is complete, the first after one frame
                    for f in frames[done_total:]:
                        alt = f[4] if len(f) > 4 and f[4] is not None else 0.0
                        im_f = f[3]
                        if im_f.is_cuda:
                            run_frame(f[0], f[1], f[2], alt, im_f.data_ptr(), out=one)
                        else:
                            run_frame(f[0], f[1], f[2], alt, None, out=one, img_host_ptr=im_f.data_ptr())
                        ctx.check(lib.amt_run_push(run, C.byref(one)))
                except Exception:
                    #  a call that failed leaves the runner in no state to go on from: the next call makes a new one
                    lib.amt_run_end(run, None)
                    lib.amt_run_destroy(run)
                    self._run = None
                    raise
                done = C.c_int32(0)
                rc_end = lib.amt_run_end(run, C.byref(done))
                try:
                    ctx.check(rc_end)
                except Exception:
                    lib.amt_run_destroy(run)
                    self._run = None
                    raise
                g.record_stream(caller)
                im.record_stream(caller)
                arenas.append((done_total, done.value, g, im))
                if done.value < m:
                    self._arena_cells *= 4                    # the grids are larger than assumed: the rest again
                    if done.value == 0 and self._arena_cells > (1 << 28):
                        raise MemoryError('amt_run: a single grid does not fit the arena')
                done_total += done.value
        caller.wait_stream(self.s_main)
        if len(arenas) == 1:
            grids, images = arenas[0][2], arenas[0][3]
        else:
            # (rare: the first call met grids larger than assumed) one arena for all frames, offsets shifted
            grids = torch.cat([a[2][:int(sum(5 * rec[k].ny * rec[k].nx for k in range(a[0], a[0] + a[1])))] for a in arenas])
            sizes = [max([0] + [rec[k].image_offset + ((bpp + 1) * rec[k].ny * rec[k].nx + 255) // 256 * 256
                                for k in range(a[0], a[0] + a[1]) if rec[k].status == 0]) for a in arenas]
            images = torch.cat([a[3][:sz] for a, sz in zip(arenas, sizes)])
            goff = ioff = 0
            for a, sz in zip(arenas, sizes):
                for k in range(a[0], a[0] + a[1]):
                    rec[k].grid_offset += goff
                    rec[k].image_offset += ioff
                goff += int(sum(5 * rec[k].ny * rec[k].nx for k in range(a[0], a[0] + a[1])))
                ioff += sz
        # frames the single-pass plan does not cover (box outside its superset grid, ...): the general path, one by one
        table = np.frombuffer(rec, dtype=_run_result_dtype())
        status = table['status']
        self.hinted += int(table['hinted'].sum())
        self.uploaded_bytes += int(table['uploaded_bytes'].sum())
        fallbacks = {}
        max_cells = int((table['ny'].astype(np.int64) * table['nx']).max()) if n else 1
        names = {0: 'single-pass', 2: 'empty', 4: 'pole-without-resolution'}
        if not status.any() and not table['two_pass'].any():
            self.plans.extend(['single-pass'] * n)
        else:
            for k in range(n):
                st = int(status[k])
                if st in names:
                    self.plans.append('two-pass' if st == 0 and table['two_pass'][k] else names[st])
                    continue
                f = frames[k]
                q = self.pipes[0]
                if f[3].is_cuda:
                    q.use_image(f[3])
                else:
                    q.set_image(f[3])
                alt = f[4] if len(f) > 4 and f[4] is not None else self.altitude
                try:
                    ppd = (table['lat_px_per_deg'][k], table['lon_px_per_deg'][k]) if self.arcsecPerPx else self.pxPerDeg
                    res = q.run(f[0], alt, f[1], f[2], fast=self.fast, min_elevation=self.min_elevation, pxPerDeg=ppd,
                                magnetic=self.magnetic, keep_on_device=keep_on_device, fuse=False)
                    res['magnetic'] = self.magnetic
                except EmptyFrame:
                    res = None
                fallbacks[k] = res
                self.plans.append(q.last_plan)
        self._arena_cells = max(self._arena_cells // 2 if len(arenas) == 1 else self._arena_cells, int(max_cells * 1.5) + 64, 1 << 12)
        self._frames_done += n
        return NativeResults(self, rec, grids, images, fallbacks, keep_on_device)

    def _process_box_first(self, frames, keep_on_device, on_batch=None):
        """
        process() with a resolution per frame (arcsecPerPx).  Three stages per batch of frames, software-pipelined over 3 x
        batch frame buffers so that the GPU never waits for the host:

            box(j)     the box pass of the batch's frames (ONE launch of the frame kernel without outputs, image or binning:
                       amt_pipe_launch_box_many), enqueued two batches ahead
            launch(j)  wait for those boxes (long finished), plateCarreeResolution per frame, the exact box as the estimate
                       of the single-pass launch (no pre-pass; its superset grid always holds the exact grid), ONE launch of
                       the fused kernel with a resolution per frame (amt_pipe_launch_many_res)
            finish(j)  wait for the exact boxes of the fused launch, lay out the grids, one finalise kernel

        in the order box(0) box(1) | launch(j) finish(j-1) box(j+2) | ..., i.e. on the stream: box(j+1), fused(j), box(j+2),
        fused(j+1), ...  A frame with a pole of its grid in view has no longitude resolution under plateCarreeResolution
        (reference resample.py:47-61: its box goes all around; the class API and the reference fail on it): its place in
        the results holds None and its entry in `plans` reads 'pole-without-resolution'; a frame without a valid pixel
        yields None / 'empty' as always.
        """
        import torch
        from .resample import plateCarreeResolution
        if self._run is not None:
            self._hint_native_reset()
        self._hint = self._hint_prev = None
        out = []
        it = iter(frames)
        B, nb = self.batch, len(self.pipes)
        caller = torch.cuda.current_stream(self.ctx.device)
        for q in self.pipes:
            q.consumer_stream = caller
        min_elev = NEG_INF if self.min_elevation is None else float(self.min_elevation)
        mag = 1 if self.magnetic else 0
        lib, ctx = self.ctx._lib, self.ctx

        def box(k0):
            """Parameters + box pass of up to B frames starting at k0 -> list of (p, cam, t, img, alt), empty at the end."""
            prepared = []
            for i in range(B):
                f = next(it, None)
                if f is None:
                    break
                hdr, cam, t, img = f[:4]
                alt = f[4] if len(f) > 4 and f[4] is not None else self.altitude
                p = hdr if not isinstance(hdr, dict) else frame_params(hdr, alt, cam, t, self.fast, magnetic=self.magnetic)
                prepared.append((p, cam, t, img, alt))
            if prepared:
                n = len(prepared)
                qs = [self.pipes[(k0 + i) % nb] for i in range(n)]
                with torch.cuda.stream(self.s_main):
                    Context.current(ctx.device)
                    handles = (C.c_void_p * n)(*[q._pipe() for q in qs])
                    pp = (C.c_void_p * n)(*[C.addressof(pr[0]) for pr in prepared])
                    ctx.check(lib.amt_pipe_launch_box_many(handles, n, pp, min_elev, mag))
            return prepared

        def launch(k0, prepared):
            """-> per frame None (no valid pixel) or its (latPxPerDeg, lonPxPerDeg); the others are launched."""
            ppd, live = [], []
            for i, pr in enumerate(prepared):
                q = self.pipes[(k0 + i) % nb]
                res = PipeResult()
                q._pcall('amt_pipe_wait', C.byref(res))
                if res.status == 2 or res.bbox[6] == 0:
                    ppd.append(None)
                    continue
                red = list(res.bbox)
                v = plateCarreeResolution(bounding_box_from_reduction(red), self.arcsecPerPx)
                if red[7] or not v[1] > 0:
                    ppd.append('pole')
                    continue
                q.start_coarse(pr[0], self.min_elevation, self.magnetic, hint=red)
                ppd.append(v)
                live.append(i)
            # the frames of the batch that have a valid pixel, as consecutive runs (a launch takes consecutive buffers)
            i = 0
            while i < len(live):
                j = i
                while j + 1 < len(live) and live[j + 1] == live[j] + 1:
                    j += 1
                self._launch(k0 + live[i], [prepared[m] for m in live[i:j + 1]], pxPerDeg=[ppd[m] for m in live[i:j + 1]])
                i = j + 1
            return ppd

        def finish(k0, ppd):
            res = []
            i = 0
            while i < len(ppd):
                if ppd[i] is None or ppd[i] == 'pole':
                    self.plans.append('empty' if ppd[i] is None else 'pole-without-resolution')
                    res.append(None)
                    i += 1
                    continue
                j = i
                while j + 1 < len(ppd) and isinstance(ppd[j + 1], tuple):
                    j += 1
                part = self._finish_batch(k0 + i, j - i + 1, keep_on_device)
                for r, v in zip(part, ppd[i:j + 1]):
                    if r is not None:
                        r['pxPerDeg'] = v
                res.extend(part)
                i = j + 1
            return res

        batches = {0: box(0)}
        batches[1] = box(len(batches[0])) if batches[0] else []
        starts = {0: 0, 1: len(batches[0])}
        launched = {}
        j = 0
        while batches.get(j):
            launched[j] = launch(starts[j], batches[j])
            if j >= 1:
                done = finish(starts[j - 1], launched.pop(j - 1))
                out.extend(done)
                if on_batch is not None:
                    on_batch(starts[j - 1], done)
                del batches[j - 1]
            starts[j + 2] = starts[j + 1] + len(batches[j + 1])
            batches[j + 2] = box(starts[j + 2]) if batches[j + 1] else []
            j += 1
        if j >= 1:
            done = finish(starts[j - 1], launched.pop(j - 1))
            out.extend(done)
            if on_batch is not None:
                on_batch(starts[j - 1], done)
        with torch.cuda.stream(self.s_main):
            for q in self.pipes:
                q.join()
        caller.wait_stream(self.s_main)
        self._hint = self._hint_prev = None
        self._frames_done += len(out)
        return out

    def finalize_stream(self):
        """The stream the single-pass results are produced on (the drivers' finalise stream, shared by the pipeline's
        buffers): consumers that want to touch results before process() returns enqueue there."""
        return self.pipes[0]._finalize_stream()

    def process(self, frames, keep_on_device=True, on_batch=None):
        """
        frames: iterable of (wcsHeader | amt_frame_params, cameraPosGCRS, photoTime, image | None[, altitude]);
        a fifth element overrides the pipeline's altitude for that frame (several shells in one sequence).
        An image can be a host array, a pinned host tensor of the buffer's dtype (uploaded asynchronously on a copy
        stream) or a device tensor of the buffer's layout, which is used in place.
        Returns the list of per-frame result dicts (see :func:`auromat_amd.resample.resample_frame`) in order;
        with keep_on_device the arrays are device tensors that are valid for work on the current stream.  A frame
        without any valid pixel yields None (the reference raises ValueError for it, mapping.py:858-859).
        `on_batch(k0, results)` is called with the results of every finished launch (frames k0, k0+1, ... of this call, in
        order) while later launches run: work on them belongs on :meth:`finalize_stream` (or behind this call's return).
        """
        import torch
        del self.plans[:]
        self.hinted = 0
        self.uploaded_bytes = 0
        if on_batch is None and self.native and isinstance(frames, (list, tuple)):
            # (an iterator — the convert driver's read-ahead generator of decoded host images — is consumed frame by frame
            # below: its images are not device-resident anyway, and materialising it would hold every decoded image of the
            # sequence in host memory before the first launch)
            if self._native_applies(frames):
                return self._process_native(frames, keep_on_device)
        if self.arcsecPerPx:
            return self._process_box_first(frames, keep_on_device, on_batch)
        if self._run is not None:
            self._hint_native_reset()
        # (the box of the latest finished frame stays from the previous call: a sequence handed over in pieces is still
        # a sequence, and every hint is checked against the new frame's camera before it is used)
        out = []
        it = iter(frames)
        B = self.batch
        caller = torch.cuda.current_stream(self.ctx.device)
        for q in self.pipes:
            q.consumer_stream = caller

        def next_batch(k0, size=B):
            """Prepare up to `size` frames starting at index k0 -> list (empty at the end of the sequence)."""
            prepared = []
            for i in range(size):
                f = next(it, None)
                if f is None:
                    break
                prepared.append(self._prepare(k0 + i, f))
            return prepared

        k = 0                                    # first frame of the batch that is finished next
        # the first launch carries one frame only: the GPU starts after one frame's preparation instead of three
        # (not with two launch streams, where a buffer set must stay on the stream of its batch parity)
        in_flight = next_batch(0, 1 if self.s_alt is None else B)
        if not in_flight:
            return out
        self._launch(0, in_flight)
        ahead = next_batch(len(in_flight))       # prepared, not launched
        while in_flight:
            n_now = len(in_flight)
            if ahead:
                self._launch(k + n_now, ahead)
                following = next_batch(k + n_now + len(ahead))
            else:
                following = []
            finished = self._finish_batch(k, n_now, keep_on_device)
            out.extend(finished)
            if on_batch is not None:
                on_batch(k, finished)
            k += n_now
            in_flight, ahead = ahead, following
        # order the caller's stream behind everything this call enqueued
        with torch.cuda.stream(self.s_main):
            for q in self.pipes:
                q.join()
        cur = caller
        cur.wait_stream(self.s_main)
        if self.s_alt is not None:
            cur.wait_stream(self.s_alt)
        if self.s_bin is not self.s_main:
            cur.wait_stream(self.s_bin)
        if keep_on_device:
            # the result tensors were allocated on the pipeline's streams: tell the caching allocator that the caller's
            # stream uses them too, so that their memory is not handed out again while the caller still reads it.
            # (Single-pass results are ONE allocation, recorded for the caller's stream when it was made — `packed` is
            # their mark; a call per array here cost 1-2 us each after the last kernel, where nothing hides it.)
            for res in out:
                if res is None or 'packed' in res:
                    continue
                for v in res.values():
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(cur)
        self._frames_done += len(out)
        return out
