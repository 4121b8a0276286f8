"""
ctypes binding of libauromat_hip.so (include/auromat_hip.h) and the per-device context.

There is no CPU fallback: if the shared library is missing or no HIP device is visible, every
compute entry point raises.  PyTorch is used only for device memory and the current stream.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

from ._build import LIB_PATH

c_double_p = C.POINTER(C.c_double)
c_void_pp = C.POINTER(C.c_void_p)


class FrameParams(C.Structure):
    """amt_frame_params"""
    _fields_ = [('width', C.c_int32), ('height', C.c_int32), ('fast_center', C.c_int32), ('reserved', C.c_int32),
                ('cd', C.c_double * 4), ('crpix', C.c_double * 2), ('rot', C.c_double * 9),
                ('cam', C.c_double * 3), ('a', C.c_double), ('b', C.c_double),
                ('a0', C.c_double), ('b0', C.c_double), ('m_geo', C.c_double * 9), ('m_sm', C.c_double * 9)]


class AllSkyParams(C.Structure):
    """amt_allsky_params"""
    _fields_ = [('size', C.c_int32), ('reserved', C.c_int32), ('xc', C.c_double), ('yc', C.c_double),
                ('k', C.c_double), ('rotation', C.c_double), ('center_offset', C.c_double),
                ('to_geo', C.c_double * 9), ('station', C.c_double * 3), ('a', C.c_double), ('b', C.c_double),
                ('a0', C.c_double), ('b0', C.c_double)]


class GeorefOut(C.Structure):
    """amt_georef_out"""
    _fields_ = [(k, C.c_void_p) for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev', 'mlat', 'mlt', 'mlat_c',
                                          'mlt_c', 'bbox')] + [('bbox_min_elevation', C.c_double)] + \
               [(k, C.c_void_p) for k in ('bin_xaxis', 'bin_yaxis', 'bin_img', 'bin_acc')] + \
               [(k, C.c_int32) for k in ('bin_img_dtype', 'bin_lon_wrap', 'bin_magnetic', 'item_order')] + \
               [('bin_events', C.c_void_p), ('bin_event_count', C.c_void_p), ('bin_event_capacity', C.c_int64),
                ('bin_pole', C.c_int32), ('row_layout', C.c_int32), ('altitude', C.c_double)]


SIP_MAX = 10


class ZenithalWcs(C.Structure):
    """amt_zenithal_wcs"""
    _fields_ = [(k, C.c_int32) for k in ('width', 'height', 'corner', 'projection')] + \
               [('cd', C.c_double * 4), ('crpix', C.c_double * 2), ('rot', C.c_double * 9), ('start_x', C.c_double),
                ('start_y', C.c_double), ('sip_order_a', C.c_int32), ('sip_order_b', C.c_int32),
                ('sip_a', (C.c_double * SIP_MAX) * SIP_MAX), ('sip_b', (C.c_double * SIP_MAX) * SIP_MAX)]


class Axis(C.Structure):
    """amt_axis"""
    _fields_ = [('edges', C.c_void_p), ('nbin', C.c_int32), ('uniform', C.c_int32),
                ('first', C.c_double), ('last', C.c_double), ('step', C.c_double), ('scale', C.c_double),
                ('last_rounded', C.c_double)]


class Grid(C.Structure):
    """amt_grid"""
    _fields_ = [('nx', C.c_int32), ('ny', C.c_int32), ('n_lat_nodes', C.c_int32), ('n_lon_nodes', C.c_int32)] + \
               [(k, C.c_double) for k in ('lat_lo', 'lat_hi', 'lon_lo', 'lon_hi', 'lat_step', 'lon_step',
                                          'lat_center_first', 'lat_center_last', 'lon_center_first',
                                          'lon_center_last')] + [('xaxis', Axis), ('yaxis', Axis)]


class SeqFrame(C.Structure):
    """amt_seq_frame"""
    _fields_ = [('ny', C.c_int32), ('nx', C.c_int32), ('nc', C.c_int32), ('index', C.c_int32),
                ('lat0', C.c_double), ('lon0', C.c_double), ('dlat', C.c_double), ('dlon', C.c_double),
                ('contains_pole', C.c_int32), ('contains_discontinuity', C.c_int32), ('magnetic', C.c_int32),
                ('reserved', C.c_int32), ('altitude', C.c_double), ('mean', C.c_void_p), ('count', C.c_void_p)]


class PipeResult(C.Structure):
    """amt_pipe_result"""
    _fields_ = [('status', C.c_int32), ('fused', C.c_int32), ('lon_wrapped', C.c_int32), ('edge_pixels', C.c_int32),
                ('bbox', C.c_double * 8), ('grid', Grid)]


class RunConfig(C.Structure):
    """amt_run_config"""
    _fields_ = [(k, C.c_int32) for k in ('width', 'height', 'img_dtype', 'fast_center', 'magnetic', 'batch', 'use_hints',
                                         'n_slots', 'two_pass', 'reserved_')] + \
               [(k, C.c_double) for k in ('altitude', 'min_elevation', 'lat_px_per_deg', 'lon_px_per_deg')] + \
               [('slots', C.POINTER(GeorefOut)), ('arcsec_per_px', C.c_double)]


class RunFrame(C.Structure):
    """amt_run_frame"""
    _fields_ = [('crval', C.c_double * 2), ('crpix', C.c_double * 2), ('cd', C.c_double * 4), ('lonpole', C.c_double),
                ('cam', C.c_double * 3), ('jd', C.c_double), ('altitude', C.c_double), ('img', C.c_void_p),
                ('img_host', C.c_void_p)]


class RunResult(C.Structure):
    """amt_run_result"""
    _fields_ = [(k, C.c_int32) for k in ('status', 'slot', 'ny', 'nx', 'contains_pole', 'lon_wrapped', 'hinted',
                                         'edge_pixels', 'two_pass', 'reserved_')] + \
               [('grid_offset', C.c_int64), ('image_offset', C.c_int64), ('bbox', C.c_double * 8), ('altitude', C.c_double),
                ('grid', Grid), ('params', FrameParams), ('lat_px_per_deg', C.c_double), ('lon_px_per_deg', C.c_double),
                ('retried', C.c_int32), ('reserved2_', C.c_int32), ('uploaded_bytes', C.c_int64)]


ABI_VERSION = 6          # include/auromat_hip.h AMT_ABI_VERSION
_I, _L, _D, _P = C.c_int, C.c_int64, C.c_double, C.c_void_p
_SIGNATURES = {
    'amt_abi_version': ([], _I),
    'amt_ctx_create': ([_I, _P, _I, c_void_pp], _I),
    'amt_ctx_destroy': ([_P], _I),
    'amt_ctx_set_stream': ([_P, _P], _I),
    'amt_ctx_get_stream': ([_P], _P),
    'amt_ctx_synchronize': ([_P], _I),
    'amt_last_error': ([_P], C.c_char_p),
    'amt_device_info': ([_P, C.c_char_p, C.c_size_t, C.POINTER(_I), C.POINTER(_I), C.POINTER(C.c_size_t)], _I),
    'amt_malloc': ([_P, C.c_size_t, c_void_pp], _I),
    'amt_free': ([_P, _P], _I),
    'amt_host_threads': ([_I, C.POINTER(_I), C.POINTER(_I)], _I),
    'amt_malloc_host': ([_P, C.c_size_t, c_void_pp], _I),
    'amt_free_host': ([_P, _P], _I),
    'amt_memcpy_h2d': ([_P, _P, _P, C.c_size_t], _I),
    'amt_memcpy_d2h': ([_P, _P, _P, C.c_size_t], _I),
    'amt_memset': ([_P, _P, _I, C.c_size_t], _I),
    'amt_event_create': ([_P, c_void_pp], _I),
    'amt_event_destroy': ([_P, _P], _I),
    'amt_event_record': ([_P, _P], _I),
    'amt_event_elapsed_ms': ([_P, _P, _P, C.POINTER(C.c_float)], _I),
    'amt_timing_enable': ([_P, _I], _I),
    'amt_timing_read': ([_P, _I, C.POINTER(C.c_double), C.POINTER(_I)], _I),
    'amt_pipe_launch_box': ([_P, C.POINTER(FrameParams), _D, _I], _I),
    'amt_pipe_launch_box_many': ([c_void_pp, C.c_int32, c_void_pp, _D, _I], _I),
    'amt_pipe_launch_many_res': ([c_void_pp, C.c_int32, c_void_pp, c_void_pp, c_void_pp, C.c_int32, _D, c_double_p, c_double_p, _I, _I], _I),
    'amt_plate_carree_resolution': ([_D, _D, _D, _D, _D, c_double_p, c_double_p], _I),
    'amt_upload_staged': ([_P, _P, _P, C.c_size_t], _I),
    'amt_padded_pitch': ([C.c_int32], C.c_int64),
    'amt_unpad_rows': ([_P, _P, C.c_int32, C.c_int32, C.c_int32, _P], _I),
    'amt_download_staged': ([_P, _P, _P, C.c_size_t], _I),
    'amt_georef_last_variant': ([_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)], _I),
    'amt_directions_tan': ([_P, C.POINTER(FrameParams), _I, _P], _I),
    'amt_directions_zenithal': ([_P, C.POINTER(ZenithalWcs), _P], _I),
    'amt_directions_tan_points': ([_P, C.POINTER(FrameParams), _P, _P, _L, _I, _P], _I),
    'amt_intersect_ellipsoid': ([_P, _D, _D, c_double_p, _P, _L, _I, _P], _I),
    'amt_intersects_ellipsoid': ([_P, _D, _D, c_double_p, _P, _L, _I, _P], _I),
    'amt_intersect_sphere': ([_P, _D, c_double_p, _P, _L, _I, _P], _I),
    'amt_ecef_to_geodetic': ([_P, _P, _P, _P, _L, _D, _D, _P, _P], _I),
    'amt_geodetic_to_ecef': ([_P, _P, _P, _D, _L, _D, _D, _P, _P, _P], _I),
    'amt_rotate_to_latlon': ([_P, c_double_p, _P, _L, _D, _D, _P, _P], _I),
    'amt_rotate_to_mlat_mlt': ([_P, c_double_p, _P, _L, _P, _P], _I),
    'amt_rotate_vectors': ([_P, c_double_p, _P, _L, _P], _I),
    'amt_latlon_to_mlat_mlt': ([_P, c_double_p, _P, _P, _D, _L, _D, _D, _P, _P], _I),
    'amt_sm_to_latlon': ([_P, c_double_p, _P, _P, _L, _D, _D, _P, _P], _I),
    'amt_rotate_pole': ([_P, c_double_p, _P, _P, _D, _L, _D, _D, _P, _P], _I),
    'amt_rotate_pole_deg': ([_P, c_double_p, _P, _P, _D, _L, _D, _D, _P, _P], _I),
    'amt_cartesian_to_spherical': ([_P, _P, _P, _P, _L, _P, _P, _P], _I),
    'amt_spherical_to_cartesian': ([_P, _P, _P, _P, _L, _P, _P, _P], _I),
    'amt_georef_allsky': ([_P, C.POINTER(AllSkyParams), _I, _P, _P, _P, _P, _P], _I),
    'amt_reproject_altitude': ([_P, _D, _D, _P, _P, _L, _D, _D, _D, _D, _P, _P], _I),
    'amt_georef_frame': ([_P, C.POINTER(FrameParams), C.POINTER(GeorefOut)], _I),
    'amt_georef_frame_dirs': ([_P, C.POINTER(FrameParams), _P, C.POINTER(GeorefOut)], _I),
    'amt_georef_coarse_bbox': ([_P, C.POINTER(FrameParams), C.c_int32, _D, _I, _P], _I),
    'amt_delaunay_create': ([_P, C.c_int64, C.POINTER(C.c_void_p)], _I),
    'amt_delaunay_destroy': ([_P], _I),
    'amt_delaunay_sizes': ([_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)], _I),
    'amt_delaunay_stats': ([_P, _P], _I),
    'amt_delaunay_slots': ([_P, c_void_pp, c_void_pp, C.POINTER(C.c_int64)], _I),
    'amt_delaunay_create_threads': ([_P, C.c_int64, C.c_int32, C.c_int64, C.POINTER(C.c_void_p)], _I),
    'amt_delaunay_build_info': ([_P, _P], _I),
    'amt_delaunay_triangles': ([_P, _P, _P], _I),
    'amt_delaunay_vertex_neighbours': ([_P, _P, _P], _I),
    'amt_delaunay_locate': ([_P, _P, C.c_int64, _P, _P, _P], _I),
    'amt_cubic_gradients_csr': ([_P, _P, C.c_int64, _P, _P, _P, C.c_int32, _P, C.c_int32, _D, C.c_int32, _P, _P], _I),
    'amt_cubic_eval': ([_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P], _I),
    'amt_georef_coarse_bbox_dirs': ([_P, C.POINTER(FrameParams), _P, C.c_int32, _D, _I, _P], _I),
    'amt_pipe_coarse_dirs': ([_P, C.POINTER(FrameParams), _P, _D, _I], _I),
    'amt_pipe_launch_dirs': ([_P, C.POINTER(FrameParams), _P, C.POINTER(GeorefOut), _P, C.c_int32, _D, _D, _D, _I, _I], _I),
    'amt_georef_sky_rows': ([C.POINTER(FrameParams)] + [C.POINTER(C.c_int32)] * 4, _I),
    'amt_georef_image_rows': ([C.POINTER(FrameParams), _D] + [C.POINTER(C.c_int32)] * 2, _I),
    'amt_mask_by_elevation': ([_P, _P, _P, C.c_int32, C.c_int32, _D, _P, _P, _P], _I),
    'amt_sanitize_masks': ([_P, _P, _P, _P, C.c_int32, C.c_int32, _I], _I),
    'amt_bbox_corners': ([_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P], _I),
    'amt_mask_outline_links': ([_P, _P, C.c_int32, C.c_int32, _P, _L, _P], _I),
    'amt_pixel_polygons': ([_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _L, _P, _P, _P], _I),
    'amt_hist2d_accumulate': ([_P, _P, _P, _L, c_void_pp, C.c_int32, C.POINTER(Axis), C.POINTER(Axis), _I, _P,
                               c_void_pp], _I),
    'amt_hist2d_finalize_mean': ([_P, _P, c_void_pp, C.c_int32, C.c_int32, C.c_int32, _P], _I),
    'amt_bin_frame': ([_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _D, C.POINTER(Axis),
                       C.POINTER(Axis), _I, _P], _I),
    'amt_bin_frame_finalize': ([_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P], _I),
    'amt_bin_frame_finalize_window': ([_P, _P] + [C.c_int32] * 8 + [_P, _P, _P, _P], _I),
    'amt_nearest_frame': ([_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _D, C.POINTER(Axis), C.POINTER(Axis), _I, _P, _P, _P,
                           _P], _I),
    'amt_nearest_gather': ([_P, _P, _L, _P, C.c_int32, C.c_int32, _P, _P, _P, _P], _I),
    'amt_points_in_polygon': ([_P, _P, _P, _L, _P, C.c_int32, _P], _I),
    'amt_grid_layout': ([_D, _D, _D, _D, _D, _D, C.POINTER(Grid)], _I),
    'amt_pipe_create': ([_P, c_void_pp], _I),
    'amt_pipe_destroy': ([_P], _I),
    'amt_pipe_coarse': ([_P, C.POINTER(FrameParams), _D, _I], _I),
    'amt_pipe_coarse_hint': ([_P, c_double_p, _I], _I),
    'amt_pipe_launch': ([_P, C.POINTER(FrameParams), C.POINTER(GeorefOut), _P, C.c_int32, _D, _D, _D, _I, _I], _I),
    'amt_pipe_launch_many': ([c_void_pp, C.c_int32, c_void_pp, c_void_pp, c_void_pp, C.c_int32, _D, _D, _D, _I, _I], _I),
    'amt_pipe_launch_dirs_many': ([c_void_pp, C.c_int32, c_void_pp, c_void_pp, c_void_pp, c_void_pp, C.c_int32, _D, _D, _D, _I, _I], _I),
    'amt_pipe_wait': ([_P, C.POINTER(PipeResult)], _I),
    'amt_pipe_finalize': ([_P, _P, _P, _P, _P], _I),
    'amt_pipe_finalize_stream': ([_P, _P], _I),
    'amt_pipe_finalize_many': ([_P, C.c_int32, _P, _P, _P, _P], _I),
    'amt_pipe_join': ([_P], _I),
    'amt_pipe_set_plan': ([_P, _I], _I),
    'amt_pipe_general_layout': ([_P, C.POINTER(PipeResult)], _I),
    'amt_pipe_general_finalize': ([_P, _P, _P, _P, _P], _I),
    'amt_frame_params_from_wcs': ([C.POINTER(RunFrame), C.c_int32, C.c_int32, C.c_int32, _D, C.c_int32,
                                   C.POINTER(FrameParams)], _I),
    'amt_run_create': ([_P, C.POINTER(RunConfig), c_void_pp], _I),
    'amt_run_destroy': ([_P], _I),
    'amt_run_process': ([_P, C.POINTER(RunFrame), C.c_int32, _P, _L, _P, _L, C.POINTER(RunResult), C.POINTER(C.c_int32)], _I),
    'amt_run_begin': ([_P, _P, _L, _P, _L, C.POINTER(RunResult), C.c_int32], _I),
    'amt_run_push': ([_P, C.POINTER(RunFrame)], _I),
    'amt_run_end': ([_P, C.POINTER(C.c_int32)], _I),
    'amt_run_reset_hints': ([_P], _I),
    'amt_seq_payload_size': ([C.POINTER(SeqFrame), C.c_int32, C.POINTER(_L)], _I),
    'amt_seq_pack': ([_P, C.POINTER(SeqFrame), C.c_int32, C.c_int32, _P, _L], _I),
    'amt_seq_unpack': ([_P, _L, C.c_int32, C.c_int32, C.POINTER(SeqFrame), C.c_int32, C.POINTER(C.c_int32)], _I),
}

_lib = None
_lib_lock = threading.Lock()


class NativeError(RuntimeError):
    pass


def host_threads(wanted):
    """Threads a host-side pool of this process may use when it would like `wanted`: the cores the process may run on divided by
    the ranks that share the node (LOCAL_WORLD_SIZE of torch.distributed.run, or AMT_LOCAL_RANKS), at least 1 — the rule of the
    library's own pools (include/auromat_hip.h amt_host_threads; the same arithmetic here, so that modules which never load the
    GPU library can size their pools).  Eight ranks on one host each start a copy pool, a triangulator and writer threads."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    ranks = 1
    for name in ('AMT_LOCAL_RANKS', 'LOCAL_WORLD_SIZE'):
        try:
            v = int(os.environ.get(name, '0'))
        except ValueError:
            v = 0
        if v > 0:
            ranks = v
            break
    return max(1, min(int(wanted), max(1, cores // ranks)))


def host_threads_report():
    """What the pools of this process were sized from and to (bench.py's per_rank records carry it)."""
    cores, ranks = C.c_int(0), C.c_int(0)
    share = lib().amt_host_threads(1 << 20, C.byref(cores), C.byref(ranks))
    copy = int(os.environ.get('AMT_COPY_THREADS', '8'))
    return dict(cores_available=cores.value, local_ranks=ranks.value, share=share,
                copy_threads=max(1, min(copy, (share + 1) // 2, 16)),
                triangulator_threads=int(os.environ.get('AMT_DELAUNAY_THREADS', '0')) or lib().amt_host_threads(16, None, None),
                io_threads=host_threads(16), omp_num_threads=os.environ.get('OMP_NUM_THREADS'))


def lib():
    """The loaded shared library (symbols typed). Raises NativeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeError('libauromat_hip.so is not built (%s missing); run `python -c "import '
                                  '__graft_entry__ as g; g.build()"` or `python -m auromat_amd._build`. '
                                  'There is no CPU fallback.' % LIB_PATH)
            # torch first: its bundled libamdhip64.so has SONAME libamdhip64.so.7, which satisfies our
            # DT_NEEDED so that torch tensors and our kernels share one HIP runtime
            import torch  # noqa: F401
            handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
            for name, (argtypes, restype) in _SIGNATURES.items():
                fn = getattr(handle, name)
                fn.argtypes = argtypes
                fn.restype = restype
            if handle.amt_abi_version() != ABI_VERSION:
                raise NativeError('ABI version mismatch: the library is %d, this package expects %d (rebuild: '
                                  'python -c "import __graft_entry__ as g; g.build()")' % (handle.amt_abi_version(), ABI_VERSION))
            _lib = handle
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


class Context(object):
    """One amt_ctx bound to a torch device and to torch's current stream on it."""
    _cache = {}
    _cache_lock = threading.Lock()

    def __init__(self, device_index):
        import torch
        self._lib = lib()
        if not torch.cuda.is_available():
            raise NativeError('no HIP device is visible; auromat_amd has no CPU fallback')
        self.device = torch.device('cuda', device_index)
        self.stream_handle = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        rc = self._lib.amt_ctx_create(device_index, C.c_void_p(self.stream_handle), 0, C.byref(h))
        if rc != 0:
            raise NativeError('amt_ctx_create failed (%d)' % rc)
        self.handle = h
        self._staging = None            # page-locked staging pieces of upload()

    @classmethod
    def current(cls, device=None):
        """Context for `device` (default: torch's current device), re-bound to torch's current stream."""
        import torch
        if not torch.cuda.is_available():
            raise NativeError('no HIP device is visible; auromat_amd has no CPU fallback')
        if device is None:
            idx = torch.cuda.current_device()
        else:
            idx = torch.device(device).index
            if idx is None:
                idx = torch.cuda.current_device()
        key = (os.getpid(), threading.get_ident(), idx)
        with cls._cache_lock:
            ctx = cls._cache.get(key)
            if ctx is None:
                ctx = cls._cache[key] = cls(idx)
        stream = torch.cuda.current_stream(ctx.device).cuda_stream
        if stream != ctx.stream_handle:
            ctx.check(ctx._lib.amt_ctx_set_stream(ctx.handle, C.c_void_p(stream)))
            ctx.stream_handle = stream
        return ctx

    def check(self, rc):
        if rc != 0:
            msg = self._lib.amt_last_error(self.handle)
            raise NativeError('libauromat_hip: %s (code %d)' % (msg.decode() if msg else '?', rc))

    def call(self, name, *args):
        self.check(getattr(self._lib, name)(self.handle, *args))

    def synchronize(self):
        self.call('amt_ctx_synchronize')

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, khz, mem = C.c_int(), C.c_int(), C.c_size_t()
        self.call('amt_device_info', name, 256, C.byref(cus), C.byref(khz), C.byref(mem))
        return dict(name=name.value.decode(), compute_units=cus.value, clock_khz=khz.value, total_mem=mem.value)

    # -- device arrays (torch tensors) ------------------------------------------------------
    def empty(self, shape, dtype=None):
        import torch
        return torch.empty(shape, dtype=dtype or torch.float64, device=self.device)

    def zeros(self, shape, dtype=None):
        import torch
        return torch.zeros(shape, dtype=dtype or torch.float64, device=self.device)

    def to_device(self, array, dtype=np.float64):
        """Host array (or device tensor) -> contiguous device tensor of `dtype`."""
        import torch
        if isinstance(array, torch.Tensor):
            t = array.to(self.device)
            want = _torch_dtype(dtype)
            if t.dtype != want:
                t = t.to(want)
            return t.contiguous()
        a = np.ascontiguousarray(array, dtype=dtype)
        if a.dtype == np.uint16:   # torch has limited uint16 support: move the bytes
            a = a.view(np.int16)
        if a.nbytes >= _PIN_MIN_BYTES and np.dtype(a.dtype) in _TORCH_DTYPES:
            out = torch.empty(a.shape, dtype=_torch_dtype(a.dtype), device=self.device)
            self.upload(a, out)
            # (a bool array is a torch.bool tensor whatever its size: the small ones go through torch.from_numpy)
            return out.view(torch.bool) if a.dtype == np.bool_ else out
        if not a.flags.writeable:          # torch refuses to wrap read-only memory silently (e.g. np.load results)
            a = a.copy()
        return torch.from_numpy(a).to(self.device)

    def upload(self, array, out):
        """
        Contiguous host array -> the device tensor `out` (same bytes) on the current stream, through a page-locked staging
        buffer in pieces: the host copies piece k + 1 into the staging buffer while the DMA engine moves piece k, so the
        whole costs about one host memcpy (a pageable hipMemcpy stages through a small internal buffer at 5-8 GB/s).
        Returns when the last piece has been handed to the DMA engine; the device side is ordered on the current stream.
        """
        src = np.ascontiguousarray(array)
        assert out.is_contiguous() and out.numel() * out.element_size() == src.nbytes, 'upload: sizes differ'
        # amt_upload_staged: a few host threads copy every n-th piece through page-locked staging pieces of their own and
        # hand it to the DMA engine on the current stream (one thread's memcpy into page-locked memory is half the link's rate)
        Context.current(self.device)
        self.call('amt_upload_staged', C.c_void_p(out.data_ptr()), C.c_void_p(src.ctypes.data), src.nbytes)

    # -- per-kernel timing ----------------------------------------------------------------------
    def timing_enable(self, enable=True):
        """True / n: bracket every (n-th) launch of the dominant kernels with HIP events; False / 0: off."""
        self.call('amt_timing_enable', int(enable))

    def timing_read(self, kernel):
        """(total ms, launches) of kernel 0 (georef) / 1 (bin) since timing was enabled. Synchronises."""
        ms, n = C.c_double(), C.c_int()
        self.call('amt_timing_read', kernel, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def last_variant(self):
        """(second, bin, frames) of the latest launch of the frame kernel: amt_georef_last_variant."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.call('amt_georef_last_variant', C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    # -- events -------------------------------------------------------------------------------
    def event(self):
        e = C.c_void_p()
        self.call('amt_event_create', C.byref(e))
        return e

    def record(self, event):
        self.call('amt_event_record', event)

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        self.call('amt_event_elapsed_ms', start, stop, C.byref(ms))
        return ms.value

    def destroy_event(self, event):
        self.call('amt_event_destroy', event)


_TORCH_DTYPES = (np.dtype(np.float64), np.dtype(np.uint8), np.dtype(np.int16), np.dtype(np.uint16), np.dtype(np.int64),
                 np.dtype(np.bool_))


def _torch_dtype(dtype):
    import torch
    return {np.dtype(np.float64): torch.float64, np.dtype(np.uint8): torch.uint8,
            np.dtype(np.int16): torch.int16, np.dtype(np.uint16): torch.int16,
            np.dtype(np.int64): torch.int64, np.dtype(np.bool_): torch.uint8}[np.dtype(dtype)]


def ptr(tensor):
    """Device pointer of a torch tensor (or None)."""
    if tensor is None:
        return None
    assert tensor.is_contiguous()
    return C.c_void_p(tensor.data_ptr())


def host3(v):
    return (C.c_double * 3)(*[float(x) for x in np.asarray(v, dtype=np.float64).ravel()])


def host9(m):
    return (C.c_double * 9)(*[float(x) for x in np.asarray(m, dtype=np.float64).ravel()])


_PIN_MIN_BYTES = 1 << 20
# page-locked bytes that to_host() hands out as result arrays at any one time (beyond it: pageable arrays)
_PINNED_RESULT_LIMIT = int(float(os.environ.get('AMT_PINNED_RESULT_GB', '2')) * (1 << 30))
_pinned_out = {'bytes': 0}


def _pinned_release(nbytes):
    _pinned_out['bytes'] -= nbytes


class _NumpyOfTorch(dict):
    def __missing__(self, key):
        import torch
        self.update({torch.float64: np.float64, torch.uint8: np.uint8, torch.int16: np.int16, torch.int64: np.int64,
                     torch.int32: np.int32, torch.float32: np.float32, torch.bool: np.bool_, torch.int8: np.int8})
        return dict.__getitem__(self, key)


_NUMPY_OF_TORCH = _NumpyOfTorch()


def to_host(tensor, dtype=None, shape=None):
    """Device tensor -> numpy array (synchronises).  Large arrays travel into page-locked memory (the DMA engines'
    rate, ~50 GB/s, instead of the 5-8 GB/s of a copy into pageable memory); the array returned IS that memory —
    torch's caching host allocator takes the block back when the array is freed, so a loop that reads one array per
    frame pays the page-locking once."""
    import torch
    nbytes = tensor.numel() * tensor.element_size()
    if tensor.is_cuda and nbytes >= _PIN_MIN_BYTES:
        t = tensor if tensor.is_contiguous() else tensor.contiguous()
        if _pinned_out['bytes'] + nbytes <= _PINNED_RESULT_LIMIT:
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()
            a = host.numpy()
            #  page-locked memory handed out as results is bounded: a caller that keeps many mappings' arrays
            # alive gets ordinary pageable arrays beyond the limit instead of pinning tens of GB
            _pinned_out['bytes'] += nbytes
            weakref.finalize(a, _pinned_release, nbytes)     # (views of `a` keep it alive through their base)
        else:
            # pageable result through the library's staged copy (worker threads + page-locked pieces: the link's rate
            # once the pages exist; a fresh allocation also pays its page faults)
            ctx = Context.current(t.device)
            a = np.empty(tuple(t.shape), dtype=_NUMPY_OF_TORCH[t.dtype])
            ctx.call('amt_download_staged', C.c_void_p(a.ctypes.data), C.c_void_p(t.data_ptr()), nbytes)
    else:
        a = tensor.cpu().numpy()
    if dtype is not None and a.dtype != np.dtype(dtype):
        a = a.view(dtype) if a.dtype.itemsize == np.dtype(dtype).itemsize else a.astype(dtype)
    if shape is not None:
        a = a.reshape(shape)
    return a
