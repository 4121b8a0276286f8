"""Kernel time of the directions-in variant (amt_georef_frame_dirs, SURVEY 8d "directions-in" row: 768.8 MB)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd._native import Context, GeorefOut, ptr
from auromat_amd.coordinates.wcs import pix2world
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import frame_header
W, H = 4240, 2832
hdr, cam, t = frame_header(W, H)
ctx = Context.current()
dirs = pix2world(hdr, W, H, corner=True, ascartesian=True, device=ctx.device)
p = frame_params(hdr, 110, cam, t, True, magnetic=False)
out = GeorefOut()
bufs = [ctx.empty((H + 1, W + 1)), ctx.empty((H + 1, W + 1)), ctx.empty((H, W)), ctx.empty((H, W)), ctx.empty((H, W)), ctx.empty((8,))]
out.lat, out.lon, out.lat_c, out.lon_c, out.elev, out.bbox = (b.data_ptr() for b in bufs)
out.bbox_min_elevation = 10.0
for mode in ('dirs', 'wcs'):
    for rep in range(2):
        ctx.timing_enable(1 if rep else 0)
        for k in range(12):
            if mode == 'dirs':
                ctx.call('amt_georef_frame_dirs', C.byref(p), ptr(dirs), C.byref(out))
            else:
                ctx.call('amt_georef_frame', C.byref(p), C.byref(out))
            torch.cuda.synchronize()
    g, n = ctx.timing_read(0)
    nc, npx = (W + 1) * (H + 1), W * H
    nbytes = (40 * nc + 24 * npx) if mode == 'dirs' else (16 * nc + 24 * npx)
    print('%s: %.1f us per frame, %.1f MB -> %.2f TB/s = %.3f of 8 TB/s' % (mode, g / n * 1e3, nbytes / 1e6, nbytes / (g / n * 1e-3) / 1e12, nbytes / (g / n * 1e-3) / 8e12))
