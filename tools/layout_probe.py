"""Round 6: the frame kernel ALONE (no binning; amt_georef_frame / amt_georef_frame_dirs on the bench frame) with contiguous and with
strip-padded output rows, same process, passes interleaved; launches back to back (no host synchronisation inside a pass).
usage: [AMT_ITEM_ORDER=..] layout_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd._native import Context, GeorefOut, ptr
from auromat_amd.coordinates.wcs import pix2world
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import sequence_frame
W, H = 4240, 2832
ctx = Context.current()
P = int(ctx._lib.amt_padded_pitch(W))
hdr, cam, t, _ = sequence_frame(0, W, H)
dirs = pix2world(hdr, W, H, corner=True, ascartesian=True, device=ctx.device)
p = frame_params(hdr, 110, cam, t, True, magnetic=False)
outs = {}
keep = []
for name, shape_c, shape_p, layout in (('contiguous', (H + 1, W + 1), (H, W), 0), ('padded', (H + 1, P), (H, P), 1)):
    o = GeorefOut()
    bufs = [ctx.empty(shape_c), ctx.empty(shape_c), ctx.empty(shape_p), ctx.empty(shape_p), ctx.empty(shape_p), ctx.empty((8,))]
    keep.append(bufs)
    o.lat, o.lon, o.lat_c, o.lon_c, o.elev, o.bbox = (b.data_ptr() for b in bufs)
    o.bbox_min_elevation = 10.0
    o.row_layout = layout
    outs[name] = o
N = 24
res = {}
for mode in ('wcs', 'dirs'):
    for rep in range(6):
        for name, o in outs.items():
            ctx.timing_enable(1)
            for k in range(N):
                if mode == 'dirs':
                    ctx.call('amt_georef_frame_dirs', C.byref(p), ptr(dirs), C.byref(o))
                else:
                    ctx.call('amt_georef_frame', C.byref(p), C.byref(o))
            torch.cuda.synchronize()
            g, n = ctx.timing_read(0)
            ctx.timing_enable(False)
            if rep:
                res.setdefault((mode, name), []).append(g / n * 1e3)
tags = ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('AMT_'))
for (mode, name), v in res.items():
    print('%-28s %-5s %-10s median %.1f us per frame (%s)' % (tags or '(default)', mode, name, float(np.median(v)), ' '.join('%.1f' % x for x in v)), flush=True)
