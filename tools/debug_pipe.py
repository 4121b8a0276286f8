"""Coarse vs exact bounding box and the single-pass decision for a few frames (run on the GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from auromat_amd.pipeline import FramePipeline
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import frame_header, frame_image
from auromat_amd._native import ptr, to_host
for (w, h, pointing) in [(256, 170, 'iss030'), (253, 171, 'iss029'), (4240, 2832, 'iss029'), (1060, 708, 'iss029')]:
    hdr, cam, t = frame_header(w, h, pointing)
    pipe = FramePipeline(w, h)
    pipe.set_image(frame_image(w, h, seed=1))
    p = frame_params(hdr, 110, cam, t, True)
    for stride in (16, 4, 1):
        red = pipe.ctx.empty((8,))
        pipe.ctx.call('amt_georef_coarse_bbox', C.byref(p), stride, 9.5, 0, ptr(red))
        print(w, h, pointing, 'coarse stride', stride, np.round(to_host(red), 3))
    res = pipe.run(hdr, 110, cam, t, min_elevation=10, pxPerDeg=10, fuse=True, keep_on_device=True)
    r = pipe._fused['result']
    print('   exact', np.round(np.array(r.bbox[:]), 3), 'status', r.status, 'fused', r.fused, pipe.last_plan,
          'grid', r.grid.nx, r.grid.ny)
