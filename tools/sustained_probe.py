"""Georef-only kernel time when launched back to back (sustained clocks, nothing concurrent) vs one at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import FramePipeline
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
pipe = FramePipeline(W, H)
pipe.set_image(frame_image(W, H))
ps = []
for k in range(60):
    hdr, cam, t, _ = sequence_frame(k, W, H)
    ps.append((frame_params(hdr, 110, cam, t, True, magnetic=False), cam, t))
for mode in ('one at a time', 'back to back', 'one at a time', 'back to back'):
    for p, cam, t in ps[:5]:
        pipe.georef(None, 110, cam, t, params=p)
    torch.cuda.synchronize()
    pipe.ctx.timing_enable(1)
    for p, cam, t in ps:
        pipe.georef(None, 110, cam, t, params=p)
        if mode == 'one at a time':
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    g, n = pipe.ctx.timing_read(0)
    pipe.ctx.timing_enable(0)
    print('%-14s georef-only kernel %.1f us' % (mode, g / n * 1e3))
