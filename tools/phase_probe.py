"""Do the sky part and the Earth part of a frame overlap inside the row kernel?  Kernel time (alone, one frame per
launch) of the fused and the georef-only kernel on (a) the bench frame (43 % of its rows are sky), (b) a frame of sky
(zenith: stores only), (c) a frame of Earth (nadir: every ray hits, high elevation).  If the bench frame takes what its
sky rows and its Earth rows take one after the other, the two phases do not overlap.
AMT_ITEM_ORDER=3 (interleaved chunk order) for the A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd.pipeline import FramePipeline, EmptyFrame
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
pipe = FramePipeline(W, H)
pipe.set_image(frame_image(W, H))
hdr, cam, t, _ = sequence_frame(0, W, H)
up = cam / np.linalg.norm(cam)
def look(v):
    return dict(hdr, CRVAL1=float(np.rad2deg(np.arctan2(v[1], v[0])) % 360), CRVAL2=float(np.rad2deg(np.arcsin(v[2]))))
cases = (('bench frame', hdr), ('sky (zenith)', look(up)), ('earth (nadir)', look(-up)))
for name, h in cases:
    for fused in (False, True):
        def once():
            if fused:
                try:
                    pipe.run(h, 110, cam, t, pxPerDeg=10, fuse=True, keep_on_device=True)
                except EmptyFrame:
                    pass
            else:
                pipe.georef(h, 110, cam, t)
        for k in range(3):
            once()
        torch.cuda.synchronize()
        pipe.ctx.timing_enable(1)
        for k in range(15):
            once()
            torch.cuda.synchronize()
        g, n = pipe.ctx.timing_read(0)
        pipe.ctx.timing_enable(0)
        hits = int((~torch.isnan(pipe.fd.lat[:, W // 2])).sum()) if pipe.fd.lat is not None else -1
        print('%-14s %-12s kernel %.4f ms  plan %s  hit rows (middle column) %d of %d  order %s' % (
            name, 'fused' if fused else 'georef-only', g / n, pipe.last_plan if fused else '-', hits, H + 1,
            os.environ.get('AMT_ITEM_ORDER', 'default')))
