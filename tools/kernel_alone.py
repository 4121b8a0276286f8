"""Duration of the fused kernel when nothing else runs beside it (frames processed strictly one after another)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import FramePipeline
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
pipe = FramePipeline(W, H)
pipe.set_image(frame_image(W, H))
for fuse in (True, False):
    for k in range(3):
        hdr, cam, t, _ = sequence_frame(k, W, H)
        pipe.run(hdr, 110, cam, t, pxPerDeg=10, fuse=fuse, keep_on_device=True)
    torch.cuda.synchronize()
    pipe.ctx.timing_enable(1)
    for k in range(3, 23):
        hdr, cam, t, _ = sequence_frame(k, W, H)
        pipe.run(hdr, 110, cam, t, pxPerDeg=10, fuse=fuse, keep_on_device=True)
        torch.cuda.synchronize()
    g, n = pipe.ctx.timing_read(0)
    b, nb = pipe.ctx.timing_read(1)
    print('fuse', fuse, 'georef kernel ms', g / n, 'bin kernel ms', b / nb if nb else None)
    pipe.ctx.timing_enable(0)
