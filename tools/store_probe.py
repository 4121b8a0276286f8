"""How much of the fused kernel's time is its stores?  The output arrays are optional (NULL = not written): time the
kernel alone with all nine, without the corner arrays, and with none (pure ray casting + binning)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
pipe = FramePipeline(W, H)
pipe.set_image(frame_image(W, H))
out = pipe._out
full = {k: getattr(out, k) for k in ('lat', 'lon', 'lat_c', 'lon_c', 'elev')}
for name, keep in (('all five arrays', ('lat', 'lon', 'lat_c', 'lon_c', 'elev')), ('centres only', ('lat_c', 'lon_c', 'elev')),
                   ('corners only', ('lat', 'lon')), ('elev only', ('elev',)), ('no stores', ())):
    for k in full:
        setattr(out, k, full[k] if k in keep else None)
    for fuse in (True,):
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
        print('%-18s fused kernel %.4f ms' % (name, g / n))
        pipe.ctx.timing_enable(0)
