"""Georef-only kernel time for every item order (same process, interleaved), full-size frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
pipe = FramePipeline(W, H)
pipe.set_image(frame_image(W, H))
tot = {o: 0.0 for o in (0, 1, 2)}
reps = 6
for rep in range(reps + 1):
    for order in tot:
        pipe._out.item_order = order
        pipe.ctx.timing_enable(1)
        for k in range(4):
            hdr, cam, t, _ = sequence_frame(k, W, H)
            pipe.georef(hdr, 110, cam, t, True, 10.0)
        torch.cuda.synchronize()
        g, n = pipe.ctx.timing_read(0)
        pipe.ctx.timing_enable(0)
        if rep:
            tot[order] += g / n
print({o: round(v / reps, 4) for o, v in tot.items()})
