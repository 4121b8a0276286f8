"""The store pattern of the row kernel on its own: a frame that looks away from the Earth (every ray misses, so the
kernel does next to no arithmetic and writes 480 MB of NaN in its usual pattern), against torch.fill_ of the same
bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
pipe = FramePipeline(W, H)
pipe.set_image(frame_image(W, H))
hdr, cam, t, _ = sequence_frame(0, W, H)
sky = dict(hdr, CRVAL1=(hdr['CRVAL1'] + 180.0) % 360, CRVAL2=-hdr['CRVAL2'])
for name, h in (('earth frame', hdr), ('sky frame', sky)):
    for k in range(3):
        pipe.georef(h, 110, cam, t)
    torch.cuda.synchronize()
    pipe.ctx.timing_enable(1)
    for k in range(20):
        pipe.georef(h, 110, cam, t)
        torch.cuda.synchronize()
    g, n = pipe.ctx.timing_read(0)
    print('%-12s georef-only kernel %.4f ms = %.2f TB/s written' % (name, g / n, 480.4e6 / (g / n * 1e-3) / 1e12))
    pipe.ctx.timing_enable(0)
bufs = [torch.empty(12014753, dtype=torch.float64, device='cuda') for _ in range(5)]
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        for b in bufs:
            b.fill_(1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
print('torch fill_ of five arrays %.4f ms = %.2f TB/s' % (dt * 1e3, 5 * 12014753 * 8 / dt / 1e12))
