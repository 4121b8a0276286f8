"""The sequence loop with 8-bit images (what developed JPEG frames are) against 16-bit ones: frames per second and the
big kernel's time, 96 full-size frames each, every frame its own resident image."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import sequence_frame
W, H, N = 4240, 2832, 96
for dt, tdt, hi in ((np.uint16, torch.int16, 65535), (np.uint8, torch.uint8, 255)):
    imgs = [torch.randint(0, hi, (H, W, 3), device='cuda', dtype=torch.int32).to(tdt) for _ in range(32)]
    frames = []
    for k in range(N):
        hdr, cam, t, _ = sequence_frame(k, W, H)
        frames.append((hdr, cam, t, imgs[k % 32]))
    seq = SequencePipeline(W, H, img_dtype=dt, pxPerDeg=10, own_image_buffers=False)
    for rep in range(6):
        seq.process(frames)
    torch.cuda.synchronize()
    seq.ctx.timing_enable(1)
    t0 = time.perf_counter()
    out = seq.process(frames)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    g, n = seq.ctx.timing_read(0)
    seq.ctx.timing_enable(0)
    print('%-7s %.4f ms/frame = %6.0f Mpixel/s; big kernel %.4f ms per frame; plans %s' % (
        np.dtype(dt).name, dt_s / N * 1e3, W * H * N / dt_s / 1e6, g / n, sorted(set(seq.plans))), flush=True)
    del seq, out, imgs, frames
    torch.cuda.empty_cache()
