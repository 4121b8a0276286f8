"""Where the wall time of SequencePipeline.process() goes with the frame loop in the library (amt_run_*) and in Python:
whole call + synchronisation for 24 / 96 / 192 resident frames, and the host-side phases of the native call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
n_img = 16
imgs = [torch.from_numpy(frame_image(W, H, seed=k).view(np.int16)).cuda() for k in range(n_img)]
def frames(n, k0=0):
    out = []
    for k in range(k0, k0 + n):
        hdr, cam, t, _ = sequence_frame(k, W, H)
        out.append((hdr, cam, t, imgs[k % n_img]))
    return out
for native in (True, False):
    seq = SequencePipeline(W, H, pxPerDeg=10, own_image_buffers=False)
    seq.native = native
    seq.process(frames(40))
    torch.cuda.synchronize()
    for n in (24, 96, 192):
        best = None
        for rep in range(3):
            fr = frames(n, 40)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = seq.process(fr)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            seq.ctx.timing_enable(0)
            cur = (t2 - t0, t1 - t0)
            best = cur if best is None or cur[0] < best[0] else best
        print('native' if native else 'python', 'frames %3d: call + sync %.3f ms = %.4f ms per frame (the call returned after %.3f ms)' % (
            n, best[0] * 1e3, best[0] * 1e3 / n, best[1] * 1e3), seq.plans.count('single-pass'), seq.hinted)
