"""Frames with a pole in view at full size (4256 x 2832): the single-pass plan (pole plan of the fused kernel) against
the two-pass plan (georef kernel, whole-array rotation, box reduction, binning kernel), frames/s through
SequencePipeline and the big kernel's own time; an ordinary frame of the same size beside them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import sequence_frame, frame_image

W, H = 4256, 2832
from auromat_amd.synthetic import pole_frame
pole_hdr, pole_cam, t = pole_frame(W, H)
img = torch.from_numpy(frame_image(W, H).view(np.int16)).cuda()
hdr, cam, t0, _ = sequence_frame(0, W, H)
N = 24
for name, frame in (('pole frame', (pole_hdr, pole_cam, t, img)), ('ordinary frame', (hdr, cam, t0, img))):
    for plan in ('single-pass', 'two-pass'):
        for keep in (True, False):
            if plan == 'two-pass' and not keep:
                continue
            seq = SequencePipeline(W, H, pxPerDeg=10, plan=plan, own_image_buffers=False, keep_coordinates=keep)
            seq.process([frame] * 6)
            torch.cuda.synchronize()
            seq.ctx.timing_enable(1)
            t1 = time.perf_counter()
            out = seq.process([frame] * N)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / N
            g, n = seq.ctx.timing_read(0)
            seq.ctx.timing_enable(0)
            print('%-15s %-11s %-16s %.3f ms/frame = %6.0f Mpixel/s; big kernel %.4f ms; plans %s; grid %s pole %s' % (
                name, plan, 'with arrays' if keep else 'grids only', dt * 1e3, W * H / dt / 1e6, g / max(n, 1),
                sorted(set(seq.plans)), tuple(out[0]['mean'].shape[:2]), out[0]['contains_pole']), flush=True)
            del seq, out
            torch.cuda.empty_cache()
