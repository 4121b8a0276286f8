"""Host-side cost of one SequencePipeline.process() call of 20 resident frames, piece by piece (which of the ~150 us outside
the GPU's span are Python): usage: python tools/host_overhead_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import frame_image, sequence_frame

W, H = 4240, 2832
seq = SequencePipeline(W, H, altitude=110, fast=True, min_elevation=10, pxPerDeg=10)
imgs = [torch.from_numpy(frame_image(W, H, seed=i).view(np.int16)).to(seq.ctx.device) for i in range(4)]
frames = []
for k in range(25):
    hdr, cam, t, _ = sequence_frame(k, W, H)
    frames.append((hdr, cam, t, imgs[k % 4], None))
for _ in range(3):
    seq.process(frames[:5]); seq.process(frames[5:])
torch.cuda.synchronize()
timed = frames[5:]
for rep in range(5):
    seq.process(frames[:5]); torch.cuda.synchronize()
    t0 = time.perf_counter(); ok = seq._native_applies(timed); t1 = time.perf_counter()
    r = seq.process(timed); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print('_native_applies %.1f us (%s); process() %.1f us; sync after %.1f us' % ((t1 - t0) * 1e6, ok, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
import cProfile, pstats
seq.process(frames[:5]); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); seq.process(timed); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
