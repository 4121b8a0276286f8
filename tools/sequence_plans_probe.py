"""Which plan every frame of the synthetic sequence takes (bench.py gives rank r the frames r * (W + K) ...): frames 0 .. N-1
through one SequencePipeline, 40 at a time.  usage: python tools/sequence_plans_probe.py [N]"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import frame_image, sequence_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
W, H = 4240, 2832
seq = SequencePipeline(W, H, altitude=110, fast=True, min_elevation=10, pxPerDeg=10)
imgs = [torch.from_numpy(frame_image(W, H, seed=i).view(np.int16)).to(seq.ctx.device) for i in range(4)]
plans = []
shapes = collections.Counter()
for k0 in range(0, n, 40):
    frames = []
    for k in range(k0, min(n, k0 + 40)):
        hdr, cam, t, _ = sequence_frame(k, W, H)
        frames.append((hdr, cam, t, imgs[k % 4], None))
    res = seq.process(frames)
    plans.extend(seq.plans)
    for r in res:
        shapes[None if r is None else tuple(r['mean'].shape[:2])] += 1
    torch.cuda.synchronize()
c = collections.Counter(plans)
print('frames %d plans %s' % (n, dict(c)))
other = [i for i, p in enumerate(plans) if p != 'single-pass']
print('frames that are not single-pass:', other[:40])
print('grid shapes (ny, nx): smallest %s largest %s' % (min(s for s in shapes if s), max(s for s in shapes if s)))
