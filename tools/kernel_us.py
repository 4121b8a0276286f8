"""Kernel microseconds per frame of the bench sequence (192 frames after spin-up) for the current environment (A/B switches of
the library are read from the environment once per process).  usage: [ENV=...] kernel_us.py [magnetic] [keep=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import sequence_frame
W, H = 4240, 2832
mag = 'magnetic' in sys.argv
keep = 'keep=0' not in sys.argv
imgs = [torch.randint(0, 65535, (H, W, 3), device='cuda', dtype=torch.int32).to(torch.int16) for _ in range(16)]
fr = [sequence_frame(k, W, H)[:3] + (imgs[k % 16], (100, 110, 120)[k % 3] if mag else None) for k in range(201)]
seq = SequencePipeline(W, H, magnetic=mag, keep_coordinates=keep)
t_end = time.perf_counter() + 0.4
while time.perf_counter() < t_end:
    seq.process(fr[:9]); torch.cuda.synchronize()
seq.ctx.timing_enable(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
r = seq.process(fr[9:]); torch.cuda.synchronize(); el = time.perf_counter() - t0
ms, n = seq.ctx.timing_read(0)
tags = ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('AMT_'))
print('%-60s kernel %.1f us per frame, %.4f ms per frame, plans %s' % (tags or '(default)', ms / n * 1e3, el / 192 * 1e3, sorted(set(seq.plans))), flush=True)
