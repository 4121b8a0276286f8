"""Host-side timeline of a 20-frame SequencePipeline.process() call through the native runner: entry, amt_run_begin, every
amt_run_push, amt_run_end, return — where the time of a short call goes before the first kernel and after the last."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import sequence_frame
W, H = 4240, 2832
imgs = [torch.randint(0, 65535, (H, W, 3), device='cuda', dtype=torch.int32).to(torch.int16) for _ in range(8)]
frames = [sequence_frame(k, W, H)[:3] + (imgs[k % 8],) for k in range(25)]
seq = SequencePipeline(W, H)
lib = seq.ctx._lib
marks = []
for name in ('amt_run_begin', 'amt_run_push', 'amt_run_end'):
    fn = getattr(lib, name)
    def wrap(fn=fn, name=name):
        def call(*a):
            t0 = time.perf_counter(); rc = fn(*a); marks.append((name, t0, time.perf_counter())); return rc
        return call
    setattr(lib, name, wrap())
for rep in range(6):
    seq.process(frames[:5]); torch.cuda.synchronize()
    del marks[:]
    seq.ctx.timing_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = seq.process(frames[5:25])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ms, n = seq.ctx.timing_read(0); seq.ctx.timing_enable(False)
    print('process %.3f ms (+%.3f sync), kernels %.3f ms for %d frames; begin at +%.0f us (took %.0f), first push +%.0f us (took %.0f), pushes total %.0f us, last push returned +%.0f us, end +%.0f us (took %.0f us), return +%.0f us'
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, ms, n, (marks[0][1] - t0) * 1e6, (marks[0][2] - marks[0][1]) * 1e6, (marks[1][1] - t0) * 1e6,
             (marks[1][2] - marks[1][1]) * 1e6, sum(m[2] - m[1] for m in marks if m[0] == 'amt_run_push') * 1e6, (marks[-2][2] - t0) * 1e6,
             (marks[-1][1] - t0) * 1e6, (marks[-1][2] - marks[-1][1]) * 1e6, (t1 - t0) * 1e6))
    if rep == 5:
        print(' '.join('%s:%.0f+%.0f' % (m[0][8:], (m[1] - t0) * 1e6, (m[2] - m[1]) * 1e6) for m in marks))
