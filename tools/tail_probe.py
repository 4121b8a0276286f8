"""Host time of the parts of a 20-frame SequencePipeline.process call: until the first launch, the loop, the closing joins
and record_stream calls (wall clock around each part, GPU running asynchronously)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd import pipeline as P
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
seq = P.SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H))
frames = [sequence_frame(k, W, H)[:3] + (None,) for k in range(400)]
marks = {}
def stamp(name):
    marks.setdefault(name, []).append(time.perf_counter())
def wrap(obj, name, before=None, after=None):
    fn = getattr(obj, name)
    def w(*a, **kw):
        if before: stamp(before)
        r = fn(*a, **kw)
        if after: stamp(after)
        return r
    setattr(obj, name, w)
wrap(seq, '_launch', 'launch_in', 'launch_out')
wrap(seq, '_finish_batch', 'finish_in', 'finish_out')
n_rec = [0]
orig_rec = torch.Tensor.record_stream
def rec(self, s):
    n_rec[0] += 1
    return orig_rec(self, s)
torch.Tensor.record_stream = rec
seq.process(frames[:200])
torch.cuda.synchronize()
for rep in range(3):
    marks.clear(); n_rec[0] = 0
    k0 = 200 + rep * 20
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = seq.process(frames[k0:k0 + 20])
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    us = lambda t: round((t - t0) * 1e6, 1)
    print('first launch in/out', us(marks['launch_in'][0]), us(marks['launch_out'][0]), '| last finish in/out', us(marks['finish_in'][-1]),
          us(marks['finish_out'][-1]), '| process returns', us(t1), '| GPU idle', us(t2), '| record_stream calls', n_rec[0])
    print('   finish calls (us):', [round((b - a) * 1e6) for a, b in zip(marks['finish_in'], marks['finish_out'])])
    print('   launch calls (us):', [round((b - a) * 1e6) for a, b in zip(marks['launch_in'], marks['launch_out'])])
