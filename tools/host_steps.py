"""Host-side time per phase of SequencePipeline.process (monkey-patched timers), batch given on the command line."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import SequencePipeline, FramePipeline
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
keep = not (len(sys.argv) > 2 and sys.argv[2] == 'grids')        # 'grids': no per-pixel coordinate arrays
seq = SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H), batch=batch, keep_coordinates=keep)
frames = [sequence_frame(k, W, H)[:3] + (None,) for k in range(70)]
acc = {}
def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name
    def w(*a, **kw):
        t = time.perf_counter(); r = fn(*a, **kw); acc.setdefault(label, []).append(time.perf_counter() - t); return r
    setattr(obj, name, w)
timed(seq, '_prepare'); timed(seq, '_launch'); timed(seq, '_finish')
for q in seq.pipes:
    timed(q, '_wait_fused', 'wait'); timed(q, '_finalize_fused', 'finalize')
seq.process(frames[:10])
for v in acc.values(): del v[:]
torch.cuda.synchronize(); t0 = time.perf_counter()
seq.process(frames[10:])
torch.cuda.synchronize(); el = time.perf_counter() - t0
print('batch', batch, 'with arrays' if keep else 'grids only', 'us/frame', round(el / 60 * 1e6, 1))
for k, v in acc.items():
    print('  %-10s n=%3d mean %.1f us  total/frame %.1f us' % (k, len(v), sum(v) / len(v) * 1e6, sum(v) / 60 * 1e6))
