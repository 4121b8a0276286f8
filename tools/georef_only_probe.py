"""BASELINE configs[1] on its own: the georeferencing kernel without fused binning, frames back to back (what bench.py's
`configs1_georef_only` variant times), per AMT_GEOREF_ROWS."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import sequence_frame
W, H = 4240, 2832
frames = [sequence_frame(k, W, H)[:3] for k in range(110)]
pipe = FramePipeline(W, H, alloc_image=False)
for rep in range(2):
    for hdr, cam, t in frames[:10]:
        pipe.georef(hdr, 110, cam, t, True, 10.0)
    pipe.ctx.timing_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for hdr, cam, t in frames[10:]:
        pipe.georef(hdr, 110, cam, t, True, 10.0)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    g, n = pipe.ctx.timing_read(0)
    pipe.ctx.timing_enable(False)
    print('rows', os.environ.get('AMT_GEOREF_ROWS', 'default'), 'ms/frame %.4f' % (el / 100 * 1e3), 'kernel ms %.4f' % (g / n), 'n', n)
