"""Fused kernel time per frame for launches of 1 / 2 / 3 frames with nothing else on the GPU (each process() call is
followed by a synchronize), against the same launches inside the running pipeline (bench.py): what the side kernels
of neighbouring batches cost the big kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
for batch in (1, 2, 3):
    seq = SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H), batch=batch)
    frames = [sequence_frame(k, W, H)[:3] + (None,) for k in range(12 * batch)]
    seq.process(frames[:batch])
    torch.cuda.synchronize()
    for mode in ('isolated', 'pipelined'):
        seq.ctx.timing_enable(1)
        if mode == 'isolated':
            for i in range(batch, len(frames), batch):
                seq.process(frames[i:i + batch])
                torch.cuda.synchronize()
        else:
            seq.process(frames[batch:])
            torch.cuda.synchronize()
        g, n = seq.ctx.timing_read(0)
        seq.ctx.timing_enable(0)
        print('batch %d %-9s fused kernel %.1f us per frame (%d frames)' % (batch, mode, g / n * 1e3, n))
