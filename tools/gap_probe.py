"""Why does k_georef_rows<true,false,0,0> take 115-121 us per frame in a loop of its own and 94-97 us inside the two-pass plan
(same frames, same outputs)?  The kernel's HIP-event time under: one or several sets of output arrays, idle gaps, a read-only
kernel in between.  usage: gap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import sequence_frame
W, H = 4240, 2832
fr = [sequence_frame(k, W, H) for k in range(8)]
pipes = [FramePipeline(W, H, alloc_image=False) for _ in range(4)]
big = torch.empty(64 * 1024 * 1024, dtype=torch.float64, device='cuda')      # 512 MB
modes = [('one set of arrays, back to back', 1, None), ('two sets of arrays, alternating', 2, None), ('four sets', 4, None),
         ('one set + a 256-MB read (sum) in between', 1, 32 * 1024 * 1024), ('two sets + a 256-MB read in between', 2, 32 * 1024 * 1024),
         ('one set + a 64-MB read in between', 1, 8 * 1024 * 1024), ('one set, back to back', 1, None)]
for name, nsets, read in modes:
    ctx = pipes[0].ctx
    for k, (hdr, cam, t, _) in enumerate(fr):
        pipes[k % nsets].georef(hdr, 110, cam, t, True, 10.0)
    torch.cuda.synchronize()
    ctx.timing_enable(1)
    t0 = time.time(); n = 0
    while time.time() - t0 < 1.5:
        for k, (hdr, cam, t, _) in enumerate(fr):
            pipes[k % nsets].georef(hdr, 110, cam, t, True, 10.0)
            if read:
                big[:read].sum()
        n += 8
    torch.cuda.synchronize()
    el = time.time() - t0
    tot, cnt = ctx.timing_read(0)
    ctx.timing_enable(False)
    print('%-44s wall %.1f us per frame, kernel (HIP events) %.1f us over %d launches' % (name, el / n * 1e6, tot / cnt * 1e3, cnt), flush=True)
