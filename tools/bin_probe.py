"""k_bin_frame alone (the binning pass of the two-pass plan: generic mappings, caller masks, non-RGB images): GPU time per
frame at full size, launches enqueued behind a sleeping kernel, arrays of a real frame.  AMT_LIB_PATH selects an A/B build
(tools/build_variant.sh)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from auromat_amd._native import Context
from auromat_amd.pipeline import FramePipeline
from auromat_amd.synthetic import sequence_frame
W, H = 4240, 2832
ctx = Context.current()
hdr, cam, t, _ = sequence_frame(0, W, H)
img = torch.randint(0, 65535, (H, W, 3), device='cuda', dtype=torch.int32).to(torch.int16)
pipe = FramePipeline(W, H, alloc_image=False)
pipe.use_image(img)
for ppd in (10.0, 25.0, 4.0):
    ref = pipe.run(hdr, 110, cam, t, fast=True, min_elevation=10, pxPerDeg=ppd, fuse=False)
    grid = ref['grid']
    xaxis, yaxis = grid.axes(ctx)
    fd = pipe.fd
    acc = torch.zeros(5 * grid.nx * grid.ny, dtype=torch.int64, device='cuda')

    def frame():
        ctx.call('amt_bin_frame', C.c_void_p(fd.lat_c.data_ptr()), C.c_void_p(fd.lon_c.data_ptr()), C.c_void_p(fd.elev.data_ptr()),
                 C.c_void_p(img.data_ptr()), 2, 3, None, H, W, 10.0, C.byref(xaxis), C.byref(yaxis), 0, C.c_void_p(acc.data_ptr()))
    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    acc.zero_()
    frame()
    torch.cuda.synchronize()
    cnt = int(acc[:grid.nx * grid.ny].sum())
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(1.0e8))
    e0.record()
    for _ in range(n):
        frame()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print('%s: %g px/deg, grid %d x %d, pixels binned %d: k_bin_frame %.1f us per frame = %.2f TB/s on 360.2 MB (%.2f of 8 TB/s)' % (
        os.environ.get('AMT_LIB_PATH', 'default'), ppd, grid.nx, grid.ny, cnt, us, 360.2e6 / us / 1e6, 360.2e6 / us / 1e6 / 8), flush=True)
