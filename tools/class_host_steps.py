"""Where the wall time of the class route's resample() goes (array input, the reference's test frame size): the image upload,
the launch + wait + finalise, the result to the host; and one per-pixel array to the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from auromat_amd._native import Context, to_host
from auromat_amd.pipeline import fused_class_pipeline
from auromat_amd.synthetic import frame_header, frame_image
W, H = 4256, 2832
hdr, cam, t = frame_header(W, H, 'iss030')
for dtype in (np.uint8, np.uint16):
    img = frame_image(W, H, seed=1, dtype=dtype)
    pipe = fused_class_pipeline(W, H, dtype)
    ctx = pipe.ctx
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe.set_image(img)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res = pipe.run(hdr, 110, cam, t, fast=True, min_elevation=10, pxPerDeg=10, fuse=True)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print('%s image %.0f MB: upload call %.2f ms (+%.2f ms until it is on the device = %.1f GB/s), run without upload %.2f ms'
          % (np.dtype(dtype).name, img.nbytes / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, img.nbytes / 1e9 / (t2 - t0), (t3 - t2) * 1e3))
x = torch.empty((H, W), dtype=torch.float64, device='cuda').fill_(1.5)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = to_host(x)
    t1 = time.perf_counter()
    print('to_host of one (H, W) float64 array (%.0f MB): %.2f ms = %.1f GB/s' % (x.numel() * 8 / 1e6, (t1 - t0) * 1e3, x.numel() * 8 / 1e9 / (t1 - t0)), type(a).__name__)
    del a
import auromat_amd._native as N
N._PINNED_RESULT_LIMIT = 0
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = to_host(x)
    t1 = time.perf_counter()
    print('to_host into a fresh PAGEABLE array (amt_download_staged; beyond the pinned-result limit): %.2f ms = %.1f GB/s'
          % ((t1 - t0) * 1e3, x.numel() * 8 / 1e9 / (t1 - t0)))
buf = np.empty((H, W))
import ctypes as C
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.call('amt_download_staged', C.c_void_p(buf.ctypes.data), C.c_void_p(x.data_ptr()), buf.nbytes)
    t1 = time.perf_counter()
    print('amt_download_staged into an array whose pages exist: %.2f ms = %.1f GB/s' % ((t1 - t0) * 1e3, buf.nbytes / 1e9 / (t1 - t0)))
