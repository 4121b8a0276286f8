"""amt_upload_staged / amt_download_staged: GB/s for the sizes of a frame's image and of one per-pixel array."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from auromat_amd._native import Context
ctx = Context.current()
for mb in (36, 72, 96):
    n = mb * 1000 * 1000
    src = np.random.RandomState(1).randint(0, 255, n).astype(np.uint8)
    dst = np.empty(n, dtype=np.uint8); dst[:] = 0
    dev = torch.empty(n, dtype=torch.uint8, device='cuda')
    up, down = [], []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.call('amt_upload_staged', C.c_void_p(dev.data_ptr()), C.c_void_p(src.ctypes.data), n)
        torch.cuda.synchronize(); up.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        ctx.call('amt_download_staged', C.c_void_p(dst.ctypes.data), C.c_void_p(dev.data_ptr()), n)
        down.append(time.perf_counter() - t0)
    assert np.array_equal(src, dst)
    print('%3d MB  threads %s: upload %.2f ms = %.1f GB/s, download %.2f ms = %.1f GB/s' % (
        mb, os.environ.get('AMT_COPY_THREADS', 'default'), min(up) * 1e3, n / 1e9 / min(up), min(down) * 1e3, n / 1e9 / min(down)), flush=True)
