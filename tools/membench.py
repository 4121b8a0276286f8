"""Calibration: streaming read / copy rates on this GPU through torch (not part of the product)."""
import time
import torch
n = 96 * 1024 * 1024 // 8
a = [torch.randn(n, dtype=torch.float64, device='cuda') for _ in range(3)]
b = torch.empty_like(a[0])
big = torch.randn(480 * 1024 * 1024 // 8, dtype=torch.float64, device='cuda')
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
dt = t(lambda: b.copy_(a[0])); print('copy 96MB->96MB: %.1f us, %.2f TB/s (r+w)' % (dt * 1e6, 2 * n * 8 / dt / 1e12))
dt = t(lambda: [x.sum() for x in a]); print('sum 3x96MB: %.1f us, %.2f TB/s read' % (dt * 1e6, 3 * n * 8 / dt / 1e12))
dt = t(lambda: (big.fill_(1.0), [x.sum() for x in a])); print('fill 480MB + sum 3x96MB: %.1f us' % (dt * 1e6))
dt = t(lambda: big.fill_(1.0)); print('fill 480MB: %.1f us, %.2f TB/s write' % (dt * 1e6, big.numel() * 8 / dt / 1e12))
