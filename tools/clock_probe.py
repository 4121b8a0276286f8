"""Shader clock and power while a kernel mix runs (rocm-smi sampled from a thread): the VALU-bound frame kernel back to back
against the same kernel alternating with the memory-bound binning pass (two-pass plan).  usage: clock_probe.py [seconds]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd.pipeline import FramePipeline, SequencePipeline
from auromat_amd.synthetic import sequence_frame
W, H = 4240, 2832
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
samples = []
stop = False


def sample():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
            sclk = [l.split('(')[-1].split('Mhz')[0] for l in out.splitlines() if 'sclk' in l]
            pw = [l.split(':')[-1].strip() for l in out.splitlines() if 'Power' in l and 'W' in l.split(':')[-1] or 'Socket Power' in l]
            samples.append((time.time(), sclk[:1], pw[:1]))
        except Exception as e:
            samples.append((time.time(), str(e), None))
        time.sleep(0.2)


fr = [sequence_frame(k, W, H) for k in range(8)]
img = torch.randint(0, 65535, (H, W, 3), device='cuda', dtype=torch.int32).to(torch.int16)
p = FramePipeline(W, H, alloc_image=False)
for name in ('georef kernel back to back', 'fused kernel (single-pass sequence)', 'two-pass sequence'):
    samples.clear()
    stop = False
    th = threading.Thread(target=sample)
    th.start()
    t0 = time.time()
    n = 0
    if name.startswith('georef'):
        p.ctx.timing_enable(1)
        while time.time() - t0 < secs:
            for hdr, cam, t, _ in fr:
                p.georef(hdr, 110, cam, t, True, 10.0)
            torch.cuda.synchronize()
            n += len(fr)
        tot, cnt = p.ctx.timing_read(0)
        p.ctx.timing_enable(False)
        kern = tot / max(cnt, 1) * 1e3
    else:
        seq = SequencePipeline(W, H, pxPerDeg=10, plan='single-pass' if name.startswith('fused') else 'two-pass', own_image_buffers=False,
                               img_dtype='uint16', keep_coordinates=True)
        frames = [(hdr, cam, t, img) for hdr, cam, t, _ in fr] * 6
        kern = float('nan')
        while time.time() - t0 < secs:
            seq.process(frames, keep_on_device=True)
            torch.cuda.synchronize()
            n += len(frames)
    el = time.time() - t0
    stop = True
    th.join()
    print('%-40s %6d frames in %.1f s = %.1f us per frame (kernel class 0: %.1f us); sclk / power samples: %s' % (
        name, n, el, el / n * 1e6, kern, ' '.join('%s/%s' % (s[1][0] if s[1] else '?', s[2][0] if s[2] else '?') for s in samples[::2])), flush=True)
    time.sleep(1.0)
