"""Probe (round 6): does a background fill on a second stream overlap with the row kernel?  Needs the variant build
`libauromat_hip_nosky.so` (tools/bg_fill_probe.hip): the row kernel's sky items return at once (arrays WRONG: timing only), and per
launch of three frames a throttled fill of the same number of bytes (3 x 212 MB) goes to a scratch buffer on another stream, launched
right before the big kernel.  usage: AMT_LIB_PATH=.../libauromat_hip_nosky.so AMT_SEQ_NATIVE=0 bg_fill_probe.py [waves] [sleep]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auromat_amd import _native
from auromat_amd.pipeline import FramePipeline, SequencePipeline
from auromat_amd.synthetic import sequence_frame
W, H = 4240, 2832
waves = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sleep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = _native.lib()
have = hasattr(lib, 'amt_probe_background_fill')
imgs = [torch.randint(0, 65535, (H, W, 3), device='cuda', dtype=torch.int32).to(torch.int16) for _ in range(16)]
fr = [sequence_frame(k, W, H)[:3] + (imgs[k % 16], None) for k in range(201)]
seq = SequencePipeline(W, H)
sky_lines = int(0.43 * 2832) * 4352 // 16 * 5                      # per frame, five arrays
scratch = torch.empty(3 * sky_lines * 16, dtype=torch.float64, device='cuda')
sB = torch.cuda.Stream()
orig = FramePipeline.georef_many
mode = {'fill': False}


def patched(pipes, *a, **kw):
    if mode['fill'] and have:
        sB.wait_stream(torch.cuda.current_stream())                # (buffers of two batches ago are free)
        lib.amt_probe_background_fill(C.c_void_p(sB.cuda_stream), C.c_void_p(scratch.data_ptr()), C.c_longlong(len(pipes) * sky_lines),
                                      C.c_int(waves), C.c_int(sleep))
    return orig(pipes, *a, **kw)


FramePipeline.georef_many = staticmethod(patched)
for label, fill in (('row kernel alone (no sky fill at all)', False), ('+ background fill, %d waves, sleep %d' % (waves, sleep), True),
                    ('row kernel alone (no sky fill at all)', False), ('+ background fill, %d waves, sleep %d' % (waves, sleep), True)):
    mode['fill'] = fill
    for _ in range(3):
        seq.process(fr[:9]); torch.cuda.synchronize()
    seq.ctx.timing_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = seq.process(fr[9:]); torch.cuda.synchronize(); el = time.perf_counter() - t0
    ms, n = seq.ctx.timing_read(0)
    print('%-50s kernel %.1f us per frame, %.4f ms per frame in all' % (label, ms / n * 1e3, el / 192 * 1e3), flush=True)
