"""Does something done right before a short sequence slow its kernels down?  20 frames (the driver's step count) straight
after: nothing, a packing of earlier results (torch.cat), a small all_reduce, the full gather of earlier results."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from auromat_amd.pipeline import SequencePipeline
from auromat_amd import sequence as S
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import sequence_frame, frame_image
W, H, N = 4240, 2832, 20
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29555')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
seq = SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H))
frames = []
for k in range(N + 5):
    hdr, cam, t, _ = sequence_frame(k, W, H)
    frames.append((frame_params(hdr, 110, cam, t, True), cam, t, None))
dev = seq.ctx.device
for rep in range(40):
    seq.process(frames)
torch.cuda.synchronize()
x = torch.ones(1024, device='cuda')


def timed(label, before):
    out = []
    for rep in range(4):
        warm = seq.process(frames[:5])
        before(warm)
        torch.cuda.synchronize()
        seq.ctx.timing_enable(1)
        t0 = time.perf_counter()
        seq.process(frames[5:])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g, n = seq.ctx.timing_read(0)
        seq.ctx.timing_enable(0)
        out.append('%.4f (kernel %.4f)' % (dt / N * 1e3, g / n))
    print('%-44s' % label, ' '.join(out), flush=True)


timed('nothing before', lambda w: None)
timed('pack_results of 20 results before', lambda w: S.pack_results((w * 4)[:20], list(range(20)), dev))
timed('a small all_reduce before', lambda w: dist.all_reduce(x))
timed('gather_device of 20 results before', lambda w: S.gather_device((w * 4)[:20], list(range(20)), dev))
timed('nothing before (again)', lambda w: None)
dist.destroy_process_group()
