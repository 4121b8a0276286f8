"""Where the time of a gathered run goes (one rank over RCCL): process() of 96 full-size frames, the gather of their
grids, the final synchronisation — against the same run without the gather."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from auromat_amd.pipeline import SequencePipeline
from auromat_amd import sequence as S
from auromat_amd.synthetic import sequence_frame, frame_image
W, H, N = 4240, 2832, 96
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29552')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
seq = SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H))
frames = [sequence_frame(k, W, H)[:3] + (None,) for k in range(N)]
dev = seq.ctx.device
for rep in range(4):
    res = seq.process(frames)
    S.gather_device(res, list(range(N)), dev)
    torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = seq.process(frames)
    t1 = time.perf_counter()
    g = S.gather_device(res, list(range(N)), dev)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    dist.barrier(); torch.cuda.synchronize()
    t4 = time.perf_counter()
    print('with gather: process %.2f ms, gather_device %.2f ms, sync %.2f ms, barrier+sync %.2f ms; total %.2f ms' % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3), flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = seq.process(frames)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print('without    : process %.2f ms, sync %.2f ms; total %.2f ms' % ((t1 - t0) * 1e3, (t3 - t1) * 1e3, (t3 - t0) * 1e3), flush=True)
# the pieces of gather_device
res = seq.process(frames); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); d, p = S.pack_results(res, list(range(N)), dev); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('pack_results: host %.2f ms, + sync %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
dist.destroy_process_group()
