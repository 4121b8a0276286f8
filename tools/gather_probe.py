"""Cost of the per-frame-grid gather (single rank over RCCL): pack, size exchange, padded gather."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from auromat_amd.pipeline import SequencePipeline
from auromat_amd import sequence as S
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29551')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
seq = SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H))
frames = [sequence_frame(k, W, H)[:3] + (None,) for k in range(30)]
res = seq.process(frames)
torch.cuda.synchronize()
dev = seq.ctx.device
for rep in range(3):
    t0 = time.perf_counter(); d, p = S.pack_results(res, list(range(30)), dev); torch.cuda.synchronize(); t1 = time.perf_counter()
    g = S.gather_device(res, list(range(30)), dev); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('pack %.2f ms  gather_device (incl. pack) %.2f ms  payload %.1f MB' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, p.numel() * 8 / 1e6))
dist.destroy_process_group()
