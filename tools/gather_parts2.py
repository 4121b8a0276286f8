"""gather_device with an agreed capacity (one collective): where its time goes, 20 frames, one rank over RCCL."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from auromat_amd.pipeline import SequencePipeline
from auromat_amd import sequence as S
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import sequence_frame, frame_image
W, H, N = 4240, 2832, 20
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29557')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
seq = SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H))
dev = seq.ctx.device
frames = []
for k in range(N):
    hdr, cam, t, _ = sequence_frame(k, W, H)
    frames.append((frame_params(hdr, 110, cam, t, True), cam, t, None))
res = seq.process(frames)
idx = list(range(N))
cap = S.agree_capacity(res, idx, dev)
for rep in range(3):
    S.gather_device(res, idx, dev, capacity=cap)
torch.cuda.synchronize()
for rep in range(4):
    res = seq.process(frames)          # as in the bench: the gather follows a run whose last kernels are still in flight
    t0 = time.perf_counter()
    descs, payload = S.pack_results(res, idx, dev); t1 = time.perf_counter()
    mf, mp = cap
    n = mf * S.DESC_LEN + mp
    buf = torch.zeros(n + 2, dtype=torch.float64, device=dev); t2 = time.perf_counter()
    buf[:descs.numel()] = descs.reshape(-1)
    buf[mf * S.DESC_LEN:mf * S.DESC_LEN + payload.numel()] = payload; t3 = time.perf_counter()
    buf[n] = float(descs.shape[0]); buf[n + 1] = float(payload.numel()); t4 = time.perf_counter()
    bufs = [torch.empty_like(buf)]; t5 = time.perf_counter()
    dist.gather(buf, bufs, dst=0); t6 = time.perf_counter()
    torch.cuda.synchronize(); t7 = time.perf_counter()
    print('pack %.0f, zeros %.0f, two copies %.0f, two fills %.0f, empty_like %.0f, gather call %.0f, sync %.0f us | total %.0f' % tuple(
        (b - a) * 1e6 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6), (t6, t7), (t0, t7))), flush=True)
dist.destroy_process_group()
