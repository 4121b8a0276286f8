"""The parts of sequence.gather_device for 20 and 192 frames (one rank over RCCL): packing, size exchange + host copy of the
sizes, padded buffer, gather, final synchronisation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from auromat_amd.pipeline import SequencePipeline
from auromat_amd import sequence as S
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import sequence_frame, frame_image
W, H = 4240, 2832
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29556')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
seq = SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H))
dev = seq.ctx.device
for N in (20, 192):
    frames = []
    for k in range(N):
        hdr, cam, t, _ = sequence_frame(k, W, H)
        frames.append((frame_params(hdr, 110, cam, t, True), cam, t, None))
    res = seq.process(frames)
    S.gather_device(res, list(range(N)), dev)
    torch.cuda.synchronize()
    for rep in range(3):
        t = [time.perf_counter()]
        descs, payload = S.pack_results(res, list(range(N)), dev); t.append(time.perf_counter())
        sizes = torch.tensor([descs.shape[0], payload.numel()], dtype=torch.int64, device=dev)
        all_sizes = [torch.zeros_like(sizes)]
        dist.all_gather(all_sizes, sizes); t.append(time.perf_counter())
        a = torch.stack(all_sizes).cpu().numpy(); t.append(time.perf_counter())
        mf, mp = int(a[:, 0].max()), int(a[:, 1].max())
        buf = torch.empty(mf * S.DESC_LEN + mp, dtype=torch.float64, device=dev)
        buf[:descs.numel()] = descs.reshape(-1)
        buf[mf * S.DESC_LEN:mf * S.DESC_LEN + payload.numel()] = payload; t.append(time.perf_counter())
        bufs = [torch.empty_like(buf)]
        dist.gather(buf, bufs, dst=0); t.append(time.perf_counter())
        torch.cuda.synchronize(); t.append(time.perf_counter())
        names = ('pack', 'all_gather call', 'sizes to host', 'padded buffer', 'gather call', 'final sync')
        print(N, 'frames:', ', '.join('%s %.0f us' % (n, (b - a_) * 1e6) for n, a_, b in zip(names, t, t[1:])), '| total %.0f us' % ((t[-1] - t[0]) * 1e6), flush=True)
dist.destroy_process_group()
