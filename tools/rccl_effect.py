"""Does an initialised RCCL communicator change how fast the big kernel runs?  The same pipelined sequence (96 frames,
batch 3) before init_process_group('nccl'), after it, after a first collective, and after destroying the group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.mapping.astrometry import frame_params
from auromat_amd.synthetic import sequence_frame, frame_image
W, H, N = 4240, 2832, 96
seq = SequencePipeline(W, H, pxPerDeg=10, shared_image=frame_image(W, H))
frames = []
for k in range(N):
    hdr, cam, t, _ = sequence_frame(k, W, H)
    frames.append((frame_params(hdr, 110, cam, t, True), cam, t, None))


def run(label):
    for rep in range(3):
        seq.process(frames)
    torch.cuda.synchronize()
    out = []
    for rep in range(3):
        seq.ctx.timing_enable(1)
        t0 = time.perf_counter()
        seq.process(frames)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g, n = seq.ctx.timing_read(0)
        seq.ctx.timing_enable(0)
        out.append('%.4f ms/frame (kernel %.4f)' % (dt / N * 1e3, g / n))
    print('%-34s' % label, '; '.join(out), flush=True)


run('before any process group')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29553')
mode = sys.argv[1] if len(sys.argv) > 1 else 'nccl'
if mode == 'nccl':
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
else:
    dist.init_process_group('gloo', rank=0, world_size=1)
run('after init_process_group(%s)' % mode)
x = torch.ones(1024, device='cuda') if mode == 'nccl' else torch.ones(1024)
dist.all_reduce(x)
torch.cuda.synchronize()
run('after a first all_reduce')
dist.destroy_process_group()
run('after destroy_process_group')
