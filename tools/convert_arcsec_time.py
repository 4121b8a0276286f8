"""`auromat-convert --resample --resolution 100` (the reference's default flags, cli/convert.py:176-185) on N full-size frames:
the sequence pipeline's box-first plan against the mapping classes frame by frame (AMT_CONVERT_CLASSES=1, what round 3 ran), and
the sequence loop alone (device-resident images, grids left on the device) against --px-per-deg."""
import os, sys, time, tempfile, shutil, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from auromat_amd.pipeline import SequencePipeline
from auromat_amd.synthetic import sequence_frame, frame_image
W, H, N = 4240, 2832, int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device('cuda', 0)
imgs = [torch.randint(0, 65535, (H, W, 3), device=dev, dtype=torch.int32).to(torch.int16) for _ in range(8)]
frames = [sequence_frame(k, W, H)[:3] + (imgs[k % 8],) for k in range(N + 9)]
for label, kw in (('pxPerDeg=10', dict(pxPerDeg=10)), ('arcsecPerPx=100 (box-first)', dict(arcsecPerPx=100)),
                  ('arcsecPerPx=100, MLat/MLT grid', dict(arcsecPerPx=100, magnetic=True))):
    seq = SequencePipeline(W, H, keep_coordinates=False, **kw)
    for _ in range(3):
        seq.process(frames[:9]); torch.cuda.synchronize()
    seq.ctx.timing_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = seq.process(frames[9:]); torch.cuda.synchronize(); el = time.perf_counter() - t0
    ms, n = seq.ctx.timing_read(0); seq.ctx.timing_enable(False)
    print('SequencePipeline(%s, grids only): %.4f ms per frame, %.0f Mpixel/s; kernels (box pass + fused) %.1f us per frame; plans %s'
          % (label, el / N * 1e3, N * W * H / 1e6 / el, ms / N * 1e3, sorted(set(seq.plans))), flush=True)
    del seq, r
# the CLI on files (.npy images: no JPEG decode), 12 frames
from auromat_amd.cli.convert import main
d = tempfile.mkdtemp()
try:
    for k in range(12):
        hdr, cam, t, seed = sequence_frame(k, W, H)
        np.save(os.path.join(d, 'f%02d.npy' % k), frame_image(W, H, seed=seed))
        hdr = dict(hdr, POSX=float(cam[0]), POSY=float(cam[1]), POSZ=float(cam[2]))
        hdr['DATE-OBS'] = t.strftime('%Y-%m-%dT%H:%M:%S.%f')
        json.dump(hdr, open(os.path.join(d, 'f%02d.json' % k), 'w'))
    for label, env in (('sequence pipeline (box-first)', None), ('mapping classes, frame by frame (round 3)', '1')):
        for rep in range(2):
            out = os.path.join(d, 'out_%s_%d' % (env, rep))
            if env:
                os.environ['AMT_CONVERT_CLASSES'] = env
            torch.cuda.synchronize(); t0 = time.perf_counter()
            main(['--data', d, '--format', 'netcdf', '--resample', '--min-elevation', '10', '--out', out, '--without-bounds'])
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            os.environ.pop('AMT_CONVERT_CLASSES', None)
        print('auromat-convert --resample (100 arcsec/px, MLat/MLT grid), 12 frames from .npy files, %s: %.1f ms per frame (second run)'
              % (label, el / 12 * 1e3), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
