"""auromat-convert --resample (the reference's default flags: 100 arcsec per pixel, MLat/MLT grid, bounds and MLat/MLT stored) over
24 full-size frames from .npy files: netCDF-4 and CDF, files written in line (AMT_CONVERT_WRITERS=0) and by writer threads."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from auromat_amd.cli.convert import main
from auromat_amd.synthetic import frame_image, sequence_frame
W, H, N = 4240, 2832, 24
d = tempfile.mkdtemp()
try:
    for k in range(N):
        hdr, cam, t, seed = sequence_frame(k, W, H)
        np.save(os.path.join(d, 'f%02d.npy' % k), frame_image(W, H, seed=seed))
        hdr = dict(hdr, POSX=float(cam[0]), POSY=float(cam[1]), POSZ=float(cam[2]))
        hdr['DATE-OBS'] = t.strftime('%Y-%m-%dT%H:%M:%S.%f')
        json.dump(hdr, open(os.path.join(d, 'f%02d.json' % k), 'w'))
    for fmt in ('netcdf', 'cdf'):
        for writers in ('0', '4', '8'):
            os.environ['AMT_CONVERT_WRITERS'] = writers
            for rep in range(2):
                out = os.path.join(d, 'out_%s_%s_%d' % (fmt, writers, rep))
                torch.cuda.synchronize(); t0 = time.perf_counter()
                main(['--data', d, '--format', fmt, '--resample', '--min-elevation', '10', '--out', out])
                torch.cuda.synchronize(); el = time.perf_counter() - t0
            size = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) / N
            print('--format %-6s AMT_CONVERT_WRITERS=%s: %.1f ms per frame (second run), %.0f KB per file' % (fmt, writers, el / N * 1e3, size / 1e3), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
