"""One rank of tests/test_gpu_sharded_frame.py: rows of ONE frame over `world` processes (gloo; all on cuda:0)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_case(name):
    from auromat_amd.synthetic import frame_header, frame_image, pole_frame
    if name == 'pole':
        w, h = 400, 320
        hdr, cam, t = pole_frame(w, h)
    else:
        w, h = 512, 340
        hdr, cam, t = frame_header(w, h, 'iss029' if name == 'dateline' else name)
        if name == 'dateline':
            from datetime import timedelta
            t = t - timedelta(minutes=80)       # same inertial geometry, the Earth 20 deg further west: 162 E .. 172 W
    return hdr, cam, t, frame_image(w, h, seed=3)


def main():
    case, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    fast = sys.argv[6] == 'fast'
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % port, rank=rank, world_size=world)
    from auromat_amd.sequence import resample_frame_sharded
    hdr, cam, t, img = make_case(case)
    res, pipe, (y0, y1) = resample_frame_sharded(hdr, 110, cam, t, img, pxPerDeg=8, min_elevation=10, fast=fast)
    arrays = pipe.host_arrays()
    np.savez(out % rank, y0=y0, y1=y1, mean=res['mean'], count=res['count'], img=res['img'], mask=res['mask'],
             lat=res['lat'], lon=res['lon'], contains_pole=res['contains_pole'],
             contains_discontinuity=res['contains_discontinuity'], band_lat_c=arrays['lat_c'], band_lon_c=arrays['lon_c'],
             band_elev=arrays['elev'], band_lat=arrays['lat'])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
