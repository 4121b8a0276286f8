"""One rank of tests/test_dist_gloo.py::test_a_failing_rank_ends_every_rank (gloo, no GPU): BASELINE configs[4]'s code path
(run_sequence / gather_checked) in which one rank's pipeline raises in the middle of its frames, or one rank's grids do not fit
the agreed gather capacity.  Every rank must END — with exit code 3 after a SequenceError that names the rank —, none may be left
waiting in a collective.  usage: _failing_rank_worker.py mode rank world port"""
import os
import sys
from datetime import timedelta

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def fake_result(k, cells=(4, 5)):
    import torch
    from auromat_amd.resample import _Grid
    grid = _Grid(cells, 40.0 + 0.3 * k, 43.0 + 0.37 * k, -100.0 + k, -96.5 + 1.2 * k)
    rs = np.random.RandomState(k)
    return dict(mean=torch.from_numpy(rs.uniform(0, 65535, (grid.ny, grid.nx, 4))),
                count=torch.from_numpy(rs.randint(0, 50, (grid.ny, grid.nx)).astype(np.float64)), grid=grid,
                contains_pole=False, contains_discontinuity=False, altitude=110.0, magnetic=False)


def main():
    mode, rank, world, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % port, rank=rank, world_size=world,
                            timeout=timedelta(seconds=60))
    from auromat_amd import pipeline, sequence
    from auromat_amd._native import NativeError

    class FakeContext(object):
        device = torch.device('cpu')

    class FakePipeline(object):
        """Stands in for SequencePipeline (no GPU here): frames are (index,) tuples; rank 2 of mode 'raise' fails at its third."""
        def __init__(self, *a, **kw):
            self.ctx = FakeContext()

        def process(self, frames, keep_on_device=True):
            out = []
            for n, f in enumerate(frames):
                if mode == 'raise' and rank == 2 and n == 2:
                    raise NativeError('amt_run_push failed (-3): hipErrorLaunchFailure (injected)')
                out.append(fake_result(f[0]))
            return out

    pipeline.SequencePipeline = FakePipeline
    frames = [(k, None, None, np.zeros((2, 2, 3), np.uint16)) for k in range(19)]
    try:
        if mode in ('raise', 'ok'):
            got = sequence.run_sequence(frames, 2, 2)
            assert (got is not None) == (rank == 0)
            if rank == 0:
                assert [f['index'] for f in got] == list(range(19))
        else:
            # 'overflow': the ranks agreed on a capacity in a warm-up with small grids; now rank 1's grids are larger
            mine = sequence.shard(len(frames), rank, world)
            warm = [fake_result(k) for k in mine]
            cap = sequence.agree_capacity(warm, mine, torch.device('cpu'))
            results = [fake_result(k, cells=(9, 9) if rank == 1 else (4, 5)) for k in mine]
            sequence.gather_checked(results, mine, torch.device('cpu'), capacity=cap)
    except sequence.SequenceError as e:
        sys.stderr.write('rank %d: SequenceError: %s\n' % (rank, e))
        dist.destroy_process_group()
        sys.exit(3)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
