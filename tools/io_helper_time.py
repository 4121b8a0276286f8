import sys, os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from auromat_amd.export import _io
import numpy as np, time
for rows in (236, 2832):
    a = np.cumsum(np.random.rand(rows, 238 if rows == 236 else 4256), axis=1)
    for nt in (1, 2, 4, 8, 16):
        _io.deflate_rows(a, 4, True, nt)
        t = time.perf_counter(); [_io.deflate_rows(a, 4, True, nt) for _ in range(5)]; print(rows, nt, '%.2f ms' % ((time.perf_counter() - t) / 5 * 1e3), flush=True)
