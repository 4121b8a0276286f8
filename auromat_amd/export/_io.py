"""
The writers' host helper ``libauromat_io.so`` (``export/csrc/amt_io.cpp``: HDF5's shuffle + deflate over the chunks of a
variable, on threads, outside the interpreter lock): built in-tree with g++ against the system's zlib, loaded through ctypes.
It only makes file writing faster — the same zlib at the same level gives the same bytes as the writers' Python path, which
they take when the helper cannot be built or loaded (``AMT_IO_HELPER=0`` forces that, for A/B runs and tests).
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'amt_io.cpp')
LIB_PATH = os.path.join(_PKG, 'lib', 'libauromat_io.so')
_lib = []
# native jobs running right now (several writer threads x several variables each can call in at once): every job gets its share
# of the requested threads instead of all of them, so that a process stays near `threads` native threads in total
_active = [0]
_active_lock = threading.Lock()


class _share(object):
    """with _share(threads) as n: n = max(1, threads // jobs running, this one included)"""

    def __init__(self, threads):
        self.threads = int(threads)

    def __enter__(self):
        with _active_lock:
            _active[0] += 1
            return max(1, self.threads // _active[0])

    def __exit__(self, *exc):
        with _active_lock:
            _active[0] -= 1
        return False


def build(force=False):
    """g++ -O2 -shared -fPIC amt_io.cpp -lz -> auromat_amd/lib/libauromat_io.so; returns the path."""
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= os.path.getmtime(SRC):
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    tmp = LIB_PATH + '.tmp.%d' % os.getpid()
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-fPIC', '-shared', '-pthread', '-Wall', '-Wextra', SRC, '-lz', '-o', tmp]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError('g++ failed:\n' + res.stdout)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


def lib():
    """the loaded helper, or None (not built and not buildable here, or switched off)"""
    if not _lib:
        handle = None
        if os.environ.get('AMT_IO_HELPER', '1') != '0':
            try:
                handle = C.CDLL(build())
                handle.amt_io_deflate_bound.restype = C.c_int64
                handle.amt_io_deflate_bound.argtypes = [C.c_int64]
                handle.amt_io_deflate_chunks.restype = C.c_int
                handle.amt_io_deflate_chunks.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
                handle.amt_io_gzip_bound.restype = C.c_int64
                handle.amt_io_gzip_bound.argtypes = [C.c_int64, C.c_int64]
                handle.amt_io_gzip_parallel.restype = C.c_int
                handle.amt_io_gzip_parallel.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                                        C.c_int32]
            except (OSError, RuntimeError, AttributeError):
                handle = None
        _lib.append(handle)
    return _lib[0]


def deflate_rows(a, level, shuffle, threads):
    """``a``: C-contiguous array, one chunk per index of its first axis -> list of the chunks' zlib streams (bytes), or None
    when the helper is not there.  ctypes releases the interpreter lock for the call."""
    h = lib()
    if h is None or a.shape[0] == 0:
        return None
    n, chunk_bytes = a.shape[0], a.nbytes // a.shape[0]
    stride = int(h.amt_io_deflate_bound(chunk_bytes))
    out = np.empty((n, stride), np.uint8)
    sizes = np.empty(n, np.int64)
    with _share(threads) as nt:
        rc = h.amt_io_deflate_chunks(a.ctypes.data, n, chunk_bytes, a.dtype.itemsize, level, 1 if shuffle else 0, out.ctypes.data,
                                     stride, sizes.ctypes.data, nt)
    if rc != 0:
        raise RuntimeError('zlib error %d' % rc)
    return [out[i, :sizes[i]].tobytes() for i in range(n)]


def gzip_parallel(a, level, threads, block_bytes=1 << 20):
    """the bytes of the C-contiguous array ``a`` as ONE gzip member, deflated in blocks of ``block_bytes`` on ``threads`` threads
    (amt_io_gzip_parallel: pigz's layout) -> bytes, or None when the helper is not there"""
    h = lib()
    if h is None:
        return None
    n = a.nbytes
    cap = int(h.amt_io_gzip_bound(n, block_bytes))
    out = np.empty(cap, np.uint8)
    out_len = C.c_int64(0)
    with _share(threads) as nt:
        rc = h.amt_io_gzip_parallel(a.ctypes.data, n, level, block_bytes, out.ctypes.data, cap, C.byref(out_len), nt)
    if rc != 0:
        raise RuntimeError('zlib error %d' % rc)
    return out[:out_len.value].tobytes()
