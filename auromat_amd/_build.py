"""
Builds libauromat_hip.so (the C ABI of include/auromat_hip.h) in-tree with hipcc for gfx950.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_DIR = os.path.join(PKG_DIR, 'lib')
# AMT_LIB_PATH: load another build of the same sources (A/B experiments with compile-time variants)
LIB_PATH = os.environ.get('AMT_LIB_PATH') or os.path.join(LIB_DIR, 'libauromat_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared',
         '-Wall', '-Wextra', '-Wno-unused-parameter',
         # resolve libamdhip64.so.7 from the process (torch ships one with the same SONAME) or from ROCm
         '-Wl,-rpath,/opt/rocm/lib']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps():
    return sources() + [os.path.join(CSRC, 'amt_common.h'), os.path.join(CSRC, 'amt_grid.h'), os.path.join(CSRC, 'amt_params.h'),
                        os.path.join(os.path.dirname(PKG_DIR), 'include', 'auromat_hip.h')]


def sources_hash():
    """sha256 over the library's sources (names and contents) and the compiler flags"""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for d in _deps():
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as fp:
            h.update(fp.read())
    return h.hexdigest()


def needs_build():
    """The library is current when the hash recorded beside it at build time (``<library>.src``) is that of the sources — file
    times do not survive a copy of the tree to another machine in any particular order.  A library without that record (built
    by hand, ``tools/build_variant.sh``) falls back to the file times."""
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(LIB_PATH + '.src') as fp:
            return fp.read().strip() != sources_hash()
    except OSError:
        pass
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False):
    """Compile every HIP source into auromat_amd/lib/libauromat_hip.so. Returns the library path."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp = sources_hash()              # (of what is about to be compiled)
    tmp = LIB_PATH + '.tmp.%d' % os.getpid()
    # every source to an object of its own, a few at a time (the row kernel's file takes most of the time, the others
    # compile beside it), then one link
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    objdir = tempfile.mkdtemp(prefix='amt_build_')
    compile_flags = [f for f in FLAGS if f != '-shared' and not f.startswith('-Wl,')]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        cmd = [HIPCC] + compile_flags + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        return obj, res

    try:
        with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:
            done = list(pool.map(compile_one, sources()))
        for obj, res in done:
            if res.returncode != 0:
                raise RuntimeError('hipcc failed:\n' + res.stdout)
        cmd = [HIPCC] + FLAGS + ['-o', tmp] + [obj for obj, _ in done]
        if verbose:
            print(' '.join(cmd))
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if res.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError('hipcc (link) failed:\n' + res.stdout)
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
    os.replace(tmp, LIB_PATH)
    with open(LIB_PATH + '.src.tmp.%d' % os.getpid(), 'w') as fp:
        fp.write(stamp + '\n')
    os.replace(LIB_PATH + '.src.tmp.%d' % os.getpid(), LIB_PATH + '.src')
    return LIB_PATH


if __name__ == '__main__':
    print(build(force=True, verbose=True))
