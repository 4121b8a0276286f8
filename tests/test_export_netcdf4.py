"""
The netCDF-4 (HDF5) writer and reader of auromat_amd/export/_nc4.py — the container of the reference's files
(`Dataset(path, 'w', format='NETCDF4')`, zlib, one row per chunk: reference export/netcdf.py:48,128-326).
The files are read back by the HDF5 library itself where the image has one (h5py with HDF5 1.10 under /opt/conda: data, types,
chunk shapes, the shuffle + deflate pipeline, fill values, attributes, dimension scales with their reference lists, and the
recorded per-variable options of the reference's exporter), by this package's own reader everywhere, and the reader is checked
against a file the HDF5 library wrote.  CPU only.
"""
import json
import os
import subprocess

import numpy as np
import numpy.ma as ma
import pytest

from conftest import GOLDEN, load_golden

H5PY = '/opt/conda/bin/python3.9'


def have_h5py():
    if not os.path.exists(H5PY):
        return False
    return subprocess.run([H5PY, '-c', 'import h5py'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0


needs_h5py = pytest.mark.skipif(not have_h5py(), reason='no HDF5 library in this image')


def h5(script, *args):
    res = subprocess.run([H5PY, '-c', script] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert res.returncode == 0, res.stderr[-3000:]
    return res.stdout


DUMP = r'''
import sys, json, h5py, numpy as np
f = h5py.File(sys.argv[1], 'r')
out = {'attrs': {}, 'sets': {}}
def plain(v):
    if isinstance(v, bytes): return v.decode()
    a = np.asarray(v)
    if a.dtype.kind == 'V' or a.dtype.kind == 'O': return 'opaque'
    return [a.dtype.str] + a.ravel().tolist()
for k, v in f.attrs.items(): out['attrs'][k] = plain(v)
for name, d in f.items():
    rec = dict(shape=list(d.shape), dtype=d.dtype.str, chunks=None if d.chunks is None else list(d.chunks), compression=d.compression,
               opts=d.compression_opts, shuffle=bool(d.shuffle), fill=plain(d.fillvalue), attrs={k: plain(v) for k, v in d.attrs.items()},
               scale=bool(h5py.h5ds.is_scale(d.id)), dims=[[s.name for s in dim.values()] for dim in d.dims],
               order=list(d.attrs.keys()))
    rl = d.attrs.get('REFERENCE_LIST')
    if rl is not None: rec['refs'] = [[f[r[0]].name, int(r[1])] for r in rl]
    if d.shape == () or d.size: 
        try: np.save(sys.argv[2] + '/' + name + '.npy', d[...])
        except Exception as e: rec['error'] = repr(e)
    out['sets'][name] = rec
print(json.dumps(out))
'''


def primitives(w):
    rng = np.random.RandomState(0)
    w.attrs['Conventions'] = 'CF-1.6'
    w.attrs['number'] = np.float64(3.5)
    w.attrs['flags'] = np.array([1, 2, 3], np.int32)
    for d, n in (('y', 130), ('x', 70), ('vertex4', 4), ('xyz', 3), ('n', 1001), ('tall', 5003), ('unused', 9)):
        w.create_dimension(d, n)
    data = dict(lat=rng.rand(130, 70), img_red=rng.randint(0, 65535, (130, 70)).astype(np.int32), lat_bounds=rng.rand(130, 70, 4),
                zen=rng.rand(130, 70).astype(np.float32), camera_pos=np.array([1., 2., 3.]), one=np.arange(1001, dtype=np.float32),
                big=rng.rand(5003, 70), edge=(rng.rand(5003, 70) * 100).astype(np.int16), b=np.array([1, -2, 3], np.int8),
                plain=rng.rand(5003, 70).astype(np.float32))
    v = w.create_variable('lat', np.float64, ('y', 'x'), zlib=True, chunksizes=(1, 70))
    v.attrs['units'] = 'degrees_north'
    v.attrs['valid_min'] = np.float64(-90)
    v = w.create_variable('img_red', np.int32, ('y', 'x'), fill_value=np.int32(-2 ** 31), zlib=True, chunksizes=(1, 70))
    v.attrs['actual_range'] = np.int32([5, 7])
    w.create_variable('lat_bounds', np.float64, ('y', 'x', 'vertex4'), zlib=True, chunksizes=(1, 70, 4))
    w.create_variable('zen', np.float32, ('y', 'x'), zlib=True, chunksizes=(1, 70))
    w.create_variable('camera_pos', np.float64, ('xyz',))
    w.create_variable('one', np.float32, ('n',), zlib=True)                                  # one chunk by default
    w.create_variable('big', np.float64, ('tall', 'x'), zlib=True, chunksizes=(1, 70))       # 5003 chunks: a three-level B-tree
    w.create_variable('edge', np.int16, ('tall', 'x'), fill_value=np.int16(-1), zlib=True, chunksizes=(64, 32))     # ragged edges
    w.create_variable('plain', np.float32, ('tall', 'x'), chunksizes=(500, 70))              # chunked, no filter
    w.create_variable('b', np.int8, ('xyz',))
    for k, a in data.items():
        w.vars[k].set(a)
    t = w.create_variable('time', np.float64)
    t.attrs['units'] = 'seconds since 1970-01-01 00:00:00'
    t.set(12345.5)
    data['time'] = np.float64(12345.5)
    w.create_variable('crs', np.int8).attrs['comment'] = 'holds no data'
    return data


def test_own_reader_reads_what_the_writer_writes(tmp_path):
    from auromat_amd.export import _nc4
    w = _nc4.Writer()
    data = primitives(w)
    path = str(tmp_path / 'p.nc')
    w.write(path)
    f = _nc4.open_file(path)
    assert isinstance(f, _nc4.File)
    assert list(f.dims.items()) == [('y', 130), ('x', 70), ('vertex4', 4), ('xyz', 3), ('n', 1001), ('tall', 5003), ('unused', 9)]
    assert f.attrs['Conventions'] == 'CF-1.6' and f.attrs['number'] == 3.5 and f.attrs['flags'].tolist() == [1, 2, 3]
    assert sorted(f.vars) == sorted(list(data) + ['crs'])
    for k, a in data.items():
        v = f.vars[k]
        assert v.data.dtype == np.asarray(a).dtype and np.array_equal(v.data, a), k
        assert v.dims == w.vars[k].dims
    assert f.vars['lat'].attrs == {'units': 'degrees_north', 'valid_min': -90.0}
    assert f.vars['img_red'].attrs['_FillValue'] == -2 ** 31 and list(f.vars['img_red'].attrs) == ['_FillValue', 'actual_range']
    assert f.vars['crs'].comment == 'holds no data' and f.vars['crs'].data == -127       # the library's default fill value


@needs_h5py
def test_hdf5_library_reads_the_file(tmp_path):
    from auromat_amd.export import _nc4
    w = _nc4.Writer()
    data = primitives(w)
    path = str(tmp_path / 'p.nc')
    w.write(path)
    out = json.loads(h5(DUMP, path, str(tmp_path)))
    assert out['attrs'] == {'Conventions': 'CF-1.6', 'number': ['<f8', 3.5], 'flags': ['<i4', 1, 2, 3]}
    sets = out['sets']
    for k, a in data.items():
        assert 'error' not in sets[k], sets[k]
        got = np.load(str(tmp_path / (k + '.npy')))
        assert got.dtype == np.asarray(a).dtype and np.array_equal(got, a), k
    assert sets['lat']['chunks'] == [1, 70] and sets['lat']['compression'] == 'gzip' and sets['lat']['opts'] == 4 and sets['lat']['shuffle']
    assert sets['lat_bounds']['chunks'] == [1, 70, 4] and sets['big']['chunks'] == [1, 70] and sets['edge']['chunks'] == [64, 32]
    assert sets['one']['chunks'] == [1001] and sets['plain']['chunks'] == [500, 70] and sets['plain']['compression'] is None
    assert sets['camera_pos']['chunks'] is None and sets['time']['shape'] == []
    assert sets['img_red']['fill'] == ['<i4', -2 ** 31] and sets['edge']['fill'] == ['<i2', -1]
    assert sets['lat']['fill'] == ['<f8', 9.969209968386869e+36]                              # NC_FILL_DOUBLE
    # dimension scales, both directions of the links
    for d, n in (('y', 130), ('x', 70), ('vertex4', 4), ('xyz', 3), ('n', 1001), ('tall', 5003), ('unused', 9)):
        s = sets[d]
        assert s['scale'] and s['shape'] == [n] and s['attrs']['CLASS'] == 'DIMENSION_SCALE'
        assert s['attrs']['NAME'] == 'This is a netCDF dimension but not a netCDF variable.%10d' % n
    assert [sets[d]['attrs']['_Netcdf4Dimid'][1] for d in ('y', 'x', 'vertex4', 'xyz', 'n', 'tall', 'unused')] == list(range(7))
    assert sets['lat_bounds']['dims'] == [['/y'], ['/x'], ['/vertex4']] and sets['big']['dims'] == [['/tall'], ['/x']]
    assert sets['time']['dims'] == [] and 'refs' not in sets['unused']
    assert sorted(map(tuple, sets['x']['refs'])) == sorted([('/lat', 1), ('/img_red', 1), ('/lat_bounds', 1), ('/zen', 1), ('/big', 1),
                                                            ('/edge', 1), ('/plain', 1)])
    assert sorted(sets['img_red']['order']) == ['DIMENSION_LIST', '_FillValue', 'actual_range']       # (listed by name)


@needs_h5py
@pytest.mark.parametrize('case', ['resampled', 'unresampled'])
def test_exported_mapping_through_the_hdf5_library(case, tmp_path):
    """export.netcdf.write(format='NETCDF4') of the reference's recorded cases: every variable's zlib / chunksizes as the
    reference passes them to createVariable (tests/golden/netcdf_layout_*.json 'options'), data and attributes as recorded."""
    from test_export_netcdf import Mapping
    from auromat_amd.export.netcdf import write
    layout = json.load(open(os.path.join(GOLDEN, 'netcdf_layout_%s.json' % case)))
    z = load_golden('netcdf_case_%s.npz' % case)
    m = Mapping(z, {'Project': 'auromat', 'Calibrated': True} if case.startswith('unresampled') else {'Project': 'auromat'}, case)
    path = str(tmp_path / (case + '.nc'))
    write(path, m, metadata={'Source_name': 'test'})
    out = json.loads(h5(DUMP, path, str(tmp_path)))
    sets = out['sets']
    dims = dict(map(tuple, layout['dims']))
    for rec in layout['vars']:
        s = sets[rec['name']]
        opt = rec['options']
        shape = [dims[d] for d in rec['dims']]
        assert s['shape'] == shape and np.dtype(s['dtype']) == np.dtype(rec['dtype']), rec['name']
        if opt.get('zlib'):
            assert s['compression'] == 'gzip' and s['shuffle'], rec['name']
            assert s['chunks'] == (list(opt['chunksizes']) if opt.get('chunksizes') else shape), rec['name']
        else:
            assert s['compression'] is None and s['chunks'] is None, rec['name']
        assert s['dims'] == [['/' + d] for d in rec['dims']], rec['name']
        if 'var_' + rec['name'] in z.files:
            got = np.load(str(tmp_path / (rec['name'] + '.npy')))
            assert np.array_equal(got, np.asarray(z['var_' + rec['name']], dtype=got.dtype).reshape(got.shape), equal_nan=True), rec['name']
    assert [[d, sets[d]['shape'][0]] for d in sorted(dims, key=lambda d: sets[d]['attrs']['_Netcdf4Dimid'][1])] == layout['dims']


@needs_h5py
def test_attributes_the_container_cannot_hold_are_refused_before_anything_is_written(tmp_path):
    """A version-1 header message has a 16-bit size: an attribute of 64 KiB or more (a long history string) and an empty
    numeric attribute are refused with a clear error by check_attribute — what export.netcdf.write calls for every global
    attribute before it writes — and by the writer itself, instead of a struct.error in the middle of write()."""
    from auromat_amd.export import _nc4
    _nc4.check_attribute('x' * 60000)                              # just below the limit: fine
    _nc4.check_attribute(np.arange(8000, dtype=np.float64))
    with pytest.raises(ValueError, match='64 KiB'):
        _nc4.check_attribute('history ' * 9000)                    # 72 KB
    with pytest.raises(ValueError, match='64 KiB'):
        _nc4.check_attribute(np.zeros(9000))
    with pytest.raises(ValueError, match='empty'):
        _nc4.check_attribute(np.zeros(0))
    with pytest.raises(TypeError):
        _nc4.check_attribute(object())
    w = _nc4.Writer()
    primitives(w)
    w.attrs['history'] = 'h' * 70000
    path = str(tmp_path / 'big_attr.nc')
    with pytest.raises(ValueError, match='64 KiB'):
        w.write(path)
    w.attrs['history'] = 'h' * 50000
    w.write(path)
    assert _nc4.open_file(path).attrs['history'] == 'h' * 50000


def test_own_reader_reads_a_file_of_the_hdf5_library(tmp_path):
    """The reader against the other implementation: a file h5py wrote with its default (oldest) structures, chunked, shuffled,
    deflated, with attributes and attached dimension scales."""
    from auromat_amd.export import _nc4
    path = str(tmp_path / 'lib.h5')
    rng = np.random.RandomState(2)
    a, b = rng.rand(300, 40), rng.randint(-5, 5, (300, 40)).astype(np.int16)
    np.save(str(tmp_path / 'a.npy'), a)
    np.save(str(tmp_path / 'b.npy'), b)
    h5(r'''
import sys, h5py, numpy as np
d = sys.argv[2]
a, b = np.load(d + '/a.npy'), np.load(d + '/b.npy')
with h5py.File(sys.argv[1], 'w') as f:
    f.attrs['title'] = np.string_('made by the library')
    f.attrs['k'] = np.int32(7)
    y = f.create_dataset('y', data=np.arange(300, dtype=np.float32)); x = f.create_dataset('x', data=np.arange(40, dtype=np.float32))
    y.make_scale('y'); x.make_scale('x')
    da = f.create_dataset('a', data=a, chunks=(7, 40), compression='gzip', compression_opts=6, shuffle=True)
    db = f.create_dataset('b', data=b, chunks=(64, 16), compression='gzip')
    dc = f.create_dataset('c', data=np.float64([1.5, 2.5]))
    dp = f.create_dataset('p', shape=(100, 8), dtype='i4', chunks=(10, 8), compression='gzip', fillvalue=-7)
    dp[20:30] = np.arange(80, dtype=np.int32).reshape(10, 8)          # one chunk of ten is ever written
    da.attrs['units'] = np.string_('m'); da.attrs['range'] = np.float64([0, 1])
    for ds in (da, db):
        ds.dims[0].attach_scale(y); ds.dims[1].attach_scale(x)
''', path, str(tmp_path))
    f = _nc4.open_file(path)
    assert f.attrs['title'] == 'made by the library' and f.attrs['k'] == 7
    assert np.array_equal(f.vars['a'].data, a) and np.array_equal(f.vars['b'].data, b) and f.vars['c'].data.tolist() == [1.5, 2.5]
    assert f.vars['a'].attrs['units'] == 'm' and f.vars['a'].attrs['range'].tolist() == [0.0, 1.0]
    assert f.vars['a'].dims == ('y', 'x') and f.vars['b'].dims == ('y', 'x')
    assert np.array_equal(f.vars['y'].data, np.arange(300, dtype=np.float32))               # a coordinate variable
    p = f.vars['p'].data
    assert np.array_equal(p[20:30], np.arange(80).reshape(10, 8)) and (p[:20] == -7).all() and (p[30:] == -7).all()


def test_files_of_the_netcdf_library_are_refused_with_a_reason(tmp_path):
    from auromat_amd.export import _nc4
    path = str(tmp_path / 'v2.h5')
    with open(path, 'wb') as fp:
        fp.write(b'\x89HDF\r\n\x1a\n' + bytes([2, 8, 8, 0]) + b'\0' * 100)
    with pytest.raises(NotImplementedError, match='nccopy'):
        _nc4.open_file(path)
    with open(path, 'wb') as fp:
        fp.write(b'not a file of either kind')
    with pytest.raises(ValueError):
        _nc4.open_file(path)


def test_export_formats_and_sizes(tmp_path):
    """The two containers of one mapping: same content through the readers, and the compressed file is the smaller one."""
    from test_export_netcdf import Mapping
    from auromat_amd.export import _nc4
    from auromat_amd.export.netcdf import write
    z = load_golden('netcdf_case_resampled.npz')
    m = Mapping(z, {'Project': 'auromat'}, 'x')
    p4, p3 = str(tmp_path / 'a4.nc'), str(tmp_path / 'a3.nc')
    write(p4, m)
    write(p3, m, format='NETCDF3_64BIT')
    with pytest.raises(ValueError):
        write(p3, m, format='NETCDF5')
    f4, f3 = _nc4.open_file(p4), _nc4.open_file(p3)
    assert list(f4.dims.items()) == list(f3.dims.items()) and list(f4.attrs) == list(f3.attrs)
    assert sorted(f4.vars) == sorted(f3.vars)
    for k, v in f3.vars.items():
        if k not in ('crs', 'mcrs'):        # (hold no data: the library's fill value in the HDF5 file, zero in the classic one)
            assert np.array_equal(f4.vars[k].data, v.data, equal_nan=True), k
        assert f4.vars[k].dims == v.dims, k
        assert list(f4.vars[k].attrs) == list(v.attrs), k
    assert os.path.getsize(p4) < os.path.getsize(p3) * 1.5                # (small grids: the HDF5 structures weigh in)


@needs_h5py
def test_random_layouts_through_the_hdf5_library(tmp_path):
    """Forty random variables — ranks 0 to 3, every type, chunked or not, filtered or not, chunk shapes that do and do not divide
    the array, with and without a fill value, up to a few thousand chunks — in one file: the HDF5 library and the reader here
    return every array as it went in."""
    from auromat_amd.export import _nc4
    rng = np.random.RandomState(11)
    w = _nc4.Writer()
    sizes = dict(a=1, b=2, c=7, d=33, e=64, f=65, g=130, h=1000, i=2049)
    for d, n in sizes.items():
        w.create_dimension(d, n)
    types = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.float32, np.float64]
    data = {}
    for k in range(40):
        rank = int(rng.randint(0, 4))
        dims = tuple(str(d) for d in rng.choice(list(sizes), rank, replace=True)) if rank else ()
        shape = tuple(sizes[d] for d in dims)
        if int(np.prod(shape)) > 3_000_000:
            dims, shape = dims[:2], shape[:2]
        dt = np.dtype(types[int(rng.randint(len(types)))])
        arr = np.asarray(rng.rand(*shape) * 200 - (0 if dt.kind == 'u' else 100)).astype(dt)
        chunks, zlib = None, False
        if shape and rng.rand() < 0.7:
            zlib = bool(rng.rand() < 0.7)
            if rng.rand() < 0.8:
                chunks = tuple(int(rng.randint(1, n + 1)) for n in shape)
                while int(np.prod([-(-n // c) for n, c in zip(shape, chunks)])) > 6000:
                    chunks = tuple(min(n, c * 2) for n, c in zip(shape, chunks))
            elif not zlib:
                chunks = None
        fill = dt.type(3) if rng.rand() < 0.4 else None
        name = 'v%02d' % k
        v = w.create_variable(name, dt, dims, fill_value=fill, zlib=zlib, chunksizes=chunks)
        v.attrs['k'] = np.int32(k)
        v.set(arr)
        data[name] = (arr, dims, chunks, zlib)
    path = str(tmp_path / 'r.nc')
    w.write(path)
    f = _nc4.open_file(path)
    for name, (arr, dims, chunks, zlib) in data.items():
        assert f.vars[name].dims == dims and np.array_equal(f.vars[name].data, arr) and f.vars[name].attrs['k'] == int(name[1:]), name
    out = json.loads(h5(DUMP, path, str(tmp_path)))
    for name, (arr, dims, chunks, zlib) in data.items():
        s = out['sets'][name]
        assert 'error' not in s, (name, s)
        got = np.load(str(tmp_path / (name + '.npy')))
        assert got.dtype == arr.dtype and np.array_equal(got, arr), name
        assert s['dims'] == [['/' + d] for d in dims], name
        assert (s['compression'] == 'gzip') == (zlib and bool(dims)), name
        if chunks is not None:
            assert s['chunks'] == list(chunks), name


def test_host_helper_gives_the_python_paths_bytes(tmp_path, monkeypatch):
    """export/csrc/amt_io.cpp (shuffle + deflate of a variable's rows on threads, outside the interpreter lock) against
    zlib.compress of the shuffled rows, and a whole file written with and without it."""
    import zlib
    from auromat_amd.export import _io, _nc4
    if _io.lib() is None:
        pytest.skip('libauromat_io.so not built / switched off')
    rs = np.random.RandomState(5)
    for dtype, shape in (('f8', (70, 33)), ('f4', (40, 17)), ('i4', (35, 5, 3)), ('i2', (3, 1000)), ('i1', (64, 64))):
        a = np.ascontiguousarray((np.cumsum(rs.rand(*shape), axis=1) * 50).astype(dtype))
        it = a.dtype.itemsize
        for threads in (1, 5):
            got = _io.deflate_rows(a.reshape(shape[0], -1), 4, True, threads)
            want = [zlib.compress(np.ascontiguousarray(a[i].reshape(-1).view(np.uint8).reshape(-1, it).T).tobytes(), 4) for i in range(shape[0])]
            assert got == want, (dtype, threads)
        assert _io.deflate_rows(a.reshape(shape[0], -1), 6, False, 2) == [zlib.compress(a[i].tobytes(), 6) for i in range(shape[0])]

    def write(path):
        w = _nc4.Writer()
        w.create_dimension('y', 90)
        w.create_dimension('x', 41)
        w.create_dimension('c', 3)
        rs = np.random.RandomState(6)
        w.create_variable('a', 'f8', ('y', 'x'), zlib=True, chunksizes=(1, 41)).set(np.cumsum(rs.rand(90, 41), axis=1))
        w.create_variable('b', 'i4', ('y', 'x', 'c'), zlib=True, chunksizes=(1, 41, 3), fill_value=-1).set(rs.randint(0, 9, (90, 41, 3)))
        w.write(path)
    write(str(tmp_path / 'with.nc'))
    monkeypatch.setattr(_io, '_lib', [None])
    assert _io.lib() is None
    write(str(tmp_path / 'without.nc'))
    assert open(str(tmp_path / 'with.nc'), 'rb').read() == open(str(tmp_path / 'without.nc'), 'rb').read()
