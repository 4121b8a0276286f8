"""
CDF export (auromat_amd/export/cdf.py over export/_cdf3.py) — the reference's exporter auromat/export/cdf.py:25-285 writes
through spacepy.pycdf / NASA's CDF library, neither of which exists in this image, so NOTHING here compares against a file
or a reading of the real library (parity of the container: unpinned).  What is held:

* the record structure of a written file against the published CDF Internal Format Description (record sizes and types,
  the chains from the GDR, counts, every byte of the file owned by exactly one record; `file(1)`'s magic),
* the variables, types, record variances, compression, attributes and their order that the reference's exporter creates
  (read off export/cdf.py line by line into EXPECTED below),
* a round trip of the reference's own exported mappings (the arrays of tests/golden/netcdf_case_*.npz) through the
  independent reader and CDFMapping (the round trip of the reference's test/export_cdf_test.py),
* CDF_TIME_TT2000 / CDF_EPOCH against their published constants.
CPU only (the exporter is host code).
"""
import gzip
import struct
from datetime import datetime, timedelta

import numpy as np
import numpy.ma as ma
import pytest

from conftest import load_golden
from test_export_netcdf import CASES, Mapping


def test_time_encodings():
    from auromat_amd.export import _cdf3 as C
    # TT2000: J2000 is 2000-01-01T12:00:00 TT = 11:58:55.816 UTC (TAI - UTC = 32 s then, TT - TAI = 32.184 s)
    assert C.datetime_to_tt2000(datetime(2000, 1, 1, 11, 58, 55, 816000)) == 0
    assert C.datetime_to_tt2000(datetime(2000, 1, 1, 12)) == 64184000000
    # the first instant after the leap second of 2016-12-31: 6209.5 days after J2000 noon, TAI - UTC = 37 s
    assert C.datetime_to_tt2000(datetime(2017, 1, 1)) == 536500869184000000
    assert C.datetime_to_tt2000(datetime(2016, 12, 31, 23, 59, 59)) == 536500869184000000 - 2 * 10**9      # the leap second between
    assert C.datetime_to_tt2000(datetime(2012, 1, 25, 9, 26, 55, 60000)) == (
        ((datetime(2012, 1, 25, 9, 26, 55) - datetime(2000, 1, 1, 12)).days * 86400 +
         (datetime(2012, 1, 25, 9, 26, 55) - datetime(2000, 1, 1, 12)).seconds + 34) * 10**9 + 60000000 + 32184000000)
    for day, _ in C.LEAP_SECONDS[1:]:
        for dt in (datetime(*day) - timedelta(seconds=1), datetime(*day), datetime(*day) + timedelta(microseconds=1)):
            assert C.tt2000_to_datetime(C.datetime_to_tt2000(dt)) == dt
    with pytest.raises(ValueError):
        C.datetime_to_tt2000(datetime(1971, 12, 31))
    # CDF_EPOCH: milliseconds since 0000-01-01; the Unix epoch is 62167219200000 ms, 2000-01-01 63113904000000 ms
    assert C.datetime_to_epoch(datetime(1970, 1, 1)) == 62167219200000.0
    assert C.datetime_to_epoch(datetime(2000, 1, 1)) == 63113904000000.0
    t = datetime(2012, 1, 25, 9, 26, 55, 60000)
    assert C.epoch_to_datetime(C.datetime_to_epoch(t)) == t


def test_type_inference():
    from auromat_amd.export import _cdf3 as C
    assert C.infer('abc') == (C.CDF_CHAR, 3, b'abc') and C.infer('') == (C.CDF_CHAR, 1, b' ')
    assert C.infer(0)[0] == C.CDF_BYTE and C.infer(200)[0] == C.CDF_UINT1 and C.infer(-200)[0] == C.CDF_INT2
    assert C.infer(65535)[0] == C.CDF_UINT2 and C.infer(65536)[0] == C.CDF_INT4 and C.infer(2**40)[0] == C.CDF_INT8
    assert C.infer(True)[0] == C.CDF_BYTE
    assert C.infer(1.5) == (C.CDF_DOUBLE, 1, struct.pack('<d', 1.5))
    assert C.infer(np.float32(1.5)) == (C.CDF_FLOAT, 1, struct.pack('<f', 1.5))
    assert C.infer(np.int32(-2**31)) == (C.CDF_INT4, 1, struct.pack('<i', -2**31))
    assert C.infer(np.array([1, 2], np.uint16)) == (C.CDF_UINT2, 2, struct.pack('<HH', 1, 2))
    assert C.infer(datetime(2000, 1, 1, 12)) == (C.CDF_TIME_TT2000, 1, struct.pack('<q', 64184000000))
    assert C.infer(datetime(2000, 1, 1), tt2000=False) == (C.CDF_EPOCH, 1, struct.pack('<d', 63113904000000.0))
    for bad in (np.array(['a', 'b']), {'a': 1}, np.complex64(1)):
        with pytest.raises(TypeError):
            C.infer(bad)


def scan(path):
    """the records of a file in file order, found WITHOUT following any pointer: [(offset, size, type)]"""
    buf = open(path, 'rb').read()
    at, out = 8, []
    while at < len(buf):
        size, kind = struct.unpack_from('>qi', buf, at)
        assert size >= 12 and at + size <= len(buf), (at, size, kind)
        out.append((at, size, kind))
        at += size
    assert at == len(buf)
    return buf, out


def test_container_structure(tmp_path):
    """Sizes, types, chains and counts as the CDF Internal Format Description gives them."""
    from auromat_amd.export import _cdf3 as C
    w = C.Writer()
    w.attrs['title'] = 'structure'
    w.attrs['numbers'] = np.array([1.5, 2.5])
    w.attrs['texts'] = ['a', 'bcd']
    w.attrs['when'] = datetime(2012, 1, 25, 9, 26, 55, 60000)
    w.new('Epoch', [datetime(2012, 1, 25), datetime(2012, 1, 26)], type=C.CDF_TIME_TT2000).attrs['VAR_TYPE'] = 'support_data'
    rs = np.random.RandomState(0)
    big = rs.rand(9, 5, 6)                                   # 9 records, compressed: two VXRs (7 entries each)
    v = w.new('big', big, compress=C.GZIP_COMPRESSION)
    v.attrs['VAR_TYPE'] = 'data'
    v.attrs['VALIDMIN'] = 0.0
    w.new('plain', rs.randint(0, 65535, (2, 3, 4)).astype(np.uint16)).attrs['VALIDMAX'] = 65535
    w.new('scalar', 7, recVary=False).attrs['VAR_NOTES'] = ''
    w.new('nrv', np.arange(6, dtype=np.float32).reshape(2, 3), recVary=False)
    path = str(tmp_path / 's.cdf')
    w.write(path)
    buf, recs = scan(path)
    assert buf[:8] == b'\xcd\xf3\x00\x01\x00\x00\xff\xff'
    kinds = [k for _, _, k in recs]
    # one CDR (312 bytes) at 8, one GDR (84 bytes, no rVariable dimensions) behind it
    assert recs[0] == (8, 312, 1) and recs[1] == (320, 84, 2) and kinds.count(1) == 1 and kinds.count(2) == 2 - 1
    gdr_off, version, release, encoding, flags = struct.unpack_from('>qiiii', buf, 8 + 12)
    assert gdr_off == 320 and version == 3 and release >= 4 and encoding == 6 and flags & 3 == 3     # TT2000 needs 3.4+; row major, single file
    rvdr, zvdr, adr, eof, n_r, n_attr, r_max, r_dims, n_z, uir = struct.unpack_from('>qqqqiiiiiq', buf, 320 + 12)
    assert rvdr == 0 and n_r == 0 and r_max == -1 and r_dims == 0 and uir == 0 and eof == len(buf)
    # attributes: 4 global + VAR_TYPE, VALIDMIN, VALIDMAX, VAR_NOTES; every ADR is 324 bytes
    assert n_attr == 8 and kinds.count(4) == 8 and all(s == 324 for _, s, k in recs if k == 4)
    assert n_z == 5 and kinds.count(8) == 5 and kinds.count(3) == 0
    sizes = {k: sorted({s for _, s, kk in recs if kk == k}) for k in set(kinds)}
    assert sizes[6] == [28 + 16 * 7] and sizes[11] == [28]                      # VXR with 7 entries, CPR with one parameter
    assert sizes[8] == [344, 344 + 16]                                         # zVDR: 344 + 8 per dimension (0 or 2 here)
    assert kinds.count(13) == 9 and kinds.count(7) == 4 and kinds.count(6) == 2 + 4
    # everything the scan found is reachable from the GDR's chains, and nothing else is (the reader lists what it walked)
    r = C.Reader(path)
    assert sorted(r.records) == recs
    # the chains: attribute numbers count up, entries name their attribute; variable numbers count up
    n, a = 0, adr
    while a:
        nxt, gr, scope, num, n_gr, max_gr, _, zh, n_ze, max_z = struct.unpack_from('>qqiiiiiqii', buf, a + 12)
        assert num == n and scope in (1, 2)
        assert (n_ze == 0 and max_z == -1 and zh == 0) if scope == 1 else (n_gr == 0 and max_gr == -1 and gr == 0)
        n, a = n + 1, nxt
    assert n == n_attr
    # values
    assert r.attrs['title'] == 'structure' and np.array_equal(r.attrs['numbers'], [1.5, 2.5]) and r.attrs['texts'] == ['a', 'bcd']
    assert r.attrs['when'] == datetime(2012, 1, 25, 9, 26, 55, 60000)
    assert r.times('Epoch') == [datetime(2012, 1, 25), datetime(2012, 1, 26)]
    assert np.array_equal(r['big'].data, big) and r['big'].compressed == 5 and r['big'].rec_vary and r['big'].dims == (5, 6)
    assert r['plain'].type == C.CDF_UINT2 and r['plain'].compressed is None and r['plain'].attrs['VALIDMAX'] == 65535
    assert r['scalar'].type == C.CDF_BYTE and not r['scalar'].rec_vary and r['scalar'].dims == () and r['scalar'][...] == 7
    assert r['scalar'].attrs['VAR_NOTES'] == ' '
    assert r['nrv'].type == C.CDF_FLOAT and r['nrv'].dims == (2, 3) and np.array_equal(r['nrv'][...], np.arange(6).reshape(2, 3))
    assert list(r['big'].attrs) == ['VAR_TYPE', 'VALIDMIN'] and list(r.vars) == ['Epoch', 'big', 'plain', 'scalar', 'nrv']
    # a compressed block is a gzip stream of one record (blocking factor 1)
    off = next(o for o, s, k in recs if k == 13)
    csize, = struct.unpack_from('>q', buf, off + 16)
    assert buf[off + 24:off + 26] == b'\x1f\x8b'
    assert gzip.decompress(buf[off + 24:off + 24 + csize]) == big[0].astype('<f8').tobytes()
    with pytest.raises(KeyError):
        w.new('big', 1)
    with pytest.raises(ValueError):
        w.new('none', np.zeros((0, 3)))


# what the reference's exporter creates (export/cdf.py:82-285), in its order: name -> (record variance, compressed)
ORDER = ['Epoch', 'lat', 'lon', 'lat_bounds', 'lon_bounds', 'altitude', 'mlat', 'mlt', 'mlat_bounds', 'mlt_bounds', 'mcrs',
         'img_red', 'img_green', 'img_blue', 'zenith_angle', 'camera_pos', 'crs']
IMG = ['VAR_TYPE', 'DEPEND_0', 'DEPEND_1', 'DEPEND_2', 'FIELDNAM', 'VALIDMIN', 'VALIDMAX', 'FILLVAL', 'UNITS']
_C = ['VAR_TYPE', 'DEPEND_0', 'DEPEND_1', 'DEPEND_2', 'UNITS', 'VALIDMIN', 'VALIDMAX', 'FIELDNAM', 'VAR_NOTES', 'crs']
_M = [k for k in _C if k != 'VAR_NOTES']
# attributes in the order the reference's exporter sets them, per variable
EXPORTER_ATTRS = {'Epoch': ['VAR_TYPE'], 'lat': _C + ['bounds'], 'lon': _C + ['bounds'], 'lat_bounds': _C, 'lon_bounds': _C,
                  'altitude': ['VAR_TYPE', 'UNITS', 'FIELDNAM', 'crs'], 'mlat': _C + ['bounds'], 'mlt': _M + ['bounds'],
                  'mlat_bounds': _C, 'mlt_bounds': _M,
                  'mcrs': ['VAR_TYPE', 'north_geomagnetic_pole_lat', 'north_geomagnetic_pole_lon', 'VAR_NOTES'], 'img': IMG,
                  'zenith_angle': ['VAR_TYPE', 'DEPEND_0', 'DEPEND_1', 'DEPEND_2', 'UNITS', 'VALIDMIN', 'VALIDMAX', 'FIELDNAM'],
                  'camera_pos': ['VAR_TYPE', 'DEPEND_0', 'UNITS', 'FIELDNAM', 'VAR_NOTES'],
                  'crs': ['VAR_TYPE', 'semi_major_axis', 'inverse_flattening', 'VAR_NOTES']}
COORD = ['VAR_TYPE', 'DEPEND_0', 'DEPEND_1', 'DEPEND_2', 'UNITS', 'VALIDMIN', 'VALIDMAX', 'FIELDNAM', 'VAR_NOTES', 'crs']


@pytest.mark.parametrize('tt2000', [True, False])
@pytest.mark.parametrize('case', CASES)
def test_exported_mapping(case, tt2000, tmp_path):
    from auromat_amd.export import _cdf3 as C
    from auromat_amd.export.cdf import write
    from auromat_amd.mapping.cdf import read_arrays
    z = load_golden('netcdf_case_%s.npz' % case)
    m = Mapping(z, {'Project': 'auromat', 'Calibrated': True}, case)
    opts = {}
    if case.endswith('_includeBounds'):
        opts['includeBounds'] = False
    if case.endswith('_includeMagCoords'):
        opts['includeMagCoords'] = False
    path = str(tmp_path / (case + '.cdf'))
    write(path, m, metadata={'Source_name': 'test'}, useTT2000=tt2000, **opts)
    buf, recs = scan(path)
    r = C.Reader(path)
    assert sorted(r.records) == recs
    # global attributes: the metadata, then the bounding box (reference cdf.py:62-80)
    assert list(r.attrs) == ['Project', 'Calibrated', 'Source_name', 'geospatial_lat_min', 'geospatial_lat_max', 'geospatial_lon_min',
                             'geospatial_lon_max', 'geospatial_lat_units', 'geospatial_lon_units']
    assert r.attrs['Calibrated'] == 1 and r.attrs['geospatial_lat_min'] == m.boundingBox.latSouth
    assert r.attrs['geospatial_lat_min'].dtype == np.float64 and r.attrs['geospatial_lon_units'] == 'degrees_east'
    want = [n for n in ORDER if ('bounds' not in n or opts.get('includeBounds', True))
            and (not n.startswith('m') or opts.get('includeMagCoords', True))]
    assert list(r.vars) == want
    assert r['Epoch'].type == (C.CDF_TIME_TT2000 if tt2000 else C.CDF_EPOCH) and r.times('Epoch') == [m.photoTime]
    h, w = m.img.shape[:2]
    for name in want:
        v = r[name]
        grid = name in ('lat', 'lon', 'mlat', 'mlt', 'zenith_angle') or name.startswith('img') or name.endswith('_bounds')
        assert v.rec_vary == (name not in ('altitude', 'mcrs', 'crs')), name
        assert (v.compressed == 5) == grid, name
        assert v.dims == (((h + 1, w + 1) if name.endswith('_bounds') else (h, w)) if grid else ((3,) if name == 'camera_pos' else ())), name
        assert v.attrs['VAR_TYPE'] == ('data' if grid else 'support_data'), name
    for name in ('lat', 'lon', 'mlat', 'mlt'):
        if name not in want:
            continue
        keys = [k for k in COORD if not (name == 'mlt' and k == 'VAR_NOTES')]
        # (a CDF numbers its attributes file-wide, in order of first use: a variable lists its own in that order)
        assert set(r[name].attrs) == set(keys + (['bounds'] if opts.get('includeBounds', True) else [])), name
        assert r[name].type == C.CDF_DOUBLE and r[name].attrs['VALIDMIN'].dtype == np.float64
    assert r['lat'].attrs['DEPEND_1'] == 'y_pixel' and r['lat'].attrs['VAR_NOTES'] == 'Geodetic latitude' and r['lat'].attrs['crs'] == 'crs'
    if 'mlt' in want:
        assert r['mlt'].attrs['DEPEND_1'] == 'y_center' and r['mlt'].attrs['UNITS'] == 'hours' and r['mlt'].attrs['VALIDMAX'] == 24.0
        assert r['mlat'].attrs['VAR_NOTES'] == ' ' and r['mlat'].attrs['crs'] == 'mcrs'
        assert set(r['mcrs'].attrs) == {'VAR_TYPE', 'north_geomagnetic_pole_lat', 'north_geomagnetic_pole_lon', 'VAR_NOTES'}
        assert np.array_equal(r['mlat'][0], m.mLatMltCenter[0].filled(np.nan), equal_nan=True)
        if 'mlt_bounds' in want:
            assert r['mlt_bounds'].attrs['DEPEND_2'] == 'x_corner' and r['mlt'].attrs['bounds'] == 'mlt_bounds'
            assert np.array_equal(r['mlt_bounds'][0], m.mLatMlt[1].filled(np.nan), equal_nan=True)
    assert r['altitude'][...] == m.altitude * 1000 and r['altitude'].type == C.CDF_DOUBLE and r['altitude'].attrs['UNITS'] == 'meters'
    assert set(r['crs'].attrs) == {'VAR_TYPE', 'semi_major_axis', 'inverse_flattening', 'VAR_NOTES'}
    assert r['crs'].attrs['inverse_flattening'] == 298.257223563
    assert r['zenith_angle'].type == C.CDF_FLOAT
    assert np.array_equal(r['zenith_angle'][0], 90 - m.elevation.filled(np.nan).astype(np.float32), equal_nan=True)
    assert np.array_equal(r['camera_pos'][0], m.cameraPosGCRS) and r['camera_pos'].attrs['VAR_NOTES'] == 'Axis order: xyz'
    # the image: the next wider signed type with a fill value where something is masked (reference cdf.py:216-252)
    masked = bool(ma.getmaskarray(m.img).any())
    wide = {np.dtype('uint8'): (np.int16, C.CDF_INT2), np.dtype('uint16'): (np.int32, C.CDF_INT4)}[m.img.dtype]
    red = r['img_red']
    assert set(red.attrs) == set(['VAR_TYPE', 'DEPEND_0', 'DEPEND_1', 'DEPEND_2', 'FIELDNAM', 'VALIDMIN', 'VALIDMAX'] +
                                 (['FILLVAL'] if masked else []) + ['UNITS'])
    # the attribute numbers themselves: order of first use over the file (the ADR chain)
    first_use = []
    for name in want:
        for k in EXPORTER_ATTRS[name if not name.startswith('img') else 'img']:
            if k not in first_use and not (k == 'bounds' and not opts.get('includeBounds', True)) and not (k == 'FILLVAL' and not masked):
                first_use.append(k)
    assert r.attr_order[len(r.attrs):] == first_use
    assert red.attrs['VALIDMAX'] == np.iinfo(m.img.dtype).max and red.attrs['FIELDNAM'] == ' '
    if masked:
        assert red.type == wide[1] and red.attrs['FILLVAL'] == np.iinfo(wide[0]).min and red.attrs['FILLVAL'].dtype == wide[0]
        assert np.array_equal(red[0], m.img[:, :, 0].astype(wide[0]).filled(np.iinfo(wide[0]).min))
    # ... and back as a mapping (reference mapping/cdf.py:97-134)
    if 'lat_bounds' in want:
        a = read_arrays(path)
        for key, ref in (('lats', m.lats), ('lons', m.lons), ('latsCenter', m.latsCenter), ('lonsCenter', m.lonsCenter)):
            assert np.array_equal(a[key].filled(np.nan), ref.filled(np.nan), equal_nan=True), key
        assert a['img'].dtype == m.img.dtype and np.array_equal(a['img'].filled(0), m.img.filled(0))
        assert np.array_equal(ma.getmaskarray(a['img']), ma.getmaskarray(m.img))
        assert a['altitude'] == m.altitude and a['photoTime'] == m.photoTime and a['metadata']['Project'] == 'auromat'


def test_exporter_options(tmp_path):
    from auromat_amd.export import _cdf3 as C
    from auromat_amd.export.cdf import write
    from auromat_amd.mapping.cdf import CDFMappingProvider, read_arrays
    z = load_golden('netcdf_case_resampled.npz')
    m = Mapping(z, {}, 'x')
    # --without-geo: a CDF may leave the geodetic coordinates out (reference cdf.py:91-144)
    p = str(tmp_path / 'nogeo.cdf')
    write(p, m, includeGeoCoords=False, compress=False)
    r = C.Reader(p)
    assert 'lat' not in r and 'lon_bounds' not in r and 'mlat' in r and all(v.compressed is None for v in r.vars.values())
    # an image without masked pixels keeps its type and gets no FILLVAL (the reference's line 230 raises here: see cdf.py's header)
    m.img = ma.masked_array(m.img.filled(7))
    p = str(tmp_path / 'unmasked.cdf')
    write(p, m)
    r = C.Reader(p)
    assert r['img_red'].type == C.CDF_UINT2 and 'FILLVAL' not in r['img_red'].attrs
    assert np.array_equal(read_arrays(p)['img'].filled(0), m.img.filled(0))
    # one channel -> 'img'
    m.img = m.img[:, :, :1]
    p1 = str(tmp_path / 'grey.cdf')
    write(p1, m)
    assert 'img' in C.Reader(p1) and read_arrays(p1)['img'].shape == m.img.shape
    m.img = ma.masked_array(np.zeros(m.img.shape[:2] + (2,), np.uint16))
    with pytest.raises(NotImplementedError):
        write(str(tmp_path / 'two.cdf'), m)
    m.img = ma.masked_array(np.zeros(m.img.shape[:2] + (3,), np.float32), mask=True)
    with pytest.raises(NotImplementedError):
        write(str(tmp_path / 'float.cdf'), m)
    with pytest.raises(TypeError):
        write(str(tmp_path / 'meta.cdf'), Mapping(z, {'bad': {'a': 1}}, 'x'))
    # the provider finds files by date (reference mapping/cdf.py:19-77)
    prov = CDFMappingProvider([p, ])
    t = Mapping(z, {}, 'x').photoTime
    assert len(prov) == 1 and prov.range == (t, t) and prov.contains(t + timedelta(seconds=2)) and not prov.contains(t + timedelta(seconds=9))
    with pytest.raises(ValueError):
        CDFMappingProvider([p, p1])


def test_large_records_are_one_gzip_member_made_in_blocks(tmp_path):
    """A record of 4 MB or more is deflated in 1-MB blocks side by side by the writers' helper (pigz's layout: one gzip member,
    every block ending on a byte boundary): any inflate reads it as one stream."""
    from auromat_amd.export import _cdf3 as C, _io
    if _io.lib() is None:
        pytest.skip('libauromat_io.so not built / switched off')
    rs = np.random.RandomState(1)
    a = np.cumsum(rs.rand(1, 700, 1000), axis=2)                     # 5.6 MB
    a[0, :100] = np.nan
    w = C.Writer()
    w.new('big', a, compress=C.GZIP_COMPRESSION)
    w.new('small', a[:, :10], compress=C.GZIP_COMPRESSION)
    path = str(tmp_path / 'big.cdf')
    w.write(path)
    buf, recs = scan(path)
    blocks = [(o, s) for o, s, k in recs if k == 13]
    assert len(blocks) == 2
    o, s = blocks[0]
    csize, = struct.unpack_from('>q', buf, o + 16)
    assert gzip.decompress(buf[o + 24:o + 24 + csize]) == a[0].astype('<f8').tobytes()
    single = len(gzip.compress(a[0].tobytes(), 5))
    assert csize < single * 1.01                                         # (no history across the cuts: a few tenths of a percent)
    r = C.Reader(path)
    assert np.array_equal(r['big'].data, a, equal_nan=True) and np.array_equal(r['small'].data, a[:, :10], equal_nan=True)
    for n in (0, 1, (1 << 16) + 3):
        b = rs.randint(0, 255, n).astype(np.uint8)
        assert gzip.decompress(_io.gzip_parallel(b, 5, 3, 1 << 16)) == b.tobytes()


def test_attribute_values_are_checked_when_they_are_assigned():
    """ a variable-scope attribute is ONE entry, an entry is never empty, a name lives in one scope — and all of it
    is refused at the assignment, not after the variables have been compressed."""
    from auromat_amd.export import _cdf3 as C
    w = C.Writer()
    v = w.new('a', np.arange(4.0))
    w.attrs['Title'] = ['two', 'entries']                    # global: one entry per text, as pycdf
    v.attrs['UNITS'] = 'deg'
    with pytest.raises(ValueError):
        v.attrs['Notes'] = ['first', 'second']               # (was: the second text dropped silently)
    with pytest.raises(ValueError):
        v.attrs['Empty'] = []
    with pytest.raises(ValueError):
        w.attrs['Empty'] = np.zeros(0)
    with pytest.raises(ValueError):
        w.attrs['UNITS'] = 'x'                               # already of variable scope
    with pytest.raises(ValueError):
        v.attrs['Title'] = 'x'                               # already global
    with pytest.raises(ValueError):
        v.attrs['n' * 256] = 1
    with pytest.raises(TypeError):
        v.attrs['Obj'] = object()
    assert list(v.attrs) == ['UNITS'] and list(w.attrs) == ['Title']
    assert C.infer([1, 2, 3])[1] == 3


def test_large_variables_compress_early_to_the_same_bytes(tmp_path, monkeypatch):
    """ Arrays of 4 MiB or more start to compress when they are handed over (`Writer.new` with a pool; `Variable.set`
    of the netCDF-4 writer) — the files must be the bytes of the late path, with the native helper and without it."""
    from concurrent.futures import ThreadPoolExecutor
    from auromat_amd.export import _cdf3 as C, _io, _nc4
    rs = np.random.RandomState(3)
    big = np.cumsum(rs.rand(3, 420, 500), axis=2)                      # three records of 1.7 MB: 5 MB, early path, zlib per record
    huge = np.cumsum(rs.rand(1, 800, 700), axis=2)                     # one record of 4.5 MB: the helper's blocks
    small = big[:, :7]

    def cdf(path, pool):
        w = C.Writer(pool=pool)
        w.attrs['Title'] = 'early / late'
        for name, a in (('big', big), ('huge', huge), ('small', small)):
            w.new(name, a, compress=C.GZIP_COMPRESSION).attrs['UNITS'] = 'deg'
        assert (w['big'].blocks is not None) == (pool is not None) and w['small'].blocks is None
        w.write(path)
        return open(path, 'rb').read()

    def nc(path):
        w = _nc4.Writer()
        w.create_dimension('y', 1300)
        w.create_dimension('x', 500)
        v = w.create_variable('a', 'f8', ('y', 'x'), zlib=True, chunksizes=(1, 500))
        v.set(np.cumsum(rs.rand(1300, 500), axis=1))                     # 5.2 MB
        early = v.early is not None
        w.create_variable('b', 'i2', ('y', 'x'), zlib=True, chunksizes=(1, 500), fill_value=-1).set(rs.randint(0, 99, (1300, 500)))
        w.write(path)
        return early, open(path, 'rb').read()

    with ThreadPoolExecutor(max_workers=3) as pool:
        with_pool = cdf(str(tmp_path / 'a.cdf'), pool)
    without_pool = cdf(str(tmp_path / 'b.cdf'), None)
    assert with_pool == without_pool
    r = C.Reader(str(tmp_path / 'a.cdf'))
    assert np.array_equal(r['big'].data, big) and np.array_equal(r['huge'].data, huge)
    rs = np.random.RandomState(4)
    early, nc_with = nc(str(tmp_path / 'a.nc'))
    assert early
    helper = _io.lib() is not None
    monkeypatch.setattr(_io, '_lib', [None])                           # AMT_IO_HELPER=0
    assert _io.lib() is None
    python_only = cdf(str(tmp_path / 'c.cdf'), None)
    # (the helper cuts a large record into blocks — another, equally valid gzip member; everything else is byte for byte)
    r2 = C.Reader(str(tmp_path / 'c.cdf'))
    assert np.array_equal(r2['huge'].data, huge) and np.array_equal(r2['big'].data, big)
    assert helper or python_only == with_pool
    rs = np.random.RandomState(4)
    _, nc_without = nc(str(tmp_path / 'b.nc'))
    assert nc_with == nc_without


def test_native_jobs_share_the_thread_budget():
    from auromat_amd.export import _io
    with _io._share(16) as a:
        assert a == 16
        with _io._share(16) as b:
            assert b == 8
            with _io._share(3) as c:
                assert c == 1
        with _io._share(16) as d:
            assert d == 8
    assert _io._active[0] == 0
