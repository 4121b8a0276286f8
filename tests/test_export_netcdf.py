"""
netCDF export (auromat_amd/export/netcdf.py) against the layout of the REAL reference's exporter
(auromat/export/netcdf.py:24-386): tests/golden/netcdf_layout_*.json lists every dimension, variable (type, dimensions,
fill value) and attribute the reference creates, in order, recorded from its own code by oracle/make_golden.py
(netcdf_layout) for an unresampled camera mapping and a resampled plate carree one, with and without pixel bounds and
MLat/MLT coordinates; netcdf_case_*.npz holds the mapping that went in and the data that came out.  The files written
here are read back with the package's own classic-format reader and compared item by item — the round trip the
reference's export_netcdf_test.py:72-89 makes.  CPU only (the exporter is host code).
"""
import json
import os
from datetime import datetime

import numpy as np
import numpy.ma as ma
import pytest

from conftest import GOLDEN, load_golden

CASES = [t + o for t in ('unresampled', 'resampled') for o in ('', '_includeBounds', '_includeMagCoords')]
# what the classic format stores for the types the netCDF-4 library accepts as they are
CLASSIC = {'uint8': 'int16', 'bool': 'int8', 'int64': 'int32', 'int': 'int32', 'float': 'float64'}


class Box(object):
    def __init__(self, b):
        self.latSouth, self.lonWest, self.latNorth, self.lonEast = [float(v) for v in b]


class Mapping(object):
    """What the exporter reads of a BaseMapping, rebuilt from the arrays of the reference's mapping."""

    def __init__(self, z, metadata, identifier):
        def masked(data, nan_filled):
            return ma.masked_array(z[data], mask=np.isnan(z[nan_filled]))
        self.lats, self.lons = masked('lats_data', 'lats'), masked('lons_data', 'lons')
        self.latsCenter, self.lonsCenter = masked('lats_c_data', 'lats_c'), masked('lons_c_data', 'lons_c')
        self.mLatMlt = (masked('mlat_data', 'mlat'), masked('mlt_data', 'mlt'))
        self.mLatMltCenter = (masked('mlat_c_data', 'mlat_c'), masked('mlt_c_data', 'mlt_c'))
        self.elevation = ma.masked_invalid(z['elev'])
        self.img = ma.masked_array(z['img'], mask=z['img_mask'])
        self.boundingBox = Box(z['bbox'])
        self.photoTime = datetime.strptime(str(z['time_iso']), '%Y-%m-%dT%H:%M:%S.%f')
        self.altitude = float(z['altitude'])
        self.cameraPosGCRS = z['cam']
        self.metadata = metadata
        self.identifier = identifier


def value_of(rec, classic=True):
    v = rec['value']
    # (the classic format has no unsigned types: the next larger signed one; netCDF-4 keeps the reference's)
    return v if rec['dtype'] == 'str' else np.asarray(v, dtype=CLASSIC.get(rec['dtype'], rec['dtype']) if classic else rec['dtype'])


def same_attr(got, rec, what, classic=True):
    want = value_of(rec, classic)
    if isinstance(want, str):
        assert got == want, what
    else:
        got = np.asarray(got)
        assert got.dtype == want.dtype, (what, got.dtype, want.dtype)
        assert np.array_equal(np.atleast_1d(got), np.atleast_1d(want), equal_nan=True), what


@pytest.mark.parametrize('fmt', ['NETCDF4', 'NETCDF3_64BIT'])
@pytest.mark.parametrize('case', CASES)
def test_written_file_has_the_references_layout_and_data(case, fmt, tmp_path):
    from auromat_amd.export import _nc4
    from auromat_amd.export.netcdf import write
    layout = json.load(open(os.path.join(GOLDEN, 'netcdf_layout_%s.json' % case)))
    z = load_golden('netcdf_case_%s.npz' % case)
    meta = {'Project': 'auromat', 'Calibrated': True} if case.startswith('unresampled') else {'Project': 'auromat'}
    m = Mapping(z, meta, case)
    opts = {}
    if case.endswith('_includeBounds'):
        opts['includeBounds'] = False
    if case.endswith('_includeMagCoords'):
        opts['includeMagCoords'] = False
    path = str(tmp_path / (case + '.nc'))
    write(path, m, metadata={'Source_name': 'test'}, format=fmt, **opts)
    f = _nc4.open_file(path)
    classic = fmt != 'NETCDF4'
    # dimensions: names, sizes, order
    assert [[k, v] for k, v in f.dims.items()] == layout['dims']
    # global attributes: names, order, values, types
    assert list(f.attrs) == [k for k, _ in layout['attrs']]
    for k, rec in layout['attrs']:
        same_attr(f.attrs[k], rec, k, classic)
    # variables: names and order, types, dimensions, _FillValue, attributes, data
    # (an HDF5 group of the old kind lists its members by name; the classic format keeps the order of creation)
    assert (list(f.vars) if classic else sorted(f.vars)) == (lambda names: names if classic else sorted(names))([v['name'] for v in layout['vars']])
    for rec in layout['vars']:
        v = f.vars[rec['name']]
        assert v.data.dtype == np.dtype(rec['dtype']), rec['name']
        assert list(v.dims) == rec['dims'], rec['name']
        names = [k for k, _ in rec['attrs']]
        if rec['fill_value'] is not None:
            assert v.attrs['_FillValue'] == rec['fill_value'] and v.attrs['_FillValue'].dtype == v.data.dtype
            names = ['_FillValue'] + names
        assert list(v.attrs) == names, rec['name']
        for k, a in rec['attrs']:
            same_attr(v.attrs[k], a, rec['name'] + '.' + k, classic)
        if 'var_' + rec['name'] not in z.files:
            continue                                        # crs / mcrs hold no data
        want = z['var_' + rec['name']]
        assert np.array_equal(v.data, np.asarray(want, dtype=v.data.dtype).reshape(v.data.shape), equal_nan=True), rec['name']
    assert layout['format'] == 'NETCDF4'                    # the reference's container
    assert open(path, 'rb').read(8) == (b'CDF\x02' + open(path, 'rb').read(8)[4:] if classic else b'\x89HDF\r\n\x1a\n')


def test_rejects_what_the_reference_rejects(tmp_path):
    from auromat_amd.export.netcdf import write
    m = Mapping(load_golden('netcdf_case_resampled.npz'), {}, 'x')
    with pytest.raises(ValueError):
        write(str(tmp_path / 'a.nc'), m, includeGeoCoords=False)
    m.img = ma.masked_array(m.img.data.astype(np.float32), mask=m.img.mask)
    with pytest.raises(NotImplementedError):
        write(str(tmp_path / 'b.nc'), m)


def test_classic_format_primitives(tmp_path):
    """Names and values that need padding, scalars, empty attribute lists, every type; scipy reads the file as well."""
    from auromat_amd.export import _nc3
    w = _nc3.Writer()
    w.create_dimension('a', 3)
    w.create_dimension('bb', 5)
    w.attrs['title'] = 'x'
    w.attrs['n'] = np.int32(7)
    v = w.create_variable('v1', np.int16, ('a', 'bb'), fill_value=-32768)
    v.attrs['units'] = 'unitless'
    v.set(np.arange(15).reshape(3, 5))
    w.create_variable('s', np.float64).set(2.5)
    w.create_variable('b', np.int8, ('a',)).set([1, -2, 3])
    w.create_variable('f', np.float32, ('bb',)).set(np.linspace(0, 1, 5))
    path = str(tmp_path / 'p.nc')
    w.write(path)
    f = _nc3.File(path)
    assert f.attrs['title'] == 'x' and f.attrs['n'] == 7
    assert np.array_equal(f.vars['v1'].data, np.arange(15).reshape(3, 5)) and f.vars['v1'].attrs['_FillValue'] == -32768
    assert f.vars['s'].data == 2.5 and np.array_equal(f.vars['b'].data, [1, -2, 3])
    assert np.array_equal(f.vars['f'].data, np.linspace(0, 1, 5).astype(np.float32))
    sio = pytest.importorskip('scipy.io')
    with sio.netcdf_file(path, 'r', mmap=False) as g:
        assert g.title == b'x' and g.dimensions == {'a': 3, 'bb': 5}
        assert np.array_equal(g.variables['v1'][:], np.arange(15).reshape(3, 5)) and g.variables['v1'].units == b'unitless'
        assert g.variables['s'][()] == 2.5 and np.array_equal(g.variables['b'][:], [1, -2, 3])
