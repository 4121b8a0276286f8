"""
Minimal writer / reader of the netCDF classic format with 64-bit offsets (CDF-2), NumPy only.

The reference writes its files with the netCDF4 library (HDF5 container, zlib-compressed, chunked; here:
:mod:`auromat_amd.export._nc4`, the default of the exporter).  Nothing the reference stores needs netCDF-4 features: every
variable is byte / int / float / double and every attribute a string or a small numeric array.  The classic format holds the
same dimensions, variables, attributes and ``_FillValue`` s, is read by netCDF4 / xarray / GDAL / ncdump like any other
netCDF file (so the reference's own ``NetCDFMapping`` reader opens it), and has no compression and no chunking.

Format: https://docs.unidata.ucar.edu/netcdf-c/current/file_format_specifications.html (classic, CDF-2).
"""
import struct
from collections import OrderedDict

import numpy as np

NC_BYTE, NC_CHAR, NC_SHORT, NC_INT, NC_FLOAT, NC_DOUBLE = 1, 2, 3, 4, 5, 6
NC_DIMENSION, NC_VARIABLE, NC_ATTRIBUTE = 0x0A, 0x0B, 0x0C
_TYPES = {NC_BYTE: np.dtype('>i1'), NC_SHORT: np.dtype('>i2'), NC_INT: np.dtype('>i4'), NC_FLOAT: np.dtype('>f4'),
          NC_DOUBLE: np.dtype('>f8')}
# the classic format has no unsigned and no 64-bit integer types: the next larger signed type takes them
_CODES = {'i1': NC_BYTE, 'u1': NC_SHORT, 'i2': NC_SHORT, 'u2': NC_INT, 'i4': NC_INT, 'f4': NC_FLOAT, 'f8': NC_DOUBLE,
          'b1': NC_BYTE}


def nc_type(dtype):
    dtype = np.dtype(dtype)
    key = dtype.kind + str(dtype.itemsize)
    if key not in _CODES:
        raise TypeError('no netCDF classic type for ' + str(dtype))
    return _CODES[key]


def _pad(n):
    return (4 - n % 4) % 4


def _name(s):
    b = s.encode('utf-8')
    return struct.pack('>i', len(b)) + b + b'\0' * _pad(len(b))


def _attr_values(value):
    """-> (nc_type, nelems, padded bytes)"""
    if isinstance(value, bytes):
        value = value.decode('utf-8')
    if isinstance(value, str):
        b = value.encode('utf-8')
        return NC_CHAR, len(b), b + b'\0' * _pad(len(b))
    a = np.atleast_1d(np.asarray(value))
    if a.dtype.kind == 'b':
        a = a.astype(np.int8)
    if a.dtype.kind in 'iu' and a.dtype.itemsize == 8:
        a = a.astype(np.int32) if np.all(np.abs(a) < 2 ** 31) else a.astype(np.float64)
    t = nc_type(a.dtype)
    b = a.astype(_TYPES[t]).tobytes()
    return t, a.size, b + b'\0' * _pad(len(b))


def _attr_list(attrs):
    if not attrs:
        return struct.pack('>ii', 0, 0)
    out = [struct.pack('>ii', NC_ATTRIBUTE, len(attrs))]
    for k, v in attrs.items():
        t, n, b = _attr_values(v)
        out += [_name(k), struct.pack('>ii', t, n), b]
    return b''.join(out)


class Variable(object):
    def __init__(self, name, dtype, dims, fill_value=None):
        self.name, self.dims = name, tuple(dims)
        self.type = nc_type(dtype)
        self.dtype = _TYPES[self.type]
        self.attrs = OrderedDict()
        if fill_value is not None:
            self.attrs['_FillValue'] = np.asarray(fill_value, dtype=self.dtype.newbyteorder('='))
        self.data = None

    def set(self, data):
        self.data = np.asarray(data)


class Writer(object):
    """Collects dimensions, global attributes and variables, then writes the file in one go."""

    def __init__(self):
        self.dims = OrderedDict()
        self.attrs = OrderedDict()
        self.vars = OrderedDict()

    def create_dimension(self, name, size):
        if int(size) <= 0:
            # (a length of 0 in the header marks the record (unlimited) dimension for classic readers)
            raise ValueError('dimension %r of size %r: fixed dimensions need a positive size' % (name, size))
        self.dims[name] = int(size)

    def create_variable(self, name, dtype, dims=(), fill_value=None, zlib=False, chunksizes=None):
        # (zlib, chunksizes: options of the netCDF-4 writer, :mod:`auromat_amd.export._nc4`; the classic format has neither)
        if isinstance(dims, str):
            dims = (dims,)
        for d in dims:
            if d not in self.dims:
                raise KeyError('unknown dimension ' + d)
        v = self.vars[name] = Variable(name, dtype, dims, fill_value)
        return v

    def write(self, path):
        dim_ids = {d: i for i, d in enumerate(self.dims)}
        head = [b'CDF\x02', struct.pack('>i', 0)]
        if self.dims:
            head.append(struct.pack('>ii', NC_DIMENSION, len(self.dims)))
            for d, n in self.dims.items():
                head += [_name(d), struct.pack('>i', n)]
        else:
            head.append(struct.pack('>ii', 0, 0))
        head.append(_attr_list(self.attrs))
        # variables: header entries need the data offsets, which need the header size: two passes
        entries, sizes, blobs = [], [], []
        for v in self.vars.values():
            shape = tuple(self.dims[d] for d in v.dims)
            data = np.zeros(shape, v.dtype) if v.data is None else v.data
            if data.shape != shape:
                data = np.broadcast_to(data, shape) if data.size == 1 else data.reshape(shape)
            blob = np.ascontiguousarray(data, dtype=v.dtype).tobytes()
            blob += b'\0' * _pad(len(blob))
            blobs.append(blob)
            sizes.append(len(blob))
            entries.append(b''.join([_name(v.name), struct.pack('>i', len(v.dims))] +
                                    [struct.pack('>i', dim_ids[d]) for d in v.dims] +
                                    [_attr_list(v.attrs), struct.pack('>i', v.type),
                                     struct.pack('>I', min(len(blob), 2 ** 32 - 1))]))
        var_head = struct.pack('>ii', NC_VARIABLE, len(entries)) if entries else struct.pack('>ii', 0, 0)
        offset = sum(len(b) for b in head) + len(var_head) + sum(len(e) + 8 for e in entries)
        with open(path, 'wb') as fp:
            for b in head:
                fp.write(b)
            fp.write(var_head)
            for e, n in zip(entries, sizes):
                fp.write(e)
                fp.write(struct.pack('>q', offset))
                offset += n
            for b in blobs:
                fp.write(b)


class _Parser(object):
    def __init__(self, buf):
        self.buf, self.pos = buf, 0

    def int(self):
        v, = struct.unpack_from('>i', self.buf, self.pos)
        self.pos += 4
        return v

    def int64(self):
        v, = struct.unpack_from('>q', self.buf, self.pos)
        self.pos += 8
        return v

    def name(self):
        n = self.int()
        s = bytes(self.buf[self.pos:self.pos + n]).decode('utf-8')
        self.pos += n + _pad(n)
        return s

    def attrs(self):
        tag, n = self.int(), self.int()
        out = OrderedDict()
        if tag == 0:
            return out
        assert tag == NC_ATTRIBUTE
        for _ in range(n):
            k = self.name()
            t, m = self.int(), self.int()
            if t == NC_CHAR:
                out[k] = bytes(self.buf[self.pos:self.pos + m]).decode('utf-8')
                self.pos += m + _pad(m)
            else:
                dt = _TYPES[t]
                a = np.frombuffer(self.buf, dt, m, self.pos).astype(dt.newbyteorder('='))
                self.pos += m * dt.itemsize + _pad(m * dt.itemsize)
                out[k] = a[0] if m == 1 else a
        return out


class File(object):
    """A classic netCDF file read into memory: ``dims`` (name -> size), ``attrs``, ``vars`` (name -> :class:`ReadVariable`)."""

    def __init__(self, path):
        with open(path, 'rb') as fp:
            buf = memoryview(fp.read())
        assert bytes(buf[:3]) == b'CDF' and buf[3] in (1, 2), 'not a netCDF classic file'
        wide = buf[3] == 2
        p = _Parser(buf)
        p.pos = 4
        p.int()                                      # numrecs
        self.dims = OrderedDict()
        tag, n = p.int(), p.int()
        if tag == NC_DIMENSION:
            for _ in range(n):
                name = p.name()
                self.dims[name] = p.int()
        self.attrs = p.attrs()
        self.vars = OrderedDict()
        tag, n = p.int(), p.int()
        dim_names = list(self.dims)
        if tag == NC_VARIABLE:
            for _ in range(n):
                name = p.name()
                nd = p.int()
                dims = tuple(dim_names[p.int()] for _ in range(nd))
                attrs = p.attrs()
                t = p.int()
                p.int()                              # vsize
                begin = p.int64() if wide else p.int()
                shape = tuple(self.dims[d] for d in dims)
                count = int(np.prod(shape)) if shape else 1
                data = np.frombuffer(buf, _TYPES[t], count, begin).reshape(shape).astype(_TYPES[t].newbyteorder('='))
                self.vars[name] = ReadVariable(name, dims, attrs, data)


class ReadVariable(object):
    def __init__(self, name, dims, attrs, data):
        self.name, self.dims, self.attrs, self.data = name, dims, attrs, data

    def __getattr__(self, k):
        try:
            return self.attrs[k]
        except KeyError:
            raise AttributeError(k)
