"""
Minimal writer of netCDF-4 files (the HDF5 container the reference writes: ``Dataset(path, 'w', format='NETCDF4')`` with
``zlib=True`` and ``chunksizes=(1, w)``, reference export/netcdf.py:48,128-326), NumPy + zlib only.

Neither the netCDF4 library nor h5py is a dependency of this package, so the file is laid out here, byte by byte, after the
HDF5 file format specification (version 1.1 structures throughout, which every HDF5 library since 1.6 reads: superblock
version 0, a symbol-table root group — local heap, one group B-tree node, one symbol-table node —, version-1 object headers,
version-3 data layouts, contiguous or chunked with a version-1 chunk B-tree, the shuffle and deflate filters) and after the
netCDF-4 file format specification ("NetCDF-4/HDF5 File Format", netcdf-c docs/file_format_specifications.md): every netCDF
dimension is an HDF5 dimension scale (``CLASS``, ``NAME``, ``_Netcdf4Dimid``, ``REFERENCE_LIST``) and every variable lists its
dimensions in ``DIMENSION_LIST`` (variable-length sequences of object references, kept in a global heap collection);
text attributes are fixed-length NUL-terminated strings, ``_FillValue`` is both an attribute and the dataset's fill value.

Same interface as :mod:`auromat_amd.export._nc3` (create_dimension / create_variable / attrs / write) plus the two per-variable
options the reference uses: ``zlib`` (deflate level 4 behind the byte shuffle, netCDF4-python's defaults) and ``chunksizes``.
Checked by reading the files back with HDF5 1.10 (h5py, h5dump, h5ls) where that exists (tests/test_export_netcdf4.py).
"""
import os
import struct
import zlib as _zlib
from collections import OrderedDict

import numpy as np

from . import _io

UNDEF = 0xFFFFFFFFFFFFFFFF
_GROUP_K = 16            # group B-tree K (internal) — one node with one child is all this writer makes
_CHUNK_K = 32            # chunk B-tree K: the library's default for superblock version 0 (which has no field for it)
_DEFLATE_LEVEL = 4       # netCDF4-python's default complevel
# threads that compress the chunks of a large array (deflate releases the interpreter lock); AMT_IO_THREADS / AMT_NC4_THREADS
# override.  Native jobs that run at the same time share this number (_io._share): four writer threads with several variables
# each stay near it instead of multiplying it.
def _io_threads():
    from .._native import host_threads
    return int(os.environ.get('AMT_IO_THREADS', os.environ.get('AMT_NC4_THREADS', '0'))) or host_threads(16)


_THREADS = max(1, min(16, _io_threads()))               # (this rank's share of the host's cores: _native.host_threads)
_ROWS_PER_TASK = 32      # chunks of one row: rows a task shuffles at once (one NumPy copy) and then deflates one by one
_POOL = []


def _pool():
    """The threads that deflate (one pool per process, made on first use: a resampled grid has a few hundred one-row chunks
    per variable and twenty variables per file — a pool per variable cost more than it gave)."""
    if not _POOL:
        from concurrent.futures import ThreadPoolExecutor
        _POOL.append(ThreadPoolExecutor(max_workers=_THREADS))
    return _POOL[0]

NOT_A_VARIABLE = 'This is a netCDF dimension but not a netCDF variable.%10d'
# the library's default fill values (netcdf.h NC_FILL_*): the HDF5 fill value of a variable without a _FillValue
_NC_FILL = {'i1': -127, 'u1': 255, 'i2': -32767, 'u2': 65535, 'i4': -2147483647, 'u4': 4294967295,
            'i8': -9223372036854775806, 'u8': 18446744073709551614, 'f4': 9.9692099683868690e+36, 'f8': 9.9692099683868690e+36}


def default_fill(dtype):
    dtype = np.dtype(dtype)
    return np.asarray(_NC_FILL[dtype.kind + str(dtype.itemsize)], dtype=dtype)


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


# ---- datatype messages ------------------------------------------------------------------------------------------------

def _dt_fixed(size, signed):
    return struct.pack('<BBBBI', 0x10, 0x08 if signed else 0x00, 0, 0, size) + struct.pack('<HH', 0, 8 * size)


def _dt_float(size):
    if size == 4:
        sign, props = 31, struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)
    else:
        sign, props = 63, struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)
    return struct.pack('<BBBBI', 0x11, 0x20, sign, 0, size) + props


def _dt_string(size):
    return struct.pack('<BBBBI', 0x13, 0x00, 0, 0, size)          # NUL-terminated, ASCII


_DT_REF = struct.pack('<BBBBI', 0x17, 0x00, 0, 0, 8)              # object reference
_DT_VLEN_REF = struct.pack('<BBBBI', 0x19, 0x00, 0, 0, 16) + _DT_REF


def _dt_reference_list():
    """compound {dataset: object reference @0, dimension: int32 @8}, 16 bytes (H5DS's ds_list_t), version 1"""
    def member(name, offset, dt):
        return _pad8(name.encode() + b'\0') + struct.pack('<IB3xII4I', offset, 0, 0, 0, 0, 0, 0, 0) + dt
    return struct.pack('<BBBBI', 0x16, 2, 0, 0, 16) + member('dataset', 0, _DT_REF) + member('dimension', 8, _dt_fixed(4, True))


def datatype_message(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind in 'iu':
        return _dt_fixed(dtype.itemsize, dtype.kind == 'i')
    if dtype.kind == 'f' and dtype.itemsize in (4, 8):
        return _dt_float(dtype.itemsize)
    raise TypeError('no netCDF-4 type for ' + str(dtype))


def _dataspace(shape):
    """version 1; () is the scalar dataspace"""
    return struct.pack('<BBBB4x', 1, len(shape), 0, 0) + b''.join(struct.pack('<Q', n) for n in shape)


def _message(mtype, data, flags=0):
    data = _pad8(data)
    if len(data) > 0xFFF8:
        # the size field of a version-1 header message has 16 bits (a larger attribute would need a dense attribute
        # storage or a continuation per message, which this writer does not lay out)
        raise ValueError('header message of %d bytes: an attribute must stay below 64 KiB' % len(data))
    return struct.pack('<HHB3x', mtype, len(data), flags) + data


def _attribute(name, dt, shape, data):
    """attribute message, version 1"""
    nm = name.encode('utf-8') + b'\0'
    ds = _dataspace(shape)
    body = struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data
    return _message(0x000C, body)


def _attr_message(name, value):
    if isinstance(value, bytes):
        value = value.decode('utf-8')
    if isinstance(value, str):
        b = value.encode('utf-8') + b'\0'
        return _attribute(name, _dt_string(len(b)), (), b)
    a = np.asarray(value)
    if a.dtype.kind == 'b':
        a = a.astype(np.int8)
    if a.dtype.kind not in 'iuf':
        raise TypeError('no netCDF-4 attribute type for ' + repr(value))
    if a.size == 0:
        raise ValueError('an empty numeric attribute (%r) cannot be stored' % name)
    a = a.astype(a.dtype.newbyteorder('<'))
    # (netCDF attributes are one-dimensional; a single number is an array of one)
    return _attribute(name, datatype_message(a.dtype), (max(a.size, 1),) if a.ndim else (1,), np.atleast_1d(a).tobytes())


def check_attribute(value):
    """raises TypeError / ValueError for values that cannot be stored — an unsupported type, an empty array, 64 KiB or more —
    (what export.netcdf checks before it writes anything)"""
    _attr_message('x', value)


def _object_header(messages):
    body = b''.join(messages)
    return struct.pack('<BBHII4x', 1, 0, len(messages), 1, len(body)) + body


# ---- the pieces of the file ------------------------------------------------------------------------------------------

class Variable(object):
    def __init__(self, name, dtype, dims, fill_value=None, zlib=False, chunksizes=None, shape=None):
        self.name, self.dims = name, tuple(dims)
        self.shape, self.early = shape, None
        self.dtype = np.dtype(dtype).newbyteorder('<')
        datatype_message(self.dtype)
        self.attrs = OrderedDict()
        self.fill_value = None if fill_value is None else np.asarray(fill_value, dtype=self.dtype)
        self.zlib, self.chunksizes = bool(zlib), (None if chunksizes is None else tuple(int(c) for c in chunksizes))
        self.data = None

    def set(self, data):
        """``data`` must not change between this call and ``Writer.write`` (it is not copied)"""
        self.data = np.asarray(data)
        self.early = None
        if self.zlib and self.shape and self.data.nbytes >= (1 << 22):
            # a large compressed array starts to deflate NOW, on the pool's threads, while the caller prepares the next
            # variable (NaN fills, cell bounds: 0.6 s of NumPy per full frame, beside 0.8 s of deflate); write() picks the
            # dataset up and gives it its attributes
            self.early = _Dataset(self.name, self.dtype, self.shape, self.data, self.fill_value, True, self.chunksizes, None)


def _shuffle(raw, itemsize):
    if itemsize == 1:
        return raw
    a = np.frombuffer(raw, np.uint8)
    return a.reshape(-1, itemsize).T.tobytes()


class _Dataset(object):
    """One HDF5 dataset: a netCDF variable or the stand-in of a dimension without a variable."""

    def __init__(self, name, dtype, shape, data, fill_value, zlib, chunks, attrs):
        self.name, self.dtype, self.shape = name, np.dtype(dtype), tuple(shape)
        self.fill_value, self.attrs = fill_value, attrs
        self.address = 0
        self.extra = []                      # attribute messages that need addresses (made in the second pass)
        n = int(np.prod(shape)) if shape else 1
        if data is None:
            self.layout, self.raw, self._chunks = 'unallocated', b'', []
            self.nbytes = n * self.dtype.itemsize
            return
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(data), shape) if np.size(data) == 1 else
                                 np.asarray(data).reshape(shape), dtype=self.dtype)
        if chunks is None and zlib and shape:
            chunks = shape                       # (small fixed dimensions: the library's default is one chunk)
        if not chunks or not shape:
            self.layout, self.raw, self._chunks = 'contiguous', a.tobytes(), []
            self.nbytes = len(self.raw)
            return
        assert len(chunks) == len(shape) and all(c >= 1 for c in chunks)
        self.layout, self.chunk_shape, self.zlib = 'chunked', tuple(chunks), zlib
        grid = [-(-s // c) for s, c in zip(shape, chunks)]
        fill = default_fill(self.dtype) if fill_value is None else fill_value
        itemsize = self.dtype.itemsize

        def one(idx):
            off = tuple(i * c for i, c in zip(idx, chunks))
            block = a[tuple(slice(o, o + c) for o, c in zip(off, chunks))]
            if block.shape != tuple(chunks):         # edge chunk: padded with the fill value
                full = np.full(chunks, fill, dtype=self.dtype)
                full[tuple(slice(0, s) for s in block.shape)] = block
                block = full
            raw = np.ascontiguousarray(block).tobytes()
            if zlib:
                raw = _zlib.compress(_shuffle(raw, itemsize), _DEFLATE_LEVEL)
            return off, raw

        todo = list(np.ndindex(*grid))
        rows = tuple(chunks) == (1,) + tuple(shape[1:]) and len(shape) >= 2
        if zlib and rows and shape[0] > _ROWS_PER_TASK:
            # The reference's layout — one row per chunk (export/netcdf.py:128-326: chunksizes=(1, w)) — for a large array:
            # a task takes a block of rows, shuffles all of them with ONE NumPy copy (per-chunk Python work under the
            # interpreter lock was what bound the first version: 34 000 chunks per frame, 8 threads, 6.4 of 8.4 s) and
            # deflates them row by row, which releases the lock; the blocks run on a few threads.  (Small arrays too: the
            # resampled grid of a convert run has ~240 rows x 6 variables, and deflate's set-up per 2-KB chunk is 20 us.)
            zero = (0,) * (len(shape) - 1)

            def block(r0):
                blk = a[r0:r0 + _ROWS_PER_TASK]
                n = blk.shape[0]
                by = blk.view(np.uint8).reshape(n, -1, itemsize)
                sh = np.ascontiguousarray(by.transpose(0, 2, 1)).reshape(n, -1) if itemsize > 1 else by.reshape(n, -1)
                return [((r0 + i,) + zero, _zlib.compress(sh[i], _DEFLATE_LEVEL)) for i in range(n)]

            # (handed to the pool here, collected when the file is laid out: the variables of a file deflate side by side)
            pool = _pool()
            if _io.lib() is not None:
                # the helper library shuffles and deflates all rows of the variable outside the interpreter lock, on threads of
                # its own (export/csrc/amt_io.cpp: same zlib, same level, same bytes)
                def rows():
                    nt = max(1, min(_THREADS, a.nbytes >> 17))
                    return [((i,) + zero, raw) for i, raw in enumerate(_io.deflate_rows(a.reshape(shape[0], -1), _DEFLATE_LEVEL, True, nt))]
                self._chunks, self._blocks = None, [pool.submit(rows)]
            else:
                self._chunks, self._blocks = None, [pool.submit(block, r0) for r0 in range(0, shape[0], _ROWS_PER_TASK)]
        elif zlib and len(todo) >= 64 and a.nbytes >= (1 << 22):
            # (other chunk shapes) deflate releases the interpreter lock: the chunks are compressed by a few threads
            self._chunks = list(_pool().map(one, todo))
        else:
            self._chunks = [one(idx) for idx in todo]

    @property
    def chunks(self):
        if self._chunks is None:
            self._chunks = [c for part in self._blocks for c in part.result()]
            self._blocks = None
        return self._chunks

    # -- sizes (known before any address is) ----------------------------------------------------------------------------
    def _key_size(self):
        return 8 + 8 * (len(self.shape) + 1)

    def _node_size(self):
        return 24 + 2 * _CHUNK_K * 8 + (2 * _CHUNK_K + 1) * self._key_size()

    def _tree_levels(self):
        """number of nodes per level, leaves first"""
        counts, n = [], len(self.chunks)
        while True:
            n = -(-n // (2 * _CHUNK_K))
            counts.append(n)
            if n == 1:
                return counts

    def data_size(self):
        if self.layout == 'chunked':
            return sum(self._tree_levels()) * self._node_size() + sum(len(raw) + (-len(raw) % 8) for _, raw in self.chunks)
        return len(_pad8(self.raw))

    # -- serialisation -----------------------------------------------------------------------------------------------------
    def header(self, data_address):
        msgs = [_message(0x0001, _dataspace(self.shape)), _message(0x0003, datatype_message(self.dtype), flags=1)]
        fv = np.asarray(default_fill(self.dtype) if self.fill_value is None else self.fill_value, dtype=self.dtype).tobytes()
        msgs.append(_message(0x0005, struct.pack('<BBBBI', 2, 3 if self.layout == 'chunked' else 2, 0, 1, len(fv)) + fv))
        if self.layout == 'chunked':
            if self.zlib:
                pipeline = struct.pack('<BB6x', 1, 2)
                pipeline += struct.pack('<HHHH', 2, 0, 1, 1) + struct.pack('<I4x', self.dtype.itemsize)      # shuffle (optional)
                pipeline += struct.pack('<HHHH', 1, 0, 1, 1) + struct.pack('<I4x', _DEFLATE_LEVEL)           # deflate (optional)
                msgs.append(_message(0x000B, pipeline, flags=1))
            dims = b''.join(struct.pack('<I', c) for c in self.chunk_shape) + struct.pack('<I', self.dtype.itemsize)
            msgs.append(_message(0x0008, struct.pack('<BBBQ', 3, 2, len(self.shape) + 1, data_address) + dims))
        elif self.layout == 'contiguous':
            msgs.append(_message(0x0008, struct.pack('<BBQQ', 3, 1, data_address, len(self.raw))))
        else:
            msgs.append(_message(0x0008, struct.pack('<BBQQ', 3, 1, UNDEF, self.nbytes)))
        for k, v in self.attrs.items():
            msgs.append(_attr_message(k, v))
        return _object_header(msgs + self.extra)

    def data_blob(self, address):
        """the bytes behind the header at `address`: chunk B-tree nodes (root first) and chunks, or the contiguous array"""
        if self.layout != 'chunked':
            return _pad8(self.raw)
        rank, ks, ns = len(self.shape), self._key_size(), self._node_size()
        levels = self._tree_levels()
        # addresses: nodes level by level from the root down, then the chunks
        node_addr, a = [], address
        for count in reversed(levels):
            node_addr.append([a + i * ns for i in range(count)])
            a += count * ns
        node_addr.reverse()                     # node_addr[level][i], level 0 = leaves
        chunk_addr = []
        for _, raw in self.chunks:
            chunk_addr.append(a)
            a += len(raw) + (-len(raw) % 8)

        def key(size, offsets):
            return struct.pack('<II', size, 0) + b''.join(struct.pack('<Q', o) for o in offsets) + struct.pack('<Q', 0)

        end_off = list(self.chunks[-1][0])
        end_off[0] += self.chunk_shape[0]
        end_key = key(0, [end_off[0]] + [0] * (rank - 1))
        # children of the current level: (first key, address)
        children = [(key(len(raw), off), ca) for (off, raw), ca in zip(self.chunks, chunk_addr)]
        blobs = {}
        for level, count in enumerate(levels):
            parents = []
            for i in range(count):
                mine = children[i * 2 * _CHUNK_K:(i + 1) * 2 * _CHUNK_K]
                last = children[(i + 1) * 2 * _CHUNK_K][0] if (i + 1) * 2 * _CHUNK_K < len(children) else end_key
                left = node_addr[level][i - 1] if i > 0 else UNDEF
                right = node_addr[level][i + 1] if i + 1 < count else UNDEF
                body = b'TREE' + struct.pack('<BBHQQ', 1, level, len(mine), left, right)
                body += b''.join(k + struct.pack('<Q', c) for k, c in mine) + last
                blobs[node_addr[level][i]] = body + b'\0' * (ns - len(body))
                parents.append((mine[0][0], node_addr[level][i]))
            children = parents
        # (the chunks themselves are not joined: 700 MB of copies for a full frame — the file takes them one by one)
        parts = [b''.join(blobs[x] for x in sorted(blobs))]
        for _, raw in self.chunks:
            parts.append(raw)
            if len(raw) % 8:
                parts.append(_ZEROS[-len(raw) % 8])
        return _Parts(parts)


_ZEROS = [b'\0' * k for k in range(8)]


class _Parts(object):
    """bytes in pieces, for file.writelines"""

    def __init__(self, parts):
        self.parts, self.n = parts, sum(len(q) for q in parts)

    def __len__(self):
        return self.n


class _Sized(object):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


class Writer(object):
    """Collects dimensions, global attributes and variables, then writes the file in one go."""

    def __init__(self):
        self.dims = OrderedDict()
        self.attrs = OrderedDict()
        self.vars = OrderedDict()

    def create_dimension(self, name, size):
        if int(size) <= 0:
            raise ValueError('dimension %r of size %r: fixed dimensions need a positive size' % (name, size))
        self.dims[name] = int(size)

    def create_variable(self, name, dtype, dims=(), fill_value=None, zlib=False, chunksizes=None):
        if isinstance(dims, str):
            dims = (dims,)
        for d in dims:
            if d not in self.dims:
                raise KeyError('unknown dimension ' + d)
        if name in self.dims and tuple(dims) != (name,):
            raise ValueError('a variable named like a dimension must be its coordinate variable')
        v = self.vars[name] = Variable(name, dtype, dims, fill_value, zlib, chunksizes, tuple(self.dims[d] for d in dims))
        return v

    def write(self, path):
        dim_ids = {d: i for i, d in enumerate(self.dims)}
        datasets = OrderedDict()
        for d, n in self.dims.items():
            if d in self.vars:
                continue                          # the coordinate variable is the dimension scale
            datasets[d] = _Dataset(d, np.dtype('<f4'), (n,), None, None, False, None, OrderedDict())
        for v in self.vars.values():
            shape = tuple(self.dims[d] for d in v.dims)
            data = v.data
            if data is None:
                data = v.fill_value if v.fill_value is not None else default_fill(v.dtype)
            attrs = OrderedDict(v.attrs)
            if v.fill_value is not None:
                attrs = OrderedDict([('_FillValue', v.fill_value)] + list(attrs.items()))
            if v.early is not None:
                datasets[v.name] = v.early
                v.early.attrs = attrs
            else:
                datasets[v.name] = _Dataset(v.name, v.dtype, shape, data, v.fill_value, v.zlib, v.chunksizes, attrs)
        names = sorted(datasets, key=lambda s: s.encode('utf-8'))
        if len(names) > 2 * 32767:
            raise ValueError('too many objects for one symbol-table node')
        leaf_k = max(_GROUP_K, -(-len(names) // 2))

        # which variables use which dimension: (variable, index of the dimension in the variable)
        users = {d: [] for d in self.dims}
        for v in self.vars.values():
            for i, d in enumerate(v.dims):
                if d != v.name or len(v.dims) != 1:
                    users[d].append((v.name, i))
                elif d == v.name:
                    pass                              # a coordinate variable does not list itself
        # global heap: one object (an object reference) per entry of every DIMENSION_LIST
        dimlist = [(v.name, i, d) for v in self.vars.values() if not (v.name in self.dims) for i, d in enumerate(v.dims)]
        heap_objects = len(dimlist)
        gcol_size = max(4096, 16 + 24 * heap_objects + 16)
        gcol_size += -gcol_size % 8

        def build(addr, sizes_only=False):
            """every piece with the addresses of `addr` (dataset name -> header address, plus 'gcol') -> ordered list of
            (name, bytes)"""
            for name, ds in datasets.items():
                ds.extra = []
                if name in self.dims:
                    label = name if name in self.vars else NOT_A_VARIABLE % self.dims[name]
                    ds.extra.append(_attr_message('CLASS', 'DIMENSION_SCALE'))
                    ds.extra.append(_attr_message('NAME', label))
                    ds.extra.append(_attribute('_Netcdf4Dimid', _dt_fixed(4, True), (), struct.pack('<i', dim_ids[name])))
                    refs = users[name]
                    if refs:
                        data = b''.join(struct.pack('<QI4x', addr[vn], i) for vn, i in refs)
                        ds.extra.append(_attribute('REFERENCE_LIST', _dt_reference_list(), (len(refs),), data))
                if name in self.vars and name not in self.dims and self.vars[name].dims:
                    v = self.vars[name]
                    data = b''
                    for i, d in enumerate(v.dims):
                        obj = dimlist.index((name, i, d)) + 1
                        data += struct.pack('<IQI', 1, addr['gcol'], obj)
                    ds.extra.append(_attribute('DIMENSION_LIST', _DT_VLEN_REF, (len(v.dims),), data))
            pieces = []
            # root group: object header (symbol table message + global attributes), local heap, B-tree node, symbol table node
            heap_data = b'\0' * 8
            name_off = {}
            for nm in names:
                name_off[nm] = len(heap_data)
                heap_data += _pad8(nm.encode('utf-8') + b'\0')
            free_off = len(heap_data)
            heap_data += struct.pack('<QQ', 1, 32) + b'\0' * 16          # one free block of 32 bytes ends the segment
            root_msgs = [_message(0x0011, struct.pack('<QQ', addr['btree'], addr['heap']))]
            root_msgs += [_attr_message(k, v) for k, v in self.attrs.items()]
            pieces.append(('root', _object_header(root_msgs)))
            pieces.append(('heap', b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap_data), free_off, addr['heap'] + 32) + heap_data))
            node = b'TREE' + struct.pack('<BBHQQ', 0, 0, 1, UNDEF, UNDEF) + struct.pack('<QQQ', 0, addr['snod'], name_off[names[-1]])
            pieces.append(('btree', node + b'\0' * (24 + (2 * _GROUP_K + 1) * 8 + 2 * _GROUP_K * 8 - len(node))))
            snod = b'SNOD' + struct.pack('<BBH', 1, 0, len(names))
            for nm in names:
                snod += struct.pack('<QQII16x', name_off[nm], addr[nm], 0, 0)
            pieces.append(('snod', snod + b'\0' * (8 + 2 * leaf_k * 40 - len(snod))))
            # global heap collection
            g = b'GCOL' + struct.pack('<B3xQ', 1, gcol_size)
            for k, (vn, i, d) in enumerate(dimlist):
                g += struct.pack('<HH4xQ', k + 1, 1, 8) + struct.pack('<Q', addr[d])
            free = gcol_size - len(g)
            if free >= 16:
                g += struct.pack('<HH4xQ', 0, 0, free)
            pieces.append(('gcol', g + b'\0' * (gcol_size - len(g))))
            for name, ds in datasets.items():
                pieces.append((name, ds.header(addr.get(name + ':data', 0))))
                if ds.layout != 'unallocated':
                    # (the sizing pass only asks for the length: the blob of a large array is hundreds of MB of joins)
                    pieces.append((name + ':data', _Sized(ds.data_size()) if sizes_only else
                                   ds.data_blob(addr.get(name + ':data', 0))))
            return pieces

        # first pass: sizes (they do not depend on the addresses), then the addresses, then the real thing
        addr = {k: 0 for k in list(datasets) + ['gcol', 'btree', 'heap', 'snod']}
        sizes = [(name, len(b)) for name, b in build(addr, sizes_only=True)]
        pos = 96                                   # behind the superblock
        for name, n in sizes:
            addr[name] = pos
            pos += n + (-n % 8)
        eof = pos
        pieces = build(addr)
        assert [(name, len(b)) for name, b in pieces] == sizes
        sb = b'\x89HDF\r\n\x1a\n' + struct.pack('<BBBBBBBB', 0, 0, 0, 0, 0, 8, 8, 0)
        sb += struct.pack('<HHI', leaf_k, _GROUP_K, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
        sb += struct.pack('<QQII', 0, addr['root'], 1, 0) + struct.pack('<QQ', addr['btree'], addr['heap'])
        assert len(sb) == 96
        with open(path, 'wb') as fp:
            fp.write(sb)
            for name, b in pieces:
                assert fp.tell() == addr[name], name
                if isinstance(b, _Parts):
                    fp.writelines(b.parts)
                else:
                    fp.write(b)
                fp.write(b'\0' * (-len(b) % 8))
            assert fp.tell() == eof


# ---- reader -------------------------------------------------------------------------------------------------------------------
# Reads what the writer above writes, and any HDF5 file made of the same (version 1.1) structures — superblock version 0 / 1, symbol
# table groups, version-1 object headers with continuation blocks, contiguous and chunked layouts with the shuffle and deflate
# filters, fixed-point / floating-point / fixed-length string types —, e.g. a file of h5py with its default settings.  Files of
# the netCDF library itself use the newer structures (version-2 object headers, link messages, fractal heaps) and are refused
# with a message that says so.

class ReadVariable(object):
    def __init__(self, name, dims, attrs, data):
        self.name, self.dims, self.attrs, self.data = name, dims, attrs, data

    def __getattr__(self, k):
        try:
            return self.attrs[k]
        except KeyError:
            raise AttributeError(k)


def _read_datatype(buf, pos):
    """-> (numpy dtype | ('S', size) | None for types the reader does not decode, size of the element)"""
    cv, b0, b1, b2, size = struct.unpack_from('<BBBBI', buf, pos)
    cls = cv & 0x0F
    order = '>' if b0 & 1 else '<'
    if cls == 0:
        return np.dtype(order + ('i' if b0 & 0x08 else 'u') + str(size)), size
    if cls == 1:
        return np.dtype(order + 'f' + str(size)), size
    if cls == 3:
        return ('S', size), size
    return None, size


class File(object):
    """A netCDF-4 file of this writer's kind read into memory: ``dims`` (name -> size), ``attrs``, ``vars``."""

    def __init__(self, path):
        with open(path, 'rb') as fp:
            self.buf = buf = memoryview(fp.read())
        assert bytes(buf[:8]) == b'\x89HDF\r\n\x1a\n', 'not an HDF5 file'
        version = buf[8]
        if version > 1 or buf[13] != 8 or buf[14] != 8:
            raise NotImplementedError('HDF5 superblock version %d (the netCDF library writes version 2 with the newer object '
                                      'headers): convert with `nccopy -k nc6`, or read it with the netCDF4 library' % version)
        entry = 24 + (4 if version == 1 else 0) + 32
        root_header = struct.unpack_from('<Q', buf, entry + 8)[0]
        objects = OrderedDict()
        root_msgs = self._messages(root_header)
        self.attrs = OrderedDict()
        for t, pos, size in root_msgs:
            if t == 0x0011:
                btree, heap = struct.unpack_from('<QQ', buf, pos)
                heap_data = struct.unpack_from('<Q', buf, heap + 24)[0]
                self._walk_group(btree, heap_data, objects)
            elif t == 0x000C:
                k, v = self._attribute(pos)
                self.attrs[k] = v
        raw = OrderedDict()
        for name, address in objects.items():
            raw[name] = self._dataset(name, address)
        # dimensions: the datasets marked as dimension scales, in the order of their netCDF ids
        scales = [(d['attrs'].get('_Netcdf4Dimid', 1 << 30), name) for name, d in raw.items()
                  if d['attrs'].get('CLASS') == 'DIMENSION_SCALE']
        self.dims = OrderedDict((name, raw[name]['shape'][0]) for _, name in sorted(scales))
        by_address = {objects[name]: name for _, name in scales}
        self.vars = OrderedDict()
        for name, d in raw.items():
            label = d['attrs'].get('NAME', '')
            if name in self.dims and isinstance(label, str) and label.startswith(NOT_A_VARIABLE[:40]):
                continue                              # a dimension without a variable
            dims = tuple(by_address.get(a, '?') for a in d['dimension_list']) if d['dimension_list'] else \
                ((name,) if name in self.dims else ())
            attrs = OrderedDict((k, v) for k, v in d['attrs'].items()
                                if k not in ('DIMENSION_LIST', 'REFERENCE_LIST', 'CLASS', 'NAME', '_Netcdf4Dimid', '_Netcdf4Coordinates'))
            self.vars[name] = ReadVariable(name, dims, attrs, d['data'])

    # -- structures --------------------------------------------------------------------------------------------------------------
    def _messages(self, address):
        """[(type, position of the data, size)] of a version-1 object header, continuation blocks included"""
        buf = self.buf
        if bytes(buf[address:address + 4]) == b'OHDR':
            raise NotImplementedError('version-2 object header (a file of the netCDF / a recent HDF5 library): not read here')
        version, _, nmsgs, _, size = struct.unpack_from('<BBHII', buf, address)
        assert version == 1, 'object header version %d' % version
        blocks, out = [(address + 16, size)], []
        while blocks and len(out) < nmsgs:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsgs:
                t, n, flags = struct.unpack_from('<HHB', buf, pos)
                if t == 0x0010:
                    blocks.append(struct.unpack_from('<QQ', buf, pos + 8))
                out.append((t, pos + 8, n))
                pos += 8 + n
        return out

    def _walk_group(self, node, heap_data, objects):
        buf = self.buf
        sig = bytes(buf[node:node + 4])
        if sig == b'TREE':
            _, level, used = struct.unpack_from('<BBH', buf, node + 4)
            for i in range(used):
                child = struct.unpack_from('<Q', buf, node + 24 + 8 + 16 * i)[0]
                self._walk_group(child, heap_data, objects)
        elif sig == b'SNOD':
            n = struct.unpack_from('<H', buf, node + 6)[0]
            for i in range(n):
                off, address = struct.unpack_from('<QQ', buf, node + 8 + 40 * i)
                end = heap_data + off
                while buf[end] != 0:
                    end += 1
                objects[bytes(buf[heap_data + off:end]).decode('utf-8')] = address
        else:
            raise ValueError('unexpected structure %r in a group' % sig)

    def _attribute(self, pos, dimension_list=None, name_only=False):
        buf = self.buf
        version, _, nsize, dtsize, dssize = struct.unpack_from('<BBHHH', buf, pos)
        assert version == 1, 'attribute message version %d' % version
        p = pos + 8
        name = bytes(buf[p:p + nsize]).split(b'\0')[0].decode('utf-8')
        p += nsize + (-nsize % 8)
        dt, esize = _read_datatype(buf, p)
        p += dtsize + (-dtsize % 8)
        rank = buf[p + 1]
        shape = struct.unpack_from('<%dQ' % rank, buf, p + 8) if rank else ()
        p += dssize + (-dssize % 8)
        n = int(np.prod(shape)) if shape else 1
        if name == 'DIMENSION_LIST':
            # variable-length sequences of one object reference each: (length, global heap collection, object index)
            refs = []
            for i in range(n):
                _, col, idx = struct.unpack_from('<IQI', buf, p + 16 * i)
                refs.append(self._heap_object(col, idx))
            return name, refs
        if dt is None:
            return name, None
        if isinstance(dt, tuple):
            return name, bytes(buf[p:p + esize]).split(b'\0')[0].decode('utf-8')
        a = np.frombuffer(buf, dt, n, p).astype(dt.newbyteorder('='))
        return name, (a[0] if n == 1 else a)

    def _heap_object(self, collection, index):
        buf = self.buf
        assert bytes(buf[collection:collection + 4]) == b'GCOL'
        size = struct.unpack_from('<Q', buf, collection + 8)[0]
        p = collection + 16
        while p + 16 <= collection + size:
            idx, _, n = struct.unpack_from('<HH4xQ', buf, p)
            if idx == index:
                return struct.unpack_from('<Q', buf, p + 16)[0]
            if idx == 0:
                break
            p += 16 + n + (-n % 8)
        raise KeyError('global heap object %d' % index)

    def _dataset(self, name, address):
        buf = self.buf
        shape, dt, layout, filters, attrs, dimlist, fill = (), None, None, [], OrderedDict(), None, None
        for t, pos, size in self._messages(address):
            if t == 0x0005 and buf[pos] in (1, 2) and (buf[pos] == 1 or buf[pos + 3]):
                fill = (pos + 8, struct.unpack_from('<I', buf, pos + 4)[0])          # (position, size) of the fill value
            if t == 0x0001:
                rank = buf[pos + 1]
                shape = tuple(struct.unpack_from('<%dQ' % rank, buf, pos + 8)) if rank else ()
            elif t == 0x0003:
                dt, _ = _read_datatype(buf, pos)
            elif t == 0x0008:
                version, cls = buf[pos], buf[pos + 1]
                assert version == 3, 'data layout version %d' % version
                if cls == 1:
                    layout = ('contiguous',) + struct.unpack_from('<QQ', buf, pos + 2)
                elif cls == 2:
                    nd = buf[pos + 2]
                    tree = struct.unpack_from('<Q', buf, pos + 3)[0]
                    layout = ('chunked', tree, struct.unpack_from('<%dI' % nd, buf, pos + 11))
                else:
                    layout = ('compact', pos + 4, struct.unpack_from('<H', buf, pos + 2)[0])
            elif t == 0x000B:
                assert buf[pos] == 1, 'filter pipeline version %d' % buf[pos]
                p = pos + 8
                for _ in range(buf[pos + 1]):
                    fid, nlen, _, ncd = struct.unpack_from('<HHHH', buf, p)
                    p += 8 + nlen + (-nlen % 8)
                    filters.append((fid, struct.unpack_from('<%dI' % ncd, buf, p)))
                    p += 4 * (ncd + (ncd & 1))
            elif t == 0x000C:
                k, v = self._attribute(pos)
                if k == 'DIMENSION_LIST':
                    dimlist = v
                attrs[k] = v
        assert dt is not None and not isinstance(dt, tuple), 'dataset %s: type not read' % name
        n = int(np.prod(shape)) if shape else 1
        if layout[0] == 'contiguous':
            data = (np.frombuffer(buf, dt, n, layout[1]) if layout[1] != UNDEF else np.zeros(0, dt))
            data = data.reshape(shape) if data.size == n else None
        elif layout[0] == 'compact':
            data = np.frombuffer(buf, dt, n, layout[1]).reshape(shape)
        else:
            chunk = tuple(layout[2][:-1])
            # (chunks that were never written read as the fill value)
            data = np.zeros(shape, dt)
            if fill is not None and fill[1] == dt.itemsize:
                data[...] = np.frombuffer(buf, dt, 1, fill[0])[0]
            self._read_chunks(layout[1], len(shape), chunk, dt, filters, data)
        if data is not None:
            data = data.astype(dt.newbyteorder('='))
        return dict(shape=shape, attrs=attrs, data=data, dimension_list=dimlist)

    def _read_chunks(self, node, rank, chunk, dt, filters, out):
        buf = self.buf
        if node == UNDEF:
            return
        assert bytes(buf[node:node + 4]) == b'TREE' and buf[node + 4] == 1
        level, used = struct.unpack_from('<BH', buf, node + 5)
        ks = 8 + 8 * (rank + 1)
        for i in range(used):
            kp = node + 24 + i * (ks + 8)
            nbytes, mask = struct.unpack_from('<II', buf, kp)
            offs = struct.unpack_from('<%dQ' % rank, buf, kp + 8)
            child = struct.unpack_from('<Q', buf, kp + ks)[0]
            if level > 0:
                self._read_chunks(child, rank, chunk, dt, filters, out)
                continue
            raw = bytes(buf[child:child + nbytes])
            for k, (fid, cd) in reversed(list(enumerate(filters))):
                if mask & (1 << k):
                    continue
                if fid == 1:
                    raw = _zlib.decompress(raw)
                elif fid == 2:
                    s = dt.itemsize
                    raw = np.frombuffer(raw, np.uint8).reshape(s, -1).T.tobytes() if s > 1 else raw
                else:
                    raise NotImplementedError('HDF5 filter %d' % fid)
            block = np.frombuffer(raw, dt).reshape(chunk)
            sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, out.shape))
            out[sel] = block[tuple(slice(0, x.stop - x.start) for x in sel)]


def open_file(path):
    """The file as :class:`File` here or as :class:`auromat_amd.export._nc3.File`, whichever format it is in."""
    from . import _nc3
    with open(path, 'rb') as fp:
        magic = fp.read(8)
    if magic[:3] == b'CDF':
        return _nc3.File(path)
    if magic == b'\x89HDF\r\n\x1a\n':
        return File(path)
    raise ValueError('%s is neither a netCDF classic nor a netCDF-4 (HDF5) file' % path)
