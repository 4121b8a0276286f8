"""
NASA's Common Data Format, version 3, single-file — the container behind the reference's CDF exporter
(auromat/export/cdf.py:25-285 hands its arrays to ``spacepy.pycdf``, which drives NASA's CDF library).  Neither exists in
this image, so this module lays the file out itself, record by record, after the published *CDF Internal Format
Description* (version 3.4 and later: 8-byte offsets), and reads such files back.

UNPINNED: no CDF library, no ``cdflib`` and no reader other than the one below has ever opened a file written here; what
the tests hold is (i) the record structure against the format description (sizes, types, chains, every byte of the file
owned by exactly one record), (ii) a round trip through :class:`Reader`, which was written from the same description but
shares no code with :class:`Writer` beyond the type table and the leap-second table, and (iii) the two time encodings
against their published epoch constants.

What a file consists of (all record headers big-endian, whatever the data encoding)::

    magic 0xCDF30001 0x0000FFFF
    CDR   descriptor: version, encoding of the values, majority, offset of the GDR
    GDR   heads of the three chains (zVariables, attributes), end of file
    ADR   one per attribute (global or variable scope) -> chain of AgrEDR / AzEDR entries (type, count, value)
    zVDR  one per variable: type, dimensions, record variance, compression -> CPR (gzip level), -> VXR index records
    VXR   index: (first record, last record, offset) -> VVR (records as they are) or CVVR (gzip-compressed block)

Only what the exporter needs is written: zVariables (the only kind spacepy creates), row majority, little-endian values
(IBMPC encoding: what the library picks on the hosts the reference runs on), no sparse records, no checksum, GZIP
compression per variable.  The reader follows the same subset plus big-endian encodings, rVariables-free files, nested
VXRs and uncompressed / compressed blocks.
"""
import os
import struct
import zlib as _zlib
from collections import OrderedDict
from datetime import datetime, timedelta

import numpy as np

MAGIC = b'\xCD\xF3\x00\x01\x00\x00\xFF\xFF'

# data types (CDF Internal Format Description, "Data types")
CDF_INT1, CDF_INT2, CDF_INT4, CDF_INT8 = 1, 2, 4, 8
CDF_UINT1, CDF_UINT2, CDF_UINT4 = 11, 12, 14
CDF_REAL4, CDF_REAL8 = 21, 22
CDF_EPOCH, CDF_EPOCH16, CDF_TIME_TT2000 = 31, 32, 33
CDF_BYTE, CDF_FLOAT, CDF_DOUBLE = 41, 44, 45
CDF_CHAR, CDF_UCHAR = 51, 52

_KIND = {CDF_INT1: 'i1', CDF_INT2: 'i2', CDF_INT4: 'i4', CDF_INT8: 'i8', CDF_UINT1: 'u1', CDF_UINT2: 'u2', CDF_UINT4: 'u4',
         CDF_REAL4: 'f4', CDF_REAL8: 'f8', CDF_EPOCH: 'f8', CDF_TIME_TT2000: 'i8', CDF_BYTE: 'i1', CDF_FLOAT: 'f4',
         CDF_DOUBLE: 'f8'}
_OF_DTYPE = {'i1': CDF_INT1, 'i2': CDF_INT2, 'i4': CDF_INT4, 'i8': CDF_INT8, 'u1': CDF_UINT1, 'u2': CDF_UINT2, 'u4': CDF_UINT4,
             'f4': CDF_FLOAT, 'f8': CDF_DOUBLE}

# record types
_CDR, _GDR, _RVDR, _ADR, _AGREDR, _VXR, _VVR, _ZVDR, _AZEDR, _CCR, _CPR, _SPR, _CVVR = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13

IBMPC_ENCODING, NETWORK_ENCODING = 6, 1
_LITTLE = (4, 6, 13, 16, 17, 19)                      # DECSTATION, IBMPC, ALPHAOSF1, ALPHAVMSi, ARM_LITTLE, IA64VMSi (IEEE; the
                                                      # VAX encodings 3, 14, 15, 20, 21 have D / G floats and are refused)
_BIG = (1, 2, 5, 7, 9, 11, 12, 18)                    # NETWORK, SUN, SGi, IBMRS, PPC, HP, NeXT, ARM_BIG
GZIP_COMPRESSION = 5
_VXR_ENTRIES = 7
# native threads that deflate the blocks of one large record (AMT_IO_THREADS; AMT_NC4_THREADS, the netCDF-4 writer's switch, is
# honoured too: both writers share the helper and its thread budget)
def _io_threads():
    from .._native import host_threads
    return int(os.environ.get('AMT_IO_THREADS', os.environ.get('AMT_NC4_THREADS', '0'))) or host_threads(16)


_GZIP_THREADS = max(1, min(16, _io_threads()))          # (this rank's share of the host's cores: _native.host_threads)
_LEAP_TABLE_DATE = 20170101

# TAI - UTC in whole seconds from the given day on (IERS Bulletin C; the table the CDF library carries for TT2000)
LEAP_SECONDS = [((1972, 1, 1), 10), ((1972, 7, 1), 11), ((1973, 1, 1), 12), ((1974, 1, 1), 13), ((1975, 1, 1), 14),
                ((1976, 1, 1), 15), ((1977, 1, 1), 16), ((1978, 1, 1), 17), ((1979, 1, 1), 18), ((1980, 1, 1), 19),
                ((1981, 7, 1), 20), ((1982, 7, 1), 21), ((1983, 7, 1), 22), ((1985, 7, 1), 23), ((1988, 1, 1), 24),
                ((1990, 1, 1), 25), ((1991, 1, 1), 26), ((1992, 7, 1), 27), ((1993, 7, 1), 28), ((1994, 7, 1), 29),
                ((1996, 1, 1), 30), ((1997, 7, 1), 31), ((1999, 1, 1), 32), ((2006, 1, 1), 33), ((2009, 1, 1), 34),
                ((2012, 7, 1), 35), ((2015, 7, 1), 36), ((2017, 1, 1), 37)]
_J2000_NOON = datetime(2000, 1, 1, 12)
_TT_MINUS_TAI_NS = 32184000000
_EPOCH_2000_MS = 63113904000000.0           # CDF_EPOCH (milliseconds since 0000-01-01T00:00) of 2000-01-01T00:00:00


def _leap(dt):
    if dt < datetime(1972, 1, 1):
        raise ValueError('CDF_TIME_TT2000 before 1972 (drifting UTC) is not supported')
    n = 10
    for day, v in LEAP_SECONDS:
        if dt >= datetime(*day):
            n = v
    return n


def datetime_to_tt2000(dt):
    """UTC ``datetime`` -> nanoseconds of Terrestrial Time since J2000 (2000-01-01T12:00:00 TT), the CDF_TIME_TT2000 value:
    TT = UTC + (TAI - UTC) + 32.184 s.  (A ``datetime`` cannot name a leap second itself; neither can the reference's.)"""
    d = dt - _J2000_NOON
    return (d.days * 86400 + d.seconds) * 10**9 + d.microseconds * 1000 + _leap(dt) * 10**9 + _TT_MINUS_TAI_NS


def tt2000_to_datetime(ns):
    ns = int(ns)
    # UTC = TT - 32.184 s - (TAI - UTC); the leap count depends on the UTC date: settle it by trying the table from the end
    for day, v in reversed([((1972, 1, 1), 10)] + LEAP_SECONDS):
        utc_ns = ns - _TT_MINUS_TAI_NS - v * 10**9
        dt = _J2000_NOON + timedelta(microseconds=utc_ns // 1000)
        if dt >= datetime(*day):
            return dt
    raise ValueError('CDF_TIME_TT2000 before 1972')


def datetime_to_epoch(dt):
    """CDF_EPOCH: milliseconds since 0000-01-01T00:00:00 as a double."""
    d = dt - datetime(2000, 1, 1)
    return _EPOCH_2000_MS + (d.days * 86400 + d.seconds) * 1000.0 + d.microseconds / 1000.0


def epoch_to_datetime(ms):
    return datetime(2000, 1, 1) + timedelta(milliseconds=float(ms) - _EPOCH_2000_MS)


def _int_type(lo, hi):
    """the smallest integer type that holds [lo, hi] (spacepy's rule for values without a NumPy type of their own)"""
    for t, (a, b) in ((CDF_BYTE, (-128, 127)), (CDF_UINT1, (0, 255)), (CDF_INT2, (-32768, 32767)), (CDF_UINT2, (0, 65535)),
                      (CDF_INT4, (-2**31, 2**31 - 1)), (CDF_UINT4, (0, 2**32 - 1)), (CDF_INT8, (-2**63, 2**63 - 1))):
        if a <= lo and hi <= b:
            return t
    raise ValueError('integer out of the range of CDF_INT8')


def infer(value, tt2000=True):
    """a Python / NumPy value -> (CDF type, number of elements, bytes in little-endian order).  NumPy arrays and scalars keep
    their own type (float64 -> CDF_DOUBLE: nothing is narrowed), plain Python numbers take the smallest integer type /
    CDF_DOUBLE, ``datetime`` CDF_TIME_TT2000 (or CDF_EPOCH), text CDF_CHAR (UTF-8; an empty text is stored as one blank,
    a CDF entry cannot be empty)."""
    if isinstance(value, bytes):
        value = value.decode('utf-8')
    if isinstance(value, str):
        raw = value.encode('utf-8') or b' '
        return CDF_CHAR, len(raw), raw
    if isinstance(value, datetime):
        value = [value]
    if isinstance(value, (list, tuple)) and value and all(isinstance(v, datetime) for v in value):
        if tt2000:
            return CDF_TIME_TT2000, len(value), np.array([datetime_to_tt2000(v) for v in value], '<i8').tobytes()
        return CDF_EPOCH, len(value), np.array([datetime_to_epoch(v) for v in value], '<f8').tobytes()
    if isinstance(value, (bool, np.bool_)):
        value = int(value)
    if isinstance(value, int) or (isinstance(value, (list, tuple)) and value and all(type(v) is int for v in value)):
        vals = [value] if isinstance(value, int) else list(value)
        t = _int_type(min(vals), max(vals))
        return t, len(vals), np.array(vals, '<' + _KIND[t]).tobytes()
    a = np.atleast_1d(np.asarray(value))
    if a.size == 0:
        raise ValueError('a CDF entry cannot be empty (%r)' % (value,))
    if a.dtype.kind == 'b':
        a = a.astype('u1')
    if a.dtype.kind in 'SU' or a.dtype.kind == 'O':
        raise TypeError('cannot store %r in one CDF entry' % (value,))
    key = a.dtype.kind + str(a.dtype.itemsize)
    if key not in _OF_DTYPE:
        raise TypeError('no CDF type for NumPy dtype %s' % a.dtype)
    return _OF_DTYPE[key], a.size, np.ascontiguousarray(a, dtype='<' + key).tobytes()


def _entries_of(value, tt2000):
    """one attribute value -> its entries [(type, n, bytes)] (a list of texts: one entry per text, as pycdf)"""
    if isinstance(value, (list, tuple)) and value and all(isinstance(s, (str, bytes)) for s in value):
        return [infer(s, tt2000) for s in value]
    return [infer(value, tt2000)]


class _Attrs(OrderedDict):
    """Attributes of the file (scope 1) or of one variable (scope 2).  A value is checked when it is assigned — not in
    ``write``, after every variable has been compressed —: its name's length, that the name is not in use in the other scope
    (a CDF attribute is either global or of variable scope), that the value has a CDF type and is not empty, and, for a
    variable, that it is ONE entry (a variable has one entry per attribute; a list of texts would need several)."""

    def __init__(self, writer, scope):
        OrderedDict.__init__(self)
        self._writer, self._scope = writer, scope

    def __setitem__(self, key, value):
        w = self._writer
        if len(key.encode('utf-8')) > 255:
            raise ValueError('attribute names are at most 255 bytes long')
        if self._scope == 1:
            clash = any(key in v.attrs for v in w.vars.values())
        else:
            clash = key in w.attrs
        if clash:
            raise ValueError('an attribute name is either global or of variable scope: %s' % key)
        entries = _entries_of(value, w.tt2000)
        if self._scope == 2 and len(entries) != 1:
            raise ValueError('a variable has ONE entry per attribute: %s = %r' % (key, value))
        OrderedDict.__setitem__(self, key, value)

    def __reduce__(self):                                   # (OrderedDict's would call __init__ without arguments)
        return (OrderedDict, (list(self.items()),))


class Var(object):
    def __init__(self, name, cdf_type, n_elems, dims, rec_vary, records, compress, writer=None):
        self.name, self.type, self.n_elems, self.dims = name, cdf_type, n_elems, tuple(int(d) for d in dims)
        self.rec_vary, self.records, self.compress = rec_vary, records, compress
        self.attrs = _Attrs(writer, 2) if writer is not None else OrderedDict()


class Writer(object):
    """Collects global attributes and zVariables, then lays the file out in one go (``write``).  The call surface is the part
    of ``spacepy.pycdf.CDF`` the reference's exporter uses: ``attrs[...] = value``, ``new(name, data, type=..., recVary=...,
    compress=...)`` and ``var.attrs[...] = value``."""

    def __init__(self, tt2000=True, pool=None):
        """``pool``: a ``concurrent.futures`` executor; a large compressed variable then starts to compress when it is created
        (``new``), beside whatever the caller does to prepare the next one, and ``write`` collects the streams."""
        self.vars = OrderedDict()
        self.tt2000 = tt2000
        self.pool = pool
        self.attrs = _Attrs(self, 1)

    def __getitem__(self, name):
        return self.vars[name]

    def new(self, name, data, type=None, recVary=True, compress=None, compress_param=5):
        """``data``: with ``recVary`` the first axis counts the records (as in pycdf), without it the whole array is the one
        record.  ``compress``: None or GZIP_COMPRESSION with level ``compress_param`` (pycdf's default: 5).  An array that
        already has the variable's type and layout is NOT copied: it must not change until ``write`` has returned."""
        if name in self.vars:
            raise KeyError('variable %s exists' % name)
        if len(name.encode('utf-8')) > 255:
            raise ValueError('variable names are at most 255 bytes long')
        seq = isinstance(data, (list, tuple)) and data and all(isinstance(v, datetime) for v in data)
        if isinstance(data, datetime) or seq:
            vals = [data] if isinstance(data, datetime) else list(data)
            if type is None:
                type = CDF_TIME_TT2000 if self.tt2000 else CDF_EPOCH
            if type == CDF_TIME_TT2000:
                a = np.array([datetime_to_tt2000(v) for v in vals], '<i8')
            elif type == CDF_EPOCH:
                a = np.array([datetime_to_epoch(v) for v in vals], '<f8')
            else:
                raise ValueError('a datetime needs a time type')
            if isinstance(data, datetime):
                a = a.reshape(())
        else:
            if isinstance(data, (bool, int)) and type is None:
                type = _int_type(int(data), int(data))
            a = np.asarray(data)
            if a.dtype.kind == 'b':
                a = a.astype('u1')
            if type is None:
                key = a.dtype.kind + str(a.dtype.itemsize)
                if key not in _OF_DTYPE:
                    raise TypeError('no CDF type for NumPy dtype %s' % a.dtype)
                type = _OF_DTYPE[key]
            a = np.asarray(a, dtype='<' + _KIND[type], order='C')          # (ascontiguousarray would make a scalar 1-D)
        if recVary:
            if a.ndim == 0:
                a = a.reshape(1)
            dims, records = a.shape[1:], a
        else:
            dims, records = a.shape, a.reshape((1,) + a.shape)
        if records.shape[0] < 1:
            raise ValueError('a variable without records is not written')
        if compress not in (None, 0, GZIP_COMPRESSION):
            raise ValueError('only GZIP compression is written')
        v = self.vars[name] = Var(name, type, 1, dims, bool(recVary), records,
                                  int(compress_param) if compress == GZIP_COMPRESSION else None, writer=self)
        v.blocks = None
        if self.pool is not None and v.compress is not None and records.nbytes >= (1 << 22):
            v.blocks = self.pool.submit(_blocks_of, v)           # (`data` must not change until write())
        return v

    # -- layout -------------------------------------------------------------------------------------------------------------
    def write(self, path, pool=None):
        """``pool``: a ``concurrent.futures`` executor for the gzip of the compressed variables (each record block is ONE gzip
        stream — the format leaves nothing to split —, so the variables of a file are what can run side by side)."""
        names = list(self.vars)
        # attributes: global ones in the order given, then the variables' in order of first use
        attr_names = [(k, 1) for k in self.attrs]
        seen = set()
        for v in self.vars.values():
            for k in v.attrs:
                if k not in seen:
                    seen.add(k)
                    attr_names.append((k, 2))
        # (names, scopes and values were checked when they were assigned: _Attrs)

        def entries_of(value):
            return _entries_of(value, self.tt2000)

        # variable data first (sizes of the compressed blocks are needed for the addresses)
        pool = pool or self.pool
        todo = [v for v in self.vars.values() if getattr(v, 'blocks', None) is None]
        fresh = dict(zip([v.name for v in todo], pool.map(_blocks_of, todo) if pool is not None else map(_blocks_of, todo)))
        blocks = [fresh[v.name] if v.name in fresh else v.blocks.result() for v in self.vars.values()]

        # pass 1: the size of every record, in file order -> addresses
        pieces = []                                               # (key, size)
        pieces.append(('cdr', 312))
        pieces.append(('gdr', 84))
        attr_entries = []
        for num, (k, scope) in enumerate(attr_names):
            pieces.append(('adr', num, 324))
            if scope == 1:
                ent = [(i, e) for i, e in enumerate(entries_of(self.attrs[k]))]
            else:
                ent = [(vi, e) for vi, nm in enumerate(names) if k in self.vars[nm].attrs
                       for e in entries_of(self.vars[nm].attrs[k])]
            attr_entries.append(ent)
            for j, (_, (t, n, raw)) in enumerate(ent):
                pieces.append(('aedr', num, j, 56 + len(raw)))
        for vi, nm in enumerate(names):
            v = self.vars[nm]
            pieces.append(('vdr', vi, 344 + 8 * len(v.dims)))
            if v.compress is not None:
                pieces.append(('cpr', vi, 28))
            nvxr = -(-len(blocks[vi]) // _VXR_ENTRIES)
            for x in range(nvxr):
                pieces.append(('vxr', vi, x, 28 + 16 * _VXR_ENTRIES))
            for bi, (_, _, kind, raw) in enumerate(blocks[vi]):
                pieces.append(('blk', vi, bi, (12 if kind == _VVR else 24) + len(raw)))
        addr, at = {}, 8
        for p in pieces:
            addr[p[:-1]] = at
            at += p[-1]
        eof = at

        def name256(s):
            b = s.encode('utf-8')
            return b + b'\0' * (256 - len(b))

        out = [MAGIC]
        copyright_ = ('\nCommon Data Format (CDF)\nlaid out by auromat_amd.export._cdf3 after the CDF Internal Format '
                      'Description; not written by the CDF library.\n').encode('ascii')
        out.append(struct.pack('>qiqiiiiiiiii', 312, _CDR, addr[('gdr',)], 3, 6, IBMPC_ENCODING, 0b11, 0, 0, 4, 2, -1)
                   + copyright_ + b'\0' * (256 - len(copyright_)))
        n_attr, n_var = len(attr_names), len(names)
        out.append(struct.pack('>qiqqqqiiiiiqiii', 84, _GDR, 0, addr[('vdr', 0)] if n_var else 0,
                               addr[('adr', 0)] if n_attr else 0, eof, 0, n_attr, -1, 0, n_var, 0, 0, _LEAP_TABLE_DATE, -1))
        for num, (k, scope) in enumerate(attr_names):
            ent = attr_entries[num]
            nxt = addr[('adr', num + 1)] if num + 1 < n_attr else 0
            head = addr[('aedr', num, 0)] if ent else 0
            top = max([i for i, _ in ent]) if ent else -1
            if scope == 1:
                out.append(struct.pack('>qiqqiiiiiqiii', 324, _ADR, nxt, head, 1, num, len(ent), top, 0, 0, 0, -1, -1) + name256(k))
            else:
                out.append(struct.pack('>qiqqiiiiiqiii', 324, _ADR, nxt, 0, 2, num, 0, -1, 0, head, len(ent), top, -1) + name256(k))
            for j, (owner, (t, n, raw)) in enumerate(ent):
                nx = addr[('aedr', num, j + 1)] if j + 1 < len(ent) else 0
                out.append(struct.pack('>qiqiiiiiiiii', 56 + len(raw), _AGREDR if scope == 1 else _AZEDR, nx, num, t, owner, n,
                                       0, 0, 0, -1, -1) + raw)
        for vi, nm in enumerate(names):
            v = self.vars[nm]
            nblk = len(blocks[vi])
            nvxr = -(-nblk // _VXR_ENTRIES)
            flags = (1 if v.rec_vary else 0) | (4 if v.compress is not None else 0)
            nxt = addr[('vdr', vi + 1)] if vi + 1 < n_var else 0
            cpr = addr[('cpr', vi)] if v.compress is not None else -1
            out.append(struct.pack('>qiqiiqqiiiiiiiqi', 344 + 8 * len(v.dims), _ZVDR, nxt, v.type, v.records.shape[0] - 1,
                                   addr[('vxr', vi, 0)], addr[('vxr', vi, nvxr - 1)], flags, 0, 0, -1, -1, v.n_elems, vi, cpr,
                                   1 if v.compress is not None else 0)
                       + name256(nm) + struct.pack('>i', len(v.dims)) + b''.join(struct.pack('>i', d) for d in v.dims)
                       + b''.join(struct.pack('>i', -1) for _ in v.dims))
            if v.compress is not None:
                out.append(struct.pack('>qiiiii', 28, _CPR, GZIP_COMPRESSION, 0, 1, v.compress))
            for x in range(nvxr):
                mine = list(range(x * _VXR_ENTRIES, min(nblk, (x + 1) * _VXR_ENTRIES)))
                first = [blocks[vi][b][0] for b in mine] + [-1] * (_VXR_ENTRIES - len(mine))
                last = [blocks[vi][b][1] for b in mine] + [-1] * (_VXR_ENTRIES - len(mine))
                offs = [addr[('blk', vi, b)] for b in mine] + [-1] * (_VXR_ENTRIES - len(mine))
                nx = addr[('vxr', vi, x + 1)] if x + 1 < nvxr else 0
                out.append(struct.pack('>qiqii', 28 + 16 * _VXR_ENTRIES, _VXR, nx, _VXR_ENTRIES, len(mine))
                           + struct.pack('>%di' % _VXR_ENTRIES, *first) + struct.pack('>%di' % _VXR_ENTRIES, *last)
                           + struct.pack('>%dq' % _VXR_ENTRIES, *offs))
            for _, _, kind, raw in blocks[vi]:
                if kind == _VVR:
                    out.append(struct.pack('>qi', 12 + len(raw), _VVR))
                else:
                    out.append(struct.pack('>qiiq', 24 + len(raw), _CVVR, 0, len(raw)))
                out.append(raw)
        with open(path, 'wb') as f:
            for b in out:
                f.write(b)
        assert sum(len(b) for b in out) == eof


def _blocks_of(v):
    """the records of a variable as the blocks its VXR lists: [(first record, last record, record type, bytes)]"""
    if v.compress is None:
        return [(0, v.records.shape[0] - 1, _VVR, v.records.tobytes())]
    per = v.records[0].nbytes
    out = []
    for r in range(v.records.shape[0]):                      # blocking factor 1: one gzip stream per record
        z = None
        if per >= (1 << 22):
            # a large record: still ONE gzip member, its deflate blocks made side by side (export/csrc/amt_io.cpp)
            from . import _io
            z = _io.gzip_parallel(v.records[r], v.compress, _GZIP_THREADS)
        if z is None:
            c = _zlib.compressobj(v.compress, _zlib.DEFLATED, 31)
            z = c.compress(np.ascontiguousarray(v.records[r]).data) + c.flush()       # (no copy of the record)
        out.append((r, r, _CVVR, z))
    return out


class ReadVar(object):
    def __init__(self, name, cdf_type, dims, rec_vary, data, compressed, attrs):
        self.name, self.type, self.dims, self.rec_vary = name, cdf_type, dims, rec_vary
        self.data, self.compressed, self.attrs = data, compressed, attrs

    def __getitem__(self, idx):
        """pycdf's indexing: a record-varying variable is indexed by record first, a non-record-varying one holds one array"""
        a = self.data if self.rec_vary else self.data[0]
        if idx is Ellipsis:
            return a
        return a[idx]


class Reader(object):
    """Reads a single-file version-3 CDF: ``attrs`` (global; a list where an attribute has several entries), ``vars`` (name
    -> :class:`ReadVar` with ``data`` of shape (records,) + dims, times as ``datetime`` via :meth:`times`), and ``records``
    (offset, size, type of every record met — what the structural tests walk)."""

    def __init__(self, path):
        with open(path, 'rb') as f:
            self.buf = buf = f.read()
        if buf[:4] != MAGIC[:4]:
            raise ValueError('not a version-3 CDF (magic %s)' % buf[:4].hex())
        if buf[4:8] != MAGIC[4:]:
            raise NotImplementedError('whole-file compression (magic word %s)' % buf[4:8].hex())
        self.records = []
        (size, kind, gdr, self.version, self.release, self.encoding, flags, _, _, self.increment, _,
         _) = struct.unpack_from('>qiqiiiiiiiii', buf, 8)
        assert kind == _CDR, kind
        self.records.append((8, size, kind))
        self.row_major, single = bool(flags & 1), bool(flags & 2)
        if not single:
            raise NotImplementedError('multi-file CDFs')
        if self.encoding in _LITTLE:
            self.order = '<'
        elif self.encoding in _BIG:
            self.order = '>'
        else:
            raise NotImplementedError('encoding %d' % self.encoding)
        _, _, rvdr, zvdr, adr, self.eof, nr, self.n_attr, _, _, self.n_zvar = struct.unpack_from('>qiqqqqiiiii', buf, gdr)
        self.records.append((gdr, struct.unpack_from('>q', buf, gdr)[0], _GDR))
        if nr or rvdr:
            raise NotImplementedError('rVariables')
        # variables
        self.vars = OrderedDict()
        by_num = {}
        while zvdr:
            (size, kind, nxt, dtype, max_rec, vxr_head, vxr_tail, vflags, srec, _, _, _, n_elems, num, cpr,
             blocking) = struct.unpack_from('>qiqiiqqiiiiiiiqi', buf, zvdr)
            assert kind == _ZVDR, kind
            self.records.append((zvdr, size, kind))
            name = buf[zvdr + 84:zvdr + 340].split(b'\0')[0].decode('utf-8')
            ndim, = struct.unpack_from('>i', buf, zvdr + 340)
            dims = struct.unpack_from('>%di' % ndim, buf, zvdr + 344)
            varys = struct.unpack_from('>%di' % ndim, buf, zvdr + 344 + 4 * ndim)
            if any(v == 0 for v in varys):
                raise NotImplementedError('dimensions without variance')
            if srec:
                raise NotImplementedError('sparse records')
            if dtype in (CDF_CHAR, CDF_UCHAR):
                dt = np.dtype('S%d' % n_elems)
            elif dtype == CDF_EPOCH16:
                raise NotImplementedError('CDF_EPOCH16')
            else:
                dt = np.dtype(self.order + _KIND[dtype])
            per = int(np.prod(dims, dtype=np.int64)) if ndim else 1
            compressed = None
            if vflags & 4:
                csize, ckind, ctype, _, pcount = struct.unpack_from('>qiiii', buf, cpr)
                assert ckind == _CPR
                self.records.append((cpr, csize, ckind))
                if ctype != GZIP_COMPRESSION:
                    raise NotImplementedError('compression type %d' % ctype)
                compressed = struct.unpack_from('>%di' % pcount, buf, cpr + 24)[0]
            nrec = max_rec + 1
            data = np.zeros((nrec, per), dt)
            have = np.zeros(nrec, bool)
            self._walk_vxr(vxr_head, data, have, per * dt.itemsize, dt)
            if not have.all():
                raise NotImplementedError('records without data (pad values)')
            if not self.row_major and ndim > 1:
                data = data.reshape((nrec,) + tuple(reversed(dims))).transpose((0,) + tuple(range(ndim, 0, -1)))
            else:
                data = data.reshape((nrec,) + tuple(dims))
            v = ReadVar(name, dtype, tuple(dims), bool(vflags & 1), data, compressed, OrderedDict())
            self.vars[name] = by_num[num] = v
            zvdr = nxt
        # attributes
        self.attrs = OrderedDict()
        self.attr_order = []                     # every attribute's name by number (global and variable scope)
        while adr:
            (size, kind, nxt, gr_head, scope, num, n_gr, max_gr, _, z_head, n_z, max_z, _) = struct.unpack_from('>qiqqiiiiiqiii', buf, adr)
            assert kind == _ADR, kind
            self.records.append((adr, size, kind))
            name = buf[adr + 68:adr + 324].split(b'\0')[0].decode('utf-8')
            self.attr_order.append(name)
            entries = []
            for head, want in ((gr_head, _AGREDR), (z_head, _AZEDR)):
                e = head
                while e:
                    esize, ekind, enext, anum, dtype, owner, n, _, _, _, _, _ = struct.unpack_from('>qiqiiiiiiiii', buf, e)
                    assert ekind == want and anum == num, (ekind, anum)
                    self.records.append((e, esize, ekind))
                    raw = buf[e + 56:e + esize]
                    entries.append((ekind, owner, self._value(dtype, n, raw)))
                    e = enext
            if scope in (1, 3):
                vals = [v for _, _, v in sorted(entries, key=lambda t: t[1])]
                self.attrs[name] = vals[0] if len(vals) == 1 else vals
            else:
                for ekind, owner, v in entries:
                    if ekind == _AZEDR:
                        by_num[owner].attrs[name] = v
            adr = nxt

    def _value(self, dtype, n, raw):
        if dtype in (CDF_CHAR, CDF_UCHAR):
            return raw[:n].decode('utf-8')
        a = np.frombuffer(raw, np.dtype(self.order + _KIND[dtype]), n)
        if dtype == CDF_TIME_TT2000:
            out = [tt2000_to_datetime(x) for x in a]
        elif dtype == CDF_EPOCH:
            out = [epoch_to_datetime(x) for x in a]
        else:
            return a[0] if n == 1 else a.copy()
        return out[0] if n == 1 else out

    def _walk_vxr(self, vxr, data, have, rec_bytes, dt):
        buf = self.buf
        while vxr:
            size, kind, nxt, n, used = struct.unpack_from('>qiqii', buf, vxr)
            assert kind == _VXR, kind
            self.records.append((vxr, size, kind))
            first = struct.unpack_from('>%di' % n, buf, vxr + 28)
            last = struct.unpack_from('>%di' % n, buf, vxr + 28 + 4 * n)
            offs = struct.unpack_from('>%dq' % n, buf, vxr + 28 + 8 * n)
            for f, l, o in list(zip(first, last, offs))[:used]:
                bsize, bkind = struct.unpack_from('>qi', buf, o)
                if bkind == _VXR:
                    self._walk_vxr(o, data, have, rec_bytes, dt)
                    continue
                self.records.append((o, bsize, bkind))
                if bkind == _VVR:
                    raw = buf[o + 12:o + bsize]
                elif bkind == _CVVR:
                    csize, = struct.unpack_from('>q', buf, o + 16)
                    raw = _zlib.decompress(buf[o + 24:o + 24 + csize], 47)
                else:
                    raise ValueError('record type %d in a VXR' % bkind)
                nrec = l - f + 1
                data[f:l + 1] = np.frombuffer(raw, dt, nrec * data.shape[1]).reshape(nrec, -1)
                have[f:l + 1] = True
            vxr = nxt

    def __getitem__(self, name):
        return self.vars[name]

    def __contains__(self, name):
        return name in self.vars

    def times(self, name):
        v = self.vars[name]
        if v.type == CDF_TIME_TT2000:
            return [tt2000_to_datetime(x) for x in v.data.ravel()]
        if v.type == CDF_EPOCH:
            return [epoch_to_datetime(x) for x in v.data.ravel()]
        raise TypeError('%s is not a time variable' % name)
