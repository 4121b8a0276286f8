"""
Reads CDF files written by :mod:`auromat_amd.export.cdf` back as mappings (reference auromat/mapping/cdf.py:19-198, which
opens them with ``spacepy.pycdf``; here :class:`auromat_amd.export._cdf3.Reader` parses the version-3 single-file container:
zVariables, GZIP-compressed or plain, either byte order).  Files of NASA's CDF library that stay inside that subset — those of
the reference's exporter do: pycdf creates zVariables only — follow the same format description, but none was available
to try (see ``_cdf3``'s header).
File parsing is host code; the mapping it returns is a :class:`GenericMapping` (device-resident like all others).
"""
import collections
import os.path

import numpy as np
import numpy.ma as ma

from ..export import _cdf3
from .mapping import BaseMappingProvider, GenericMapping


def read_arrays(path, i=0):
    """-> dict(lats, lons, latsCenter, lonsCenter, elevation, img, altitude, cameraPosGCRS, photoTime, metadata) of record
    ``i`` of an exported file, as the reference's ``CDFMapping.__init__`` assembles them (mapping/cdf.py:97-134)."""
    root = _cdf3.Reader(path)
    var = root.vars
    altitude = var['altitude'][...] / 1000
    cameraPosGCRS = np.array(var['camera_pos'][i])
    photoTime = root.times('Epoch')[i]
    if 'img' in var:
        img = _convertImgDtype(var['img'][i], var['img'].attrs.get('FILLVAL'))
        img = img[:, :, None] if img.ndim == 2 else img
    else:
        bands = [_convertImgDtype(var[k][i], var[k].attrs.get('FILLVAL')) for k in ('img_red', 'img_green', 'img_blue')]
        img = ma.dstack(bands)
    latsCenter, lonsCenter = var['lat'][i], var['lon'][i]
    lats = var[var['lat'].attrs['bounds']][i]
    lons = var[var['lon'].attrs['bounds']][i]
    assert var['altitude'].attrs['UNITS'] == 'meters'
    assert var['camera_pos'].attrs['UNITS'] == 'kilometers'
    return dict(lats=ma.masked_invalid(lats), lons=ma.masked_invalid(lons), latsCenter=ma.masked_invalid(latsCenter),
                lonsCenter=ma.masked_invalid(lonsCenter),
                elevation=ma.masked_invalid(90 - var['zenith_angle'][i].astype(np.float64)), img=img,
                altitude=float(altitude), cameraPosGCRS=cameraPosGCRS, photoTime=photoTime,
                metadata=collections.OrderedDict(root.attrs))


class CDFMapping(GenericMapping):
    def __init__(self, cdfPath, i=0):
        a = read_arrays(cdfPath, i)
        identifier = os.path.splitext(os.path.basename(cdfPath))[0]
        GenericMapping.__init__(self, a['lats'], a['lons'], a['latsCenter'], a['lonsCenter'], a['elevation'], a['altitude'],
                                a['img'], a['cameraPosGCRS'], a['photoTime'], identifier, metadata=a['metadata'])


class CDFMappingProvider(BaseMappingProvider):
    """Mappings from a list of exported files, looked up by date: (file, record) per date (reference mapping/cdf.py:19-77)."""

    def __init__(self, cdfPaths, maxTimeOffset=3):
        BaseMappingProvider.__init__(self, maxTimeOffset=maxTimeOffset)
        self.cdfPaths = cdfPaths
        datemap = {}
        for path_idx, path in enumerate(cdfPaths):
            for cdf_idx, date in enumerate(_cdf3.Reader(path).times('Epoch')):
                if date in datemap:
                    raise ValueError('The date ' + str(date) + ' is appearing twice in the CDF files ' + path + ' and ' +
                                     cdfPaths[datemap[date][0]])
                datemap[date] = (path_idx, cdf_idx)
        self.datemap = collections.OrderedDict(sorted(datemap.items()))

    def __len__(self):
        return len(self.datemap)

    @property
    def range(self):
        return list(self.datemap.keys())[0], list(self.datemap.keys())[-1]

    def _nearest(self, date):
        dates = list(self.datemap.keys())
        idx = int(np.argmin([abs((d - date).total_seconds()) for d in dates]))
        return dates[idx], abs((dates[idx] - date).total_seconds())

    def contains(self, date):
        return self._nearest(date)[1] <= self.maxTimeOffset

    def get(self, date):
        found, offset = self._nearest(date)
        if offset > self.maxTimeOffset:
            raise ValueError('Closest mapping found at ' + str(found) + ' but offset > ' + str(self.maxTimeOffset) +
                             ' seconds, requested: ' + str(date))
        path_idx, cdf_idx = self.datemap[found]
        return CDFMapping(self.cdfPaths[path_idx], cdf_idx)

    def getById(self, identifier):
        raise NotImplementedError

    def getSequence(self, dateBegin=None, dateEnd=None):
        if not dateBegin:
            dateBegin = self.range[0]
        if not dateEnd:
            dateEnd = self.range[1]
        for date in [d for d in self.datemap if dateBegin <= d <= dateEnd]:
            path_idx, cdf_idx = self.datemap[date]
            yield CDFMapping(self.cdfPaths[path_idx], cdf_idx)


def _convertImgDtype(arr, fillval):
    if arr.dtype in [np.uint8, np.uint16, np.uint32]:
        return ma.masked_array(arr)
    arr = ma.masked_equal(arr, fillval, copy=False) if fillval is not None else ma.masked_array(arr)
    for signed, unsigned in ((np.int16, np.uint8), (np.int32, np.uint16), (np.int64, np.uint32)):
        if arr.dtype == signed:
            assert 0 <= np.min(arr) <= np.max(arr) <= np.iinfo(unsigned).max
            return arr.astype(unsigned)
    raise NotImplementedError('Data type not supported: ' + str(arr.dtype))
