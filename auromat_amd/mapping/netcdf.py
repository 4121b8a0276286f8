"""
Reads files written by :mod:`auromat_amd.export.netcdf` — netCDF-4 as laid out by :mod:`auromat_amd.export._nc4` or netCDF
classic — back as mappings (reference auromat/mapping/netcdf.py).  Files of the reference's exporter come from the netCDF
library, whose HDF5 structures (version-2 object headers) are not parsed here: convert them to the classic format with
``nccopy -k nc6`` (= ``-k 64-bit-offset``, CDF-2) or ``-k classic`` (CDF-1); CDF-5 is not read.
File parsing is host code; the mapping it returns is a :class:`GenericMapping` (device-resident like all others).
"""
import collections
import os.path
from datetime import datetime, timedelta

import numpy as np
import numpy.ma as ma

from ..export import _nc4
from .mapping import BaseMappingProvider, GenericMapping


def read_arrays(path):
    """
    -> dict(lats, lons, latsCenter, lonsCenter, elevation, img, altitude, cameraPosGCRS, photoTime, metadata) of one
    exported mapping, as the reference's ``NetCDFMapping.__init__`` assembles them (mapping/netcdf.py:96-157): corner
    grids rebuilt from the cell bounds, images back to their unsigned type, elevation = 90 - zenith angle.
    """
    f = _nc4.open_file(path)
    var = f.vars
    altitude = var['altitude'].data / 1000
    cameraPosGCRS = np.array(var['camera_pos'].data)
    photoTime = _readDate(var['time'])
    if 'img' in var:
        img = _convertImgDtype(_masked_fill(var['img']))[:, :, None]
    else:
        img = ma.dstack([_convertImgDtype(_masked_fill(var[k])) for k in ('img_red', 'img_green', 'img_blue')])
    latsCenter, lonsCenter = var['lat'].data, var['lon'].data
    latBounds = var[var['lat'].attrs['bounds']].data
    lonBounds = var[var['lon'].attrs['bounds']].data
    if latsCenter.ndim == 1:
        latsCenter, lonsCenter = np.dstack(np.meshgrid(latsCenter, lonsCenter)).T
        assert np.all(latBounds[:-1, 1] == latBounds[1:, 0])
        assert np.all(lonBounds[:-1, 1] == lonBounds[1:, 0])
        latBounds = np.concatenate((latBounds[:, 0], [latBounds[-1, 1]]))
        lonBounds = np.concatenate((lonBounds[:, 0], [lonBounds[-1, 1]]))
        lats, lons = np.dstack(np.meshgrid(latBounds, lonBounds)).T
    else:
        lats = np.empty((latsCenter.shape[0] + 1, latsCenter.shape[1] + 1), latBounds.dtype)
        lons = np.empty_like(lats)
        for grid, bounds in [(lats, latBounds), (lons, lonBounds)]:
            np.testing.assert_array_equal(bounds[:-1, :-1, 2], bounds[:-1, 1:, 3])
            np.testing.assert_array_equal(bounds[:-1, :-1, 2], bounds[1:, 1:, 0])
            np.testing.assert_array_equal(bounds[:-1, :-1, 2], bounds[1:, :-1, 1])
            grid[:-1, :-1] = bounds[:, :, 0]
            grid[-1, :-1] = bounds[-1, :, 3]
            grid[:-1, -1] = bounds[:, -1, 1]
            grid[-1, -1] = bounds[-1, -1, 2]
    assert var['altitude'].attrs['units'] == 'meters'
    assert var['camera_pos'].attrs['units'] == 'kilometers'
    return dict(lats=ma.masked_invalid(lats), lons=ma.masked_invalid(lons), latsCenter=ma.masked_invalid(latsCenter),
                lonsCenter=ma.masked_invalid(lonsCenter),
                elevation=ma.masked_invalid(90 - var['zenith_angle'].data.astype(np.float64)), img=img,
                altitude=float(altitude), cameraPosGCRS=cameraPosGCRS, photoTime=photoTime,
                metadata=collections.OrderedDict(f.attrs))


class NetCDFMapping(GenericMapping):
    def __init__(self, cdfPath):
        a = read_arrays(cdfPath)
        identifier = os.path.splitext(os.path.basename(cdfPath))[0]
        GenericMapping.__init__(self, a['lats'], a['lons'], a['latsCenter'], a['lonsCenter'], a['elevation'], a['altitude'],
                                a['img'], a['cameraPosGCRS'], a['photoTime'], identifier, metadata=a['metadata'])


class NetCDFMappingProvider(BaseMappingProvider):
    """Mappings from a list of exported files, looked up by date (reference mapping/netcdf.py:20-76)."""

    def __init__(self, cdfPaths, maxTimeOffset=3):
        BaseMappingProvider.__init__(self, maxTimeOffset=maxTimeOffset)
        self.cdfPaths = cdfPaths
        datemap = {}
        for path_idx, path in enumerate(cdfPaths):
            date = _readDate(_nc3.File(path).vars['time'])
            if date in datemap:
                raise ValueError('The date ' + str(date) + ' is appearing twice in the NetCDF files ' + path + ' and ' +
                                 cdfPaths[datemap[date]])
            datemap[date] = path_idx
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
        return NetCDFMapping(self.cdfPaths[self.datemap[found]])

    def getById(self, identifier):
        raise NotImplementedError

    def getSequence(self, dateBegin=None, dateEnd=None):
        if not dateBegin:
            dateBegin = self.range[0]
        if not dateEnd:
            dateEnd = self.range[1]
        for date in [d for d in self.datemap if dateBegin <= d <= dateEnd]:
            yield NetCDFMapping(self.cdfPaths[self.datemap[date]])


def _masked_fill(v):
    fill = v.attrs.get('_FillValue')
    return ma.masked_equal(v.data, fill) if fill is not None else ma.masked_array(v.data)


def _convertImgDtype(arr):
    if arr.dtype in [np.uint8, np.uint16]:
        return arr
    elif arr.dtype == np.int16:
        assert 0 <= np.min(arr) <= np.max(arr) <= np.iinfo(np.uint8).max
        return arr.astype(np.uint8)
    elif arr.dtype == np.int32:
        assert 0 <= np.min(arr) <= np.max(arr) <= np.iinfo(np.uint16).max
        return arr.astype(np.uint16)
    raise NotImplementedError('Data type not supported: ' + str(arr.dtype))


def _readDate(date_var):
    assert date_var.attrs['units'] == 'seconds since 1970-01-01 00:00:00'
    return datetime(1970, 1, 1) + timedelta(seconds=float(date_var.data))
