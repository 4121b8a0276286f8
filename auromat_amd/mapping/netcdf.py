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
from ._catalogue import DateCatalogue, unsigned_pixels
from .mapping import BaseMappingProvider, GenericMapping


def _edges_from_intervals(bounds, name):
    """(n, 2) cell intervals of a rectilinear axis -> the n + 1 edges; neighbouring cells must share their edge."""
    if not np.array_equal(bounds[1:, 0], bounds[:-1, 1]):
        raise ValueError('the %s bounds are not contiguous' % name)
    return np.append(bounds[:, 0], bounds[-1, 1])


def _corner_grid(bounds, name):
    """(ny, nx, 4) cell vertices — upper left, upper right, lower right, lower left, the order the exporter writes — -> the
    (ny + 1, nx + 1) grid of corners, after checking that the four cells around every corner agree on it (NaN = missing corner)."""
    ny, nx = bounds.shape[:2]
    grid = np.empty((ny + 1, nx + 1), bounds.dtype)
    grid[:ny, :nx] = bounds[..., 0]
    grid[:ny, nx] = bounds[:, -1, 1]
    grid[ny, :nx] = bounds[-1, :, 3]
    grid[ny, nx] = bounds[-1, -1, 2]
    views = {1: grid[:ny, 1:], 2: grid[1:, 1:], 3: grid[1:, :nx]}
    for vertex, view in views.items():
        if not np.array_equal(view, bounds[..., vertex], equal_nan=True):
            raise ValueError('the %s bounds of neighbouring cells disagree about their common corners' % name)
    return grid


def read_arrays(path):
    """
    -> dict(lats, lons, latsCenter, lonsCenter, elevation, img, altitude, cameraPosGCRS, photoTime, metadata) of one exported
    mapping (what the reference's ``NetCDFMapping`` holds, mapping/netcdf.py:96-157): corner grids rebuilt from the CF cell
    bounds — intervals per axis for a plate carree mapping, four vertices per cell otherwise —, images back to their unsigned
    pixel type, elevation = 90 - zenith angle, altitude in km.
    """
    f = _nc4.open_file(path)
    var = f.vars
    for name, unit in (('altitude', 'meters'), ('camera_pos', 'kilometers')):
        if var[name].attrs['units'] != unit:
            raise ValueError('%s in %s, expected %s' % (name, var[name].attrs['units'], unit))
    pixels = lambda v: unsigned_pixels(v.data, v.attrs.get('_FillValue'))
    if 'img' in var:
        img = pixels(var['img'])[:, :, None]
    else:
        img = ma.dstack([pixels(var[k]) for k in ('img_red', 'img_green', 'img_blue')])
    centres = {k: var[k].data for k in ('lat', 'lon')}
    bounds = {k: var[var[k].attrs['bounds']].data for k in ('lat', 'lon')}
    if centres['lat'].ndim == 1:
        # plate carree: one axis each; every array is (n_lat, n_lon)
        latsCenter, lonsCenter = np.meshgrid(centres['lat'], centres['lon'], indexing='ij')
        lats, lons = np.meshgrid(_edges_from_intervals(bounds['lat'], 'latitude'), _edges_from_intervals(bounds['lon'], 'longitude'),
                                 indexing='ij')
    else:
        latsCenter, lonsCenter = centres['lat'], centres['lon']
        lats, lons = _corner_grid(bounds['lat'], 'latitude'), _corner_grid(bounds['lon'], 'longitude')
    return dict(lats=ma.masked_invalid(lats), lons=ma.masked_invalid(lons), latsCenter=ma.masked_invalid(latsCenter),
                lonsCenter=ma.masked_invalid(lonsCenter),
                elevation=ma.masked_invalid(90 - var['zenith_angle'].data.astype(np.float64)), img=img,
                altitude=float(var['altitude'].data) / 1000, cameraPosGCRS=np.array(var['camera_pos'].data),
                photoTime=_readDate(var['time']), metadata=collections.OrderedDict(f.attrs))


class NetCDFMapping(GenericMapping):
    def __init__(self, cdfPath):
        a = read_arrays(cdfPath)
        identifier = os.path.splitext(os.path.basename(cdfPath))[0]
        GenericMapping.__init__(self, a['lats'], a['lons'], a['latsCenter'], a['lonsCenter'], a['elevation'], a['altitude'],
                                a['img'], a['cameraPosGCRS'], a['photoTime'], identifier, metadata=a['metadata'])


class NetCDFMappingProvider(BaseMappingProvider):
    """Mappings from a list of exported netCDF files (one mapping each), looked up by date (reference mapping/netcdf.py:20-76)."""

    def __init__(self, cdfPaths, maxTimeOffset=3):
        BaseMappingProvider.__init__(self, maxTimeOffset=maxTimeOffset)
        self.cdfPaths = list(cdfPaths)
        self.catalogue = DateCatalogue(((_readDate(_nc4.open_file(path).vars['time']), path, path) for path in self.cdfPaths),
                                       'the list of netCDF files')

    def __len__(self):
        return len(self.catalogue)

    @property
    def range(self):
        return self.catalogue.span

    def contains(self, date):
        return self.catalogue.within(date, self.maxTimeOffset)

    def get(self, date):
        return NetCDFMapping(self.catalogue.pick(date, self.maxTimeOffset))

    def getById(self, identifier):
        raise NotImplementedError

    def getSequence(self, dateBegin=None, dateEnd=None):
        for path in self.catalogue.between(dateBegin, dateEnd):
            yield NetCDFMapping(path)


def _readDate(date_var):
    assert date_var.attrs['units'] == 'seconds since 1970-01-01 00:00:00'
    return datetime(1970, 1, 1) + timedelta(seconds=float(date_var.data))
