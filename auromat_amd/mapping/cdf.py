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
from ._catalogue import DateCatalogue, unsigned_pixels
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
        img = unsigned_pixels(var['img'][i], var['img'].attrs.get('FILLVAL'))
        img = img[:, :, None] if img.ndim == 2 else img
    else:
        bands = [unsigned_pixels(var[k][i], var[k].attrs.get('FILLVAL')) for k in ('img_red', 'img_green', 'img_blue')]
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
    """Mappings from a list of exported CDF files, looked up by date; a file may hold several records (reference
    mapping/cdf.py:19-77)."""

    def __init__(self, cdfPaths, maxTimeOffset=3):
        BaseMappingProvider.__init__(self, maxTimeOffset=maxTimeOffset)
        self.cdfPaths = list(cdfPaths)
        self.catalogue = DateCatalogue(((date, (path, record), path) for path in self.cdfPaths
                                        for record, date in enumerate(_cdf3.Reader(path).times('Epoch'))), 'the list of CDF files')

    def __len__(self):
        return len(self.catalogue)

    @property
    def range(self):
        return self.catalogue.span

    def contains(self, date):
        return self.catalogue.within(date, self.maxTimeOffset)

    def get(self, date):
        return CDFMapping(*self.catalogue.pick(date, self.maxTimeOffset))

    def getById(self, identifier):
        raise NotImplementedError

    def getSequence(self, dateBegin=None, dateEnd=None):
        for path, record in self.catalogue.between(dateBegin, dateEnd):
            yield CDFMapping(path, record)
