"""
Exports mappings into NASA's CDF file format following the ISTP/IACG guidelines, one self-contained file per mapping — the
variables, attributes, record variances and compression of the reference's exporter (auromat/export/cdf.py:25-285), created
in the same order.  The reference hands everything to ``spacepy.pycdf`` / NASA's CDF library; here the container is laid
out by :mod:`auromat_amd.export._cdf3` (version-3 single-file CDF, zVariables, GZIP level 5 per variable as pycdf's
``compress=GZIP_COMPRESSION``).  Read that module's header first: **no CDF library has opened these files**.

Where pycdf infers a type from the values, this writer keeps the arrays' own types — float64 stays CDF_DOUBLE, float32
CDF_FLOAT, uint16 CDF_UINT2, the masked image int16 / int32 / int64 as the reference casts it — and gives plain Python
numbers in attributes the smallest integer type / CDF_DOUBLE (pycdf's rule); empty texts become one blank (a CDF entry
cannot be empty).  One line of the reference is not followed: its unmasked-image branch reads ``img_`` before assigning it
(export/cdf.py:230, an ``UnboundLocalError``); the evident intent — the image's own data and type, no FILLVAL — is what is
written.

Pure host code, like the netCDF exporter: any object with the attributes of ``BaseMapping`` will do.
"""
import numpy as np

from . import _cdf3
from ._nc4 import _pool
from ..coordinates.transform import northGeomagneticPoleLocation


def write(outputPath, mapping, metadata={}, includeBounds=True, includeMagCoords=True, includeGeoCoords=True, compress=True,
          useTT2000=True):
    """
    :param str outputPath:
    :param auromat_amd.mapping.mapping.BaseMapping mapping:
    :param dict metadata: additional metadata, overwrites `mapping.metadata` entries if existing; a dictionary of root
                          attributes (see http://spdf.gsfc.nasa.gov/istp_guide/gattributes.html)
    :param bool includeBounds: stores the coordinates of each pixel corner (in addition to the center)
    :param bool includeMagCoords: include geomagnetic latitude-magnetic local time coordinates
    :param bool includeGeoCoords: include geodetic coordinates
    :param bool compress: use (GZIP) compression for variables
    :param bool useTT2000: CDF_TIME_TT2000 for times (else CDF_EPOCH); needs CDF 3.4.0 or higher for reading

    The mapping's arrays are read while this call runs, partly on other threads (large variables start to compress when they
    are created): do not change them from another thread before it returns.
    """
    z = _cdf3.GZIP_COMPRESSION if compress else None
    root = _cdf3.Writer(tt2000=bool(useTT2000), pool=_pool() if compress else None)

    def nan(a):
        return np.ma.filled(a, np.nan)

    def put(var, *pairs):
        for key, value in pairs:
            var.attrs[key] = value
        return var

    # ROOT ATTRIBUTES (reference cdf.py:62-80)
    metadata = dict(list((mapping.metadata or {}).items()) + list(metadata.items()))
    for k, v in metadata.items():
        if isinstance(v, bool):
            v = int(v)
        try:
            _cdf3.infer(v)
        except TypeError:
            raise TypeError('Cannot store global attribute "{}" with value {}'.format(k, repr(v)))
        root.attrs[k] = v
    bb = mapping.boundingBox
    root.attrs['geospatial_lat_min'] = float(bb.latSouth)
    root.attrs['geospatial_lat_max'] = float(bb.latNorth)
    root.attrs['geospatial_lon_min'] = float(bb.lonWest)
    root.attrs['geospatial_lon_max'] = float(bb.lonEast)
    root.attrs['geospatial_lat_units'] = 'degrees_north'
    root.attrs['geospatial_lon_units'] = 'degrees_east'

    # VARIABLES (reference cdf.py:82-285)
    put(root.new('Epoch', [mapping.photoTime], type=_cdf3.CDF_TIME_TT2000 if useTT2000 else _cdf3.CDF_EPOCH),
        ('VAR_TYPE', 'support_data'))

    def coordinate(name, data, depend, units, lo, hi, fieldnam, notes, crs):
        v = put(root.new(name, nan(data)[np.newaxis, :], compress=z), ('VAR_TYPE', 'data'), ('DEPEND_0', 'Epoch'),
                ('DEPEND_1', 'y_' + depend), ('DEPEND_2', 'x_' + depend), ('UNITS', units), ('VALIDMIN', lo), ('VALIDMAX', hi),
                ('FIELDNAM', fieldnam))
        if notes is not None:
            put(v, ('VAR_NOTES', notes))
        return put(v, ('crs', crs))

    if includeGeoCoords:
        lat = coordinate('lat', mapping.latsCenter, 'pixel', 'degrees', -90.0, 90.0, 'Latitude of pixel center',
                         'Geodetic latitude', 'crs')
        lon = coordinate('lon', mapping.lonsCenter, 'pixel', 'degrees', -180.0, 180.0, 'Longitude of pixel center',
                         'Geodetic longitude', 'crs')
        if includeBounds:
            lat.attrs['bounds'] = 'lat_bounds'
            lon.attrs['bounds'] = 'lon_bounds'
            coordinate('lat_bounds', mapping.lats, 'corner', 'degrees', -90.0, 90.0, 'Latitude of pixel corner',
                       'Geodetic latitude', 'crs')
            coordinate('lon_bounds', mapping.lons, 'corner', 'degrees', -180.0, 180.0, 'Longitude of pixel corner',
                       'Geodetic longitude', 'crs')

    put(root.new('altitude', float(mapping.altitude * 1000), recVary=False), ('VAR_TYPE', 'support_data'), ('UNITS', 'meters'),
        ('FIELDNAM', 'Height above reference ellipsoid'), ('crs', 'crs'))

    if includeMagCoords:
        mlats, mlts = mapping.mLatMltCenter
        mlat = coordinate('mlat', mlats, 'pixel', 'degrees', -90.0, 90.0, 'Geomagnetic latitude of pixel center', '', 'mcrs')
        # (the reference names mlt's dependencies y_center / x_center, the other center arrays' y_pixel / x_pixel: kept)
        mlt = coordinate('mlt', mlts, 'center', 'hours', 0.0, 24.0, 'Magnetic local time of pixel center', None, 'mcrs')
        if includeBounds:
            mlat.attrs['bounds'] = 'mlat_bounds'
            mlt.attrs['bounds'] = 'mlt_bounds'
            mlats, mlts = mapping.mLatMlt
            coordinate('mlat_bounds', mlats, 'corner', 'degrees', -90.0, 90.0, 'Geomagnetic latitude of pixel corner', '', 'mcrs')
            coordinate('mlt_bounds', mlts, 'corner', 'hours', 0.0, 24.0, 'Magnetic local time of pixel corner', None, 'mcrs')
        magPoleLat, magPoleLon = northGeomagneticPoleLocation(mapping.photoTime)
        put(root.new('mcrs', 0, recVary=False),                 # (holds no data: a carrier of attributes)
            ('VAR_TYPE', 'support_data'), ('north_geomagnetic_pole_lat', float(magPoleLat)),
            ('north_geomagnetic_pole_lon', float(magPoleLon)),
            ('VAR_NOTES', 'Geocentric MLat/MLT system based on the given geomagnetic pole position'))

    img = mapping.img
    if np.any(np.ma.getmaskarray(img)):
        # CDF supports much more types than netCDF: the next wider signed type holds the image and a fill value
        imgDtypeMap = {np.dtype(np.uint8): np.int16, np.dtype(np.uint16): np.int32, np.dtype(np.uint32): np.int64}
        if img.dtype not in imgDtypeMap:
            raise NotImplementedError('Image data type not supported: ' + str(img.dtype))
        imgDtype = imgDtypeMap[img.dtype]
        imgFillval = imgDtype(np.iinfo(imgDtype).min)
        img_ = img.astype(imgDtype).filled(imgFillval)
    else:
        img_ = np.ma.getdata(img)
        imgFillval = None
    if img_.shape[2] == 1:
        bands = ['img']
    elif img_.shape[2] == 3:
        bands = ['img_red', 'img_green', 'img_blue']
    else:
        raise NotImplementedError
    for i, band in enumerate(bands):
        v = put(root.new(band, img_[np.newaxis, :, :, i], compress=z), ('VAR_TYPE', 'data'), ('DEPEND_0', 'Epoch'),
                ('DEPEND_1', 'y_pixel'), ('DEPEND_2', 'x_pixel'), ('FIELDNAM', ''), ('VALIDMIN', int(np.iinfo(img.dtype).min)),
                ('VALIDMAX', int(np.iinfo(img.dtype).max)))
        if imgFillval:
            put(v, ('FILLVAL', imgFillval))
        put(v, ('UNITS', 'unitless'))

    zen = 90 - nan(mapping.elevation)[np.newaxis, :].astype(np.float32)
    put(root.new('zenith_angle', zen, compress=z), ('VAR_TYPE', 'data'), ('DEPEND_0', 'Epoch'), ('DEPEND_1', 'y_pixel'),
        ('DEPEND_2', 'x_pixel'), ('UNITS', 'degrees'), ('VALIDMIN', 0.0), ('VALIDMAX', 90.0),
        ('FIELDNAM', 'Absolute sensor zenith angle of pixel center'))
    put(root.new('camera_pos', np.array([mapping.cameraPosGCRS], np.float64)), ('VAR_TYPE', 'support_data'), ('DEPEND_0', 'Epoch'),
        ('UNITS', 'kilometers'), ('FIELDNAM', 'Camera position in cartesian GCRS coordinates'), ('VAR_NOTES', 'Axis order: xyz'))
    put(root.new('crs', 0, recVary=False),                      # (holds no data: a carrier of attributes)
        ('VAR_TYPE', 'support_data'), ('semi_major_axis', 6378137.0), ('inverse_flattening', 298.257223563),
        ('VAR_NOTES', 'Geographic Coordinate System, WGS 84'))

    root.write(outputPath)
