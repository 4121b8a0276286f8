"""
Exports mappings into the netCDF file format following the CF 1.6 and NODC conventions, one self-contained file per
mapping — the layout of the reference's exporter (auromat/export/netcdf.py:24-386): same dimensions, variables,
data types, ``_FillValue`` s and attributes, created in the same order, in the same container: netCDF-4 (an HDF5 file laid
out by :mod:`auromat_amd.export._nc4`, zlib-compressed in the reference's chunks) or, on request, netCDF classic with 64-bit
offsets (:mod:`auromat_amd.export._nc3`: no compression, no chunking).

Pure host code: it takes any object with the attributes of ``BaseMapping`` (masked NumPy arrays), so the small
resampled grids of the frame pipeline go to disk without ever materialising per-pixel arrays on the host.
"""
from datetime import datetime

import numpy as np

from . import _nc3, _nc4
from ..coordinates.transform import northGeomagneticPoleLocation
from ..mapping.mapping import isPlateCarree


def _with_range(var, masked):
    """sets ``actual_range`` = [min, max] over the unmasked values (reference netcdf.py:114-126: np.min / np.max of the masked
    array) and returns the array with NaN where masked — min and max taken from that one NaN-filled copy (a masked-array
    reduction of 12 M values costs 13 ms, the NaN-skipping one 3)"""
    filled = np.ma.filled(masked, np.nan)
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            var.attrs['actual_range'] = np.float64([np.nanmin(filled), np.nanmax(filled)])
    return filled


def write(outputPath, mapping, metadata={}, includeBounds=True, includeMagCoords=True, includeGeoCoords=True,
          use1dIfPossible=True, compress=True, format='NETCDF4'):
    """
    :param str outputPath:
    :param auromat_amd.mapping.mapping.BaseMapping mapping:
    :param dict metadata: additional metadata, overwrites mapping.metadata entries if existing; a dictionary of root
                          attributes (see http://wiki.esipfed.org/index.php/Attribute_Convention_for_Data_Discovery)
    :param bool includeBounds: stores the coordinates of each pixel corner (in addition to the center)
    :param bool includeMagCoords: include geomagnetic latitude-magnetic local time coordinates
    :param bool compress: zlib-compress the arrays (netCDF-4 only; deflate level 4 behind the byte shuffle, the defaults of
                          the netCDF4 library the reference calls with ``zlib=compress``)
    :param str format: 'NETCDF4' (the reference's container: HDF5, chunked like the reference's files — one row of the arrays
                       per chunk —, written by :mod:`auromat_amd.export._nc4`) or 'NETCDF3_64BIT' (classic format with 64-bit
                       offsets, :mod:`auromat_amd.export._nc3`: no compression, no chunking; what
                       :class:`auromat_amd.mapping.netcdf.NetCDFMapping` reads)

    The mapping's arrays are read while this call runs, partly on other threads (large arrays start to compress as soon as
    they have been handed to the writer): do not change them from another thread before it returns.  ``AMT_IO_THREADS`` (or
    ``AMT_NC4_THREADS``) bounds the native threads of a process, whatever the number of files being written at once.
    """
    if not includeGeoCoords:
        raise ValueError('Geodetic coordinates cannot be disabled for netCDF as they are essential to the format')
    if format not in ('NETCDF4', 'NETCDF3_64BIT'):
        raise ValueError("format must be 'NETCDF4' or 'NETCDF3_64BIT'")
    root = _nc4.Writer() if format == 'NETCDF4' else _nc3.Writer()
    z = bool(compress)
    # ROOT ATTRIBUTES (reference netcdf.py:50-72)
    root.attrs['Conventions'] = 'CF-1.6'
    metadata = dict(list((mapping.metadata or {}).items()) + list(metadata.items()))
    for k, v in metadata.items():
        if isinstance(v, bool):
            v = np.uint8(v)
        try:
            _nc3._attr_values(v)
            _nc4.check_attribute(v)
        except TypeError:
            raise TypeError('Cannot store global attribute "{}" with value {}'.format(k, repr(v)))
        root.attrs[k] = v

    isLatLonPlateCarree = use1dIfPossible and isPlateCarree(mapping.lats, mapping.lons)
    isMLatMltPlateCarree = (use1dIfPossible and isPlateCarree(*mapping.mLatMlt)) if includeMagCoords else None

    bb = mapping.boundingBox
    root.attrs['geospatial_lat_min'] = np.float64(bb.latSouth)
    root.attrs['geospatial_lat_max'] = np.float64(bb.latNorth)
    root.attrs['geospatial_lon_min'] = np.float64(bb.lonWest)
    root.attrs['geospatial_lon_max'] = np.float64(bb.lonEast)
    root.attrs['geospatial_lat_units'] = 'degrees_north'
    root.attrs['geospatial_lon_units'] = 'degrees_east'

    # DIMENSIONS (reference netcdf.py:74-93)
    w, h = mapping.img.shape[1], mapping.img.shape[0]
    if isLatLonPlateCarree:
        root.create_dimension('lats', h)
        root.create_dimension('lons', w)
    if isMLatMltPlateCarree:
        root.create_dimension('mlats', h)
        root.create_dimension('mlts', w)
    if not isLatLonPlateCarree or isMLatMltPlateCarree is False:        # "is False": None means no magnetic coordinates
        root.create_dimension('y', h)
        root.create_dimension('x', w)
    if includeBounds:
        if isLatLonPlateCarree or isMLatMltPlateCarree:
            root.create_dimension('vertex2', 2)         # for 1D coordinate arrays
        if not isLatLonPlateCarree or isMLatMltPlateCarree is False:
            root.create_dimension('vertex4', 4)         # for 2D coordinate arrays
    root.create_dimension('channel', mapping.img.shape[2])
    root.create_dimension('xyz', 3)

    # COORDINATE VARIABLES (reference netcdf.py:95-180); time as POSIX seconds, missing float values as NaN
    time = root.create_variable('time', np.float64)
    time.attrs['units'] = 'seconds since 1970-01-01 00:00:00'
    time.attrs['calendar'] = 'gregorian'
    time.attrs['standard_name'] = 'time'
    time.attrs['axis'] = 'T'
    time.attrs['long_name'] = ''
    time.attrs['comment'] = ''
    time.set(_unix(mapping.photoTime))

    if isLatLonPlateCarree:
        # the unmasked arrays: CF coordinate arrays have no missing values
        latsCenter, lonsCenter = mapping.latsCenter.data[:, 0], mapping.lonsCenter.data[0, :]
        lat = root.create_variable('lat', np.float64, ('lats',), zlib=z)
        lat.attrs['actual_range'] = np.float64([latsCenter[-1], latsCenter[0]])
        lat.set(latsCenter)
        lon = root.create_variable('lon', np.float64, ('lons',), zlib=z)
        lon.attrs['actual_range'] = np.float64([lonsCenter[0], lonsCenter[-1]])
        lon.set(lonsCenter)
    else:
        # auxiliary 2D coordinate variables with missing values (a documented deviation of the reference from CF 1.6)
        lat = root.create_variable('lat', np.float64, ('y', 'x'), zlib=z, chunksizes=(1, w))
        lat.set(_with_range(lat, mapping.latsCenter))
        lon = root.create_variable('lon', np.float64, ('y', 'x'), zlib=z, chunksizes=(1, w))
        lon.set(_with_range(lon, mapping.lonsCenter))

    lat.attrs['units'] = 'degrees_north'
    lat.attrs['valid_min'] = np.float64(-90)
    lat.attrs['valid_max'] = np.float64(90)
    lat.attrs['standard_name'] = 'latitude'
    lat.attrs['axis'] = 'Y'
    lat.attrs['long_name'] = 'Latitude'
    lat.attrs['comment'] = 'Geodetic latitude'

    lon.attrs['units'] = 'degrees_east'
    lon.attrs['valid_min'] = np.float64(-180)
    lon.attrs['valid_max'] = np.float64(180)
    lon.attrs['standard_name'] = 'longitude'
    lon.attrs['axis'] = 'X'
    lon.attrs['long_name'] = 'Longitude'
    lon.attrs['comment'] = 'Geodetic longitude'

    altitude = root.create_variable('altitude', np.int32)
    altitude.attrs['units'] = 'meters'
    altitude.attrs['standard_name'] = 'height_above_reference_ellipsoid'
    altitude.attrs['axis'] = 'Z'
    altitude.attrs['long_name'] = ''
    altitude.set(mapping.altitude * 1000)

    if includeBounds:
        lat.attrs['bounds'] = 'lat_bounds'
        lon.attrs['bounds'] = 'lon_bounds'
        if isLatLonPlateCarree:
            root.create_variable('lat_bounds', np.float64, ('lats', 'vertex2'), zlib=z, chunksizes=(h, 2)).set(_bounds1d(mapping.lats.data[:, 0]))
            root.create_variable('lon_bounds', np.float64, ('lons', 'vertex2'), zlib=z, chunksizes=(w, 2)).set(_bounds1d(mapping.lons.data[0, :]))
        else:
            root.create_variable('lat_bounds', np.float64, ('y', 'x', 'vertex4'), zlib=z, chunksizes=(1, w, 4)).set(_bounds2d(mapping.lats.filled(np.nan)))
            root.create_variable('lon_bounds', np.float64, ('y', 'x', 'vertex4'), zlib=z, chunksizes=(1, w, 4)).set(_bounds2d(mapping.lons.filled(np.nan)))

    if includeMagCoords:
        # non-standard: CF 1.6 has no convention for coordinates in a second system (reference netcdf.py:203-277)
        mlats, mlts = mapping.mLatMltCenter
        if isMLatMltPlateCarree:
            mlatsCenter, mltsCenter = mlats.data[:, 0], mlts.data[0, :]
            mlat = root.create_variable('mlat', np.float64, ('mlats',), zlib=z)
            mlat.attrs['actual_range'] = np.float64([mlatsCenter[-1], mlatsCenter[0]])
            mlat.set(mlatsCenter)
            mlt = root.create_variable('mlt', np.float64, ('mlts',), zlib=z)
            mlt.attrs['actual_range'] = np.float64([mltsCenter[0], mltsCenter[-1]])
            mlt.set(mltsCenter)
        else:
            mlat = root.create_variable('mlat', np.float64, ('y', 'x'), zlib=z, chunksizes=(1, w))
            mlat.set(_with_range(mlat, mlats))
            mlt = root.create_variable('mlt', np.float64, ('y', 'x'), zlib=z, chunksizes=(1, w))
            mlt.set(_with_range(mlt, mlts))
        mlat.attrs['long_name'] = 'Geomagnetic latitude'
        mlat.attrs['units'] = 'degrees'
        mlat.attrs['valid_min'] = np.float64(-90)
        mlat.attrs['valid_max'] = np.float64(90)
        mlat.attrs['crs'] = 'mcrs'
        mlt.attrs['long_name'] = 'Magnetic local time'
        mlt.attrs['units'] = 'hours'
        mlt.attrs['valid_min'] = np.float64(0)
        mlt.attrs['valid_max'] = np.float64(24)
        mlt.attrs['crs'] = 'mcrs'
        if includeBounds:
            mlat.attrs['bounds'] = 'mlat_bounds'
            mlt.attrs['bounds'] = 'mlt_bounds'
            mlats, mlts = mapping.mLatMlt
            if isMLatMltPlateCarree:
                root.create_variable('mlat_bounds', np.float64, ('mlats', 'vertex2'), zlib=z, chunksizes=(h, 2)).set(_bounds1d(mlats.data[:, 0]))
                root.create_variable('mlt_bounds', np.float64, ('mlts', 'vertex2'), zlib=z, chunksizes=(w, 2)).set(_bounds1d(mlts.data[0, :]))
            else:
                root.create_variable('mlat_bounds', np.float64, ('y', 'x', 'vertex4'), zlib=z, chunksizes=(1, w, 4)).set(_bounds2d(mlats.filled(np.nan)))
                root.create_variable('mlt_bounds', np.float64, ('y', 'x', 'vertex4'), zlib=z, chunksizes=(1, w, 4)).set(_bounds2d(mlts.filled(np.nan)))
        magPoleLat, magPoleLon = northGeomagneticPoleLocation(mapping.photoTime)
        mcrs = root.create_variable('mcrs', np.int8)        # holds no actual data
        mcrs.attrs['north_geomagnetic_pole_lat'] = np.float64(magPoleLat)
        mcrs.attrs['north_geomagnetic_pole_lon'] = np.float64(magPoleLon)
        mcrs.attrs['comment'] = 'Geocentric MLat/MLT system based on the given geomagnetic pole position'

    # DATA VARIABLES (reference netcdf.py:279-355)
    y = 'lats' if isLatLonPlateCarree else 'y'
    x = 'lons' if isLatLonPlateCarree else 'x'
    # netCDF has no unsigned types except byte: uint8 -> int16, uint16 -> int32
    imgDtypeMap = {np.dtype(np.uint8): np.int16, np.dtype(np.uint16): np.int32}
    if mapping.img.dtype not in imgDtypeMap:
        raise NotImplementedError('Image data type not supported: ' + str(mapping.img.dtype))
    imgDtype = imgDtypeMap[mapping.img.dtype]
    imgFillval = imgDtype(np.iinfo(imgDtype).min)
    img_ = mapping.img.astype(imgDtype).filled(imgFillval)
    if img_.shape[2] == 1:
        bands = ['img']
    elif img_.shape[2] == 3:
        bands = ['img_red', 'img_green', 'img_blue']
    else:
        raise NotImplementedError
    for i, band in enumerate(bands):
        img = root.create_variable(band, imgDtype, (y, x), fill_value=imgFillval, zlib=z, chunksizes=(1, w))
        img.attrs['units'] = 'unitless'
        img.attrs['valid_min'] = imgDtype(np.iinfo(mapping.img.dtype).min)
        img.attrs['valid_max'] = imgDtype(np.iinfo(mapping.img.dtype).max)
        img.attrs['actual_range'] = imgDtype([np.min(mapping.img[:, :, i]), np.max(mapping.img[:, :, i])])
        img.attrs['coordinates'] = 'altitude time' if isLatLonPlateCarree else 'lat lon altitude time'
        img.attrs['grid_mapping'] = 'crs'
        img.set(img_[:, :, i])

    # netCDF-CF knows no elevation angle but a zenith angle
    zena = 90 - mapping.elevation
    zenith_angle = root.create_variable('zenith_angle', np.float32, (y, x), zlib=z, chunksizes=(1, w))
    zenith_angle.attrs['units'] = 'degrees'
    zenith_angle.attrs['cell_methods'] = 'time: lat: lon: point' if isLatLonPlateCarree else 'time: y: x: point'
    zenith_angle.attrs['valid_min'] = np.float32(0)
    zenith_angle.attrs['valid_max'] = np.float32(90)
    zenith_angle.attrs['actual_range'] = [np.min(zena), np.max(zena)]
    zenith_angle.attrs['standard_name'] = 'zenith_angle'
    zenith_angle.attrs['coordinates'] = 'altitude time' if isLatLonPlateCarree else 'lat lon altitude time'
    zenith_angle.attrs['grid_mapping'] = 'crs'
    zenith_angle.attrs['long_name'] = 'Absolute sensor zenith angle'
    zenith_angle.set(zena.filled(np.nan))

    cameraPos = root.create_variable('camera_pos', np.float64, ('xyz',))
    cameraPos.attrs['units'] = 'kilometers'
    cameraPos.attrs['cell_methods'] = 'time: point'
    cameraPos.attrs['coordinates'] = 'time'
    cameraPos.attrs['long_name'] = 'Camera position in cartesian GCRS coordinates'
    cameraPos.attrs['comment'] = 'Axis order: xyz'
    cameraPos.set(mapping.cameraPosGCRS)

    crs = root.create_variable('crs', np.int8)              # holds no actual data
    crs.attrs['grid_mapping_name'] = 'latitude_longitude'   # = unknown projection lat/lon coordinate system
    crs.attrs['semi_major_axis'] = 6378137.0
    crs.attrs['inverse_flattening'] = 298.257223563
    crs.attrs['comment'] = 'Geographic Coordinate System, WGS 84'
    root.write(outputPath)


def _bounds1d(arr):
    assert arr.ndim == 1
    arr = arr[:, None]
    bounds = np.concatenate((arr[:-1], arr[1:]), axis=1)
    assert bounds.shape == (arr.shape[0] - 1, 2)
    return bounds


def _bounds2d(arr):
    assert arr.ndim == 2
    arr = arr[:, :, None]
    bounds = np.concatenate((arr[0:-1, 0:-1], arr[0:-1, 1:], arr[1:, 1:], arr[1:, 0:-1]), axis=2)
    assert bounds.shape == (arr.shape[0] - 1, arr.shape[1] - 1, 4)
    return bounds


def _unix(dt):
    return (dt - datetime(1970, 1, 1)).total_seconds()
