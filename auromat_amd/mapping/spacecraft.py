"""
Spacecraft (ISS) camera mappings from WCS headers and image arrays
(reference auromat/mapping/spacecraft.py).  File / TLE / network plumbing of the reference module
(FITS reading, ephemeris look-up, image decoding) is outside the accelerated path: headers are
dicts, images are arrays, and the camera position comes from the header cards or the caller.
"""
from __future__ import division

from datetime import datetime, timedelta

import numpy as np

from ..coordinates.geodesic import wgs84A, wgs84B
from ..coordinates.intersection import ellipsoidLineIntersects
from .astrometry import BaseAstrometryMapping


def getPhotoTime(header):
    """DATE-OBS card as datetime, or None (reference fits.py:365-379)."""
    dateobs = header.get('DATE-OBS')
    if dateobs is None:
        return None
    try:
        return datetime.strptime(dateobs, '%Y-%m-%dT%H:%M:%S.%f')
    except ValueError:
        return datetime.strptime(dateobs, '%Y-%m-%dT%H:%M:%S')


def getSpacecraftPosition(header):
    """(xyz, date) from POSX/Y/Z + DATE-OBS, or (None, None) (reference fits.py:393-405)."""
    date = getPhotoTime(header)
    if header.get('POSX') is None or date is None:
        return None, None
    return np.array([header['POSX'], header['POSY'], header['POSZ']], dtype=np.float64), date


def getShiftedSpacecraftPosition(header):
    """(xyz, shifted date, delta) from POS?SHIF + DATESHIF, or (None, None, None) (reference fits.py:427-442)."""
    date = getPhotoTime(header)
    shift = header.get('DATESHIF')
    if header.get('POSXSHIF') is None or date is None or shift is None:
        return None, None, None
    delta = timedelta(seconds=shift)
    xyz = np.array([header['POSXSHIF'], header['POSYSHIF'], header['POSZSHIF']], dtype=np.float64)
    return xyz, date + delta, delta


def frame_inputs(header):
    """(cameraPosGCRS, photoTime) of a frame from its header cards: the shifted ones (POS?SHIF, DATE-OBS + DATESHIF)
    when present, else POS? and DATE-OBS — the choice ``getMapping`` makes (reference spacecraft.py:437-452)."""
    cam, t, _ = getShiftedSpacecraftPosition(header)
    if cam is None:
        cam, t = getSpacecraftPosition(header)
    if cam is None:
        raise ValueError('Spacecraft position is missing in the header (POSX/POSY/POSZ cards)')
    return cam, t


class BaseSpacecraftMapping(BaseAstrometryMapping):
    """
    A camera in/on a spacecraft looking both on earth and the stars; the stars gave the WCS
    solution from which each pixel's direction follows (reference spacecraft.py:487-555).
    """

    def __init__(self, wcsHeader, alti, cameraPosGCRS, photoTime, identifier, metadata=None,
                 originalPhotoTime=None, fastCenterCalculation=False):
        BaseAstrometryMapping.__init__(self, wcsHeader, alti, cameraPosGCRS, photoTime, identifier, metadata,
                                       fastCenterCalculation=fastCenterCalculation)
        self._originalPhotoTime = photoTime if originalPhotoTime is None else originalPhotoTime

    @property
    def originalPhotoTime(self):
        return self._originalPhotoTime

    @property
    def intersectsEarth(self):
        """Boolean array (h,w): whether a pixel center's ray intersects the (un-inflated) earth."""
        def make():
            d = self.cameraToPixelCenterDirection
            hit = ellipsoidLineIntersects(wgs84A, wgs84B, self.cameraPosGCRS, d.reshape(-1, 3))
            return hit.reshape(d.shape[0], d.shape[1])
        return self._cached('intersects_earth', make)

    def isConsistent(self, starPxCoords=None):
        """
        Plausibility check of timestamp + astrometric solution (reference spacecraft.py:523-555):
        neither every nor no pixel may hit the earth, and no star used for solving may be covered by it.
        """
        hits = self.intersectsEarth
        if np.all(hits) or not np.any(hits):
            return False
        if starPxCoords is not None and np.any(hits[starPxCoords[:, 1], starPxCoords[:, 0]]):
            return False
        return True


class ArraySpacecraftMapping(BaseSpacecraftMapping):
    """
    Spacecraft mapping over an RGB image array (reference spacecraft.py:583-595).
    The image carries the centre mask (class invariant, mapping.py:299-316).
    """

    def __init__(self, wcsHeader, alti, img, cameraPosGCRS, photoTime, identifier, metadata=None,
                 originalPhotoTime=None, fastCenterCalculation=False):
        img = np.asarray(img)
        assert img.ndim == 3
        assert img.dtype in [np.uint8, np.uint16]
        assert img.shape[:2] == (wcsHeader['IMAGEH'], wcsHeader['IMAGEW'])
        BaseSpacecraftMapping.__init__(self, wcsHeader, alti, cameraPosGCRS, photoTime, identifier, metadata,
                                       originalPhotoTime=originalPhotoTime,
                                       fastCenterCalculation=fastCenterCalculation)
        self._img_array = img


# name used by BASELINE.json's north star
ArrayMapping = ArraySpacecraftMapping


def getMapping(imagePathOrArray, wcsPathOrHeader, timeshift=None, noradId=None, tleFolder=None, spacetrack=None,
               altitude=110, fastCenterCalculation=False, metadata=None, nosanitize=False, identifier=None,
               cameraPosGCRS=None):
    """
    Build a mapping from an image (array, or path of an image file) and a WCS header (dict, or path of a ``.wcs``
    header-only FITS file) — the reference's signature and positional order (spacecraft.py:380-426, 428-485).  Photo
    time and camera position are taken from the header: the shifted cards (DATESHIF, POS?SHIF) if present, else
    DATE-OBS and POS?.  Files are read with :func:`auromat_amd.util.image.loadImage` (Pillow; not for RAW files) and
    :func:`auromat_amd.fits.readHeader`.

    Not supported (the reference's network plumbing, out of this package's scope) and rejected with a clear error
    instead of being ignored: ``noradId`` / ``tleFolder`` / ``spacetrack`` (camera position from two-line elements:
    pass ``cameraPosGCRS`` instead, which is also needed with ``timeshift`` or for a header without POS? cards).
    ``nosanitize`` only affects the reference's file-based mappings and is accepted.
    """
    imageArray, wcsHeader = imagePathOrArray, wcsPathOrHeader
    if isinstance(wcsHeader, str):
        from ..fits import readHeader
        wcsHeader = readHeader(wcsHeader)
    if isinstance(imageArray, str):
        from ..util.image import loadImage
        if identifier is None:
            import os
            identifier = os.path.splitext(os.path.basename(imageArray))[0]
        imageArray = loadImage(imageArray)
    if noradId is not None or tleFolder is not None or spacetrack is not None:
        raise NotImplementedError('noradId / tleFolder / spacetrack (camera position from two-line elements via '
                                  'pyephem) are not part of auromat_amd: pass cameraPosGCRS=[x, y, z] (km, GCRS)')
    originalPhotoTime = getPhotoTime(wcsHeader)
    if originalPhotoTime is None:
        raise ValueError('DATE-OBS missing in FITS header')
    if timeshift is not None:
        photoTime, cam = originalPhotoTime + timeshift, None
    else:
        cam, photoTime, _ = getShiftedSpacecraftPosition(wcsHeader)
        if cam is None:
            photoTime = originalPhotoTime
            cam, _ = getSpacecraftPosition(wcsHeader)
    if cameraPosGCRS is not None:
        cam = np.asarray(cameraPosGCRS, dtype=np.float64)
    if cam is None:
        raise ValueError('Spacecraft position is missing in the header; pass cameraPosGCRS '
                         '(TLE propagation is not part of this package)')
    return ArraySpacecraftMapping(wcsHeader, altitude, imageArray, cam, photoTime, identifier, metadata,
                                  originalPhotoTime=originalPhotoTime,
                                  fastCenterCalculation=fastCenterCalculation)


def getMappingSequence(imageArrays, wcsHeaders, metadatas=None, timeshift=None, altitude=110,
                       fastCenterCalculation=False):
    """
    Generator of mappings for corresponding images and headers (reference spacecraft.py:308-332).
    Frames are independent; :mod:`auromat_amd.sequence` shards them across GPUs.
    """
    if metadatas is None:
        metadatas = [None] * len(wcsHeaders)
    for img, hdr, meta in zip(imageArrays, wcsHeaders, metadatas):
        yield getMapping(img, hdr, timeshift=timeshift, altitude=altitude,
                         fastCenterCalculation=fastCenterCalculation, metadata=meta)
