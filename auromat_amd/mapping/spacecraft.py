"""
Spacecraft (ISS) camera mappings from WCS headers and images (reference auromat/mapping/spacecraft.py): arrays and
header dicts, or image files and astrometry.net ``.wcs`` files (:mod:`auromat_amd.fits`, :mod:`auromat_amd.util.image`),
one at a time, as sequences, or through the reference's folder providers.  The camera position comes from the header
cards or the caller; the reference's TLE / ephemeris look-up is not part of this package.
"""
from __future__ import division

import os
from datetime import datetime, timedelta

import numpy as np

from ..coordinates.geodesic import wgs84A, wgs84B
from ..coordinates.intersection import ellipsoidLineIntersects
from .astrometry import BaseAstrometryMapping
from .mapping import BaseMappingProvider


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
    from ..coordinates.wcs import is_plain_tan, zenithal_directions_device
    if not is_plain_tan(wcsHeader):
        # another zenithal projection or SIP terms (the reference hands such headers to astropy.wcs, wcs.py:54-56): the
        # corner directions from the device generator (amt_directions_zenithal; coordinates.wcs.zenithal_pix2world is its
        # NumPy restatement), everything downstream through the directions-in kernel; centres are the mean of their four
        # corner hits (the fast mode), whatever fastCenterCalculation says
        from .astrometry import DirectionArrayMapping
        img = np.asarray(imageArray)
        dirs = zenithal_directions_device(wcsHeader, img.shape[1], img.shape[0], corner=True)
        m = DirectionArrayMapping(dirs, altitude, img, cam, photoTime, identifier, metadata)
        m.originalPhotoTime = originalPhotoTime
        return m
    return ArraySpacecraftMapping(wcsHeader, altitude, imageArray, cam, photoTime, identifier, metadata,
                                  originalPhotoTime=originalPhotoTime,
                                  fastCenterCalculation=fastCenterCalculation)


def getMappingSequence(imagePathsOrArrays, wcsPaths, metadatas=None, timeshift=None, noradId=None, tleFolder=None,
                       spacetrack=None, altitude=110, parallel=False, fastCenterCalculation=False):
    """
    Generator of mappings for corresponding images (paths or arrays) and headers (``.wcs`` paths or dicts), in the
    given order — the reference's signature and positional order (spacecraft.py:308-332).  ``parallel`` (a process
    pool per frame in the reference) is accepted and ignored: frames are independent and
    :mod:`auromat_amd.sequence` / :class:`auromat_amd.pipeline.SequencePipeline` shard and pipeline them on the GPUs.
    """
    if not metadatas:
        metadatas = [None] * len(wcsPaths)
    for img, hdr, meta in zip(imagePathsOrArrays, wcsPaths, metadatas):
        yield getMapping(img, hdr, timeshift, noradId, tleFolder, spacetrack, altitude=altitude,
                         fastCenterCalculation=fastCenterCalculation, metadata=meta)


def getShiftedPhotoTime(header):
    """The corrected photo time or, if not available, the original one (reference fits.py:381-391)."""
    _, d, _ = getShiftedSpacecraftPosition(header)
    return d if d is not None else getPhotoTime(header)


def _parse_dates(dic):
    # metadata.json stores datetimes as ISO strings (reference spacecraft.py: _parseDates)
    for k, v in list(dic.items()):
        if isinstance(v, str):
            for fmt in ('%Y-%m-%dT%H:%M:%S.%f', '%Y-%m-%dT%H:%M:%S'):
                try:
                    dic[k] = datetime.strptime(v, fmt)
                    break
                except ValueError:
                    pass
    return dic


def _load_metadata(path):
    import json
    if path and os.path.exists(path):
        with open(path) as fp:
            return json.load(fp, object_hook=_parse_dates)
    return None


def _metadata_of(metadata, identifier):
    if not metadata:
        return None
    return dict(list(metadata['sequence_metadata'].items()) + list(metadata['image_metadata'][identifier].items()))


class SpacecraftMappingProvider(BaseMappingProvider):
    """
    Mappings of a folder (or of two path lists) of image files and their astrometry.net ``.wcs`` solutions, by date,
    identifier or as a sequence — the reference's provider (spacecraft.py:40-248) on
    :func:`auromat_amd.fits.readHeader` / :func:`getMapping`.  A ``metadata.json`` next to the images
    (``sequence_metadata`` + per-image ``image_metadata``) is attached to the mappings.
    """

    def __init__(self, imageSequenceFolder, wcsFolder=None, imageFileExtension=None, timeshift=None, noradId=None,
                 tleFolder=None, spacetrack=None, altitude=110, maxTimeOffset=3, sequenceInParallel=False,
                 fastCenterCalculation=False):
        """
        :param imageSequenceFolder: folder path or a list of image file paths
        :param wcsFolder: folder path or a list of wcs file paths; optional if imageSequenceFolder is a folder that
                          contains the wcs files as well
        """
        BaseMappingProvider.__init__(self, maxTimeOffset=maxTimeOffset)
        if wcsFolder is None:
            assert not isinstance(imageSequenceFolder, list), \
                'The wcsFolder parameter is required if imageSequenceFolder is a list'
            wcsFolder = imageSequenceFolder
        if isinstance(imageSequenceFolder, list) and isinstance(wcsFolder, list):
            self.imageSequenceFolder = self.wcsFolder = None
            self.imagePaths, self.wcsPaths = list(imageSequenceFolder), list(wcsFolder)
            self._imageFileExtension = os.path.splitext(self.imagePaths[0])[1][1:]
            self._match()
        elif not isinstance(imageSequenceFolder, list) and not isinstance(wcsFolder, list):
            self.imageSequenceFolder, self.wcsFolder = imageSequenceFolder, wcsFolder
            self._imageFileExtension = imageFileExtension
            self.reload()
        else:
            raise ValueError('imageSequenceFolder and wcsFolder must be both path lists or folder paths')
        self.timeshift, self.noradId, self.tleFolder, self.spacetrack = timeshift, noradId, tleFolder, spacetrack
        self.altitude, self.fastCenterCalculation = altitude, fastCenterCalculation
        self.metadata = _load_metadata(os.path.join(os.path.dirname(self.imagePaths[0]), 'metadata.json')) \
            if self.imagePaths else None
        self._sequenceInParallel = sequenceInParallel

    def __len__(self):
        return len(self.wcsPaths)

    def reload(self):
        """Refresh to the current state of the folders (no-op for path lists)."""
        if self.wcsFolder is None:
            return
        self.wcsPaths = sorted(os.path.join(self.wcsFolder, f) for f in os.listdir(self.wcsFolder) if f.endswith('.wcs'))
        try:
            ext = '.' + self.imageFileExtension
            self.imagePaths = sorted(os.path.join(self.imageSequenceFolder, f)
                                     for f in os.listdir(self.imageSequenceFolder) if f.endswith(ext))
        except ValueError:
            self.imagePaths, self.wcsPaths = [], []
        self._match()

    def _match(self):
        """Every solution needs its image; solutions sorted by (shifted) photo time."""
        images = {os.path.splitext(os.path.basename(p))[0]: p for p in self.imagePaths}
        ids = [os.path.splitext(os.path.basename(p))[0] for p in self.wcsPaths]
        missing = [i for i in ids if i not in images]
        assert not missing, 'no image for the solutions ' + str(missing)
        from ..fits import readHeader
        dated = sorted((getShiftedPhotoTime(readHeader(p)), p, i) for p, i in zip(self.wcsPaths, ids))
        self.dates = [d for d, _, _ in dated]
        self.wcsPaths = [p for _, p, _ in dated]
        self.ids = [i for _, _, i in dated]
        self._imageOf = images

    @property
    def imageFileExtension(self):
        """e.g. 'jpg'; found from the files when not given."""
        if self._imageFileExtension is None:
            names = os.listdir(self.imageSequenceFolder)
            solved = [f for f in os.listdir(self.wcsFolder) if f.endswith('.wcs')]
            if self.imageSequenceFolder == self.wcsFolder:
                names = [f for f in names if f not in solved]
            for wcs in solved:
                base = os.path.splitext(wcs)[0]
                matches = [f for f in names if os.path.splitext(f)[0] == base]
                if len(matches) == 1:
                    self._imageFileExtension = os.path.splitext(matches[0])[1][1:]
                    break
                elif len(matches) > 1:
                    raise ValueError('Image file extension not given but multiple candidates exist: ' + str(matches))
            if self._imageFileExtension is None:
                raise ValueError('Image file extension could not be determined. Make sure that there exists at least '
                                 'one .wcs file and a corresponding image with the same filename base.')
        return self._imageFileExtension

    @property
    def range(self):
        return self.dates[0], self.dates[-1]

    @property
    def unsolvedIds(self):
        solved = set(self.ids)
        return sorted(i for i in self._imageOf if i not in solved)

    def _nearest(self, date):
        from ..utils import findNearest
        idx = findNearest(self.dates, date)
        return idx, abs(self.dates[idx] - date).total_seconds()

    def contains(self, date):
        return bool(self.dates) and self._nearest(date)[1] <= self.maxTimeOffset

    def get(self, date):
        if not self.dates:
            raise ValueError('No image found')
        idx, offset = self._nearest(date)
        if offset > self.maxTimeOffset:
            raise ValueError('No image found')
        identifier = self.ids[idx]
        return getMapping(self._imageOf[identifier], self.wcsPaths[idx], self.timeshift, self.noradId, self.tleFolder,
                          self.spacetrack, altitude=self.altitude, fastCenterCalculation=self.fastCenterCalculation,
                          metadata=_metadata_of(self.metadata, identifier))

    def getById(self, identifier):
        matched = [i for i in self.ids if identifier in i]
        if len(matched) != 1:
            raise ValueError('Ambiguous or unknown identifier: ' + str(matched))
        return self.get(self.dates[self.ids.index(matched[0])])

    def getSequence(self, dateBegin=None, dateEnd=None):
        assert dateBegin is None and dateEnd is None, 'Date ranges not supported'
        metadatas = [_metadata_of(self.metadata, i) for i in self.ids] if self.metadata else None
        return getMappingSequence([self._imageOf[i] for i in self.ids], self.wcsPaths, metadatas, self.timeshift,
                                  self.noradId, self.tleFolder, self.spacetrack, altitude=self.altitude,
                                  parallel=self._sequenceInParallel, fastCenterCalculation=self.fastCenterCalculation)


class SpacecraftMappingPathProvider(BaseMappingProvider):
    """The same for explicit path lists, sequence access only (reference spacecraft.py:250-300)."""

    def __init__(self, imagePaths, wcsPaths, metadataPath=None, timeshift=None, noradId=None, tleFolder=None,
                 spacetrack=None, altitude=110, maxTimeOffset=3, sequenceInParallel=False, fastCenterCalculation=False):
        BaseMappingProvider.__init__(self, maxTimeOffset=maxTimeOffset)
        assert len(imagePaths) == len(wcsPaths)
        from ..fits import readHeader
        pairs = sorted(zip(wcsPaths, imagePaths), key=lambda wi: getPhotoTime(readHeader(wi[0])))
        self.wcsPaths = [w for w, _ in pairs]
        self.imagePaths = [i for _, i in pairs]
        self.timeshift, self.noradId, self.tleFolder, self.spacetrack = timeshift, noradId, tleFolder, spacetrack
        self.altitude, self.sequenceInParallel, self.fastCenterCalculation = altitude, sequenceInParallel, fastCenterCalculation
        self.metadata = _load_metadata(metadataPath)

    def __len__(self):
        return len(self.wcsPaths)

    @property
    def imageFileExtension(self):
        return os.path.splitext(self.imagePaths[0])[1][1:]

    @property
    def range(self):
        from ..fits import readHeader
        return tuple(getShiftedPhotoTime(readHeader(p)) if self.timeshift is None else
                     getPhotoTime(readHeader(p)) + self.timeshift for p in (self.wcsPaths[0], self.wcsPaths[-1]))

    def contains(self, date):
        raise NotImplementedError

    def get(self, date):
        raise NotImplementedError

    def getById(self, identifier):
        raise NotImplementedError

    def getSequence(self, dateBegin=None, dateEnd=None):
        assert dateBegin is None and dateEnd is None, 'Date ranges not supported'
        metadatas = None
        if self.metadata:
            metadatas = [_metadata_of(self.metadata, os.path.splitext(os.path.basename(p))[0]) for p in self.imagePaths]
        return getMappingSequence(self.imagePaths, self.wcsPaths, metadatas, self.timeshift, self.noradId, self.tleFolder,
                                  self.spacetrack, altitude=self.altitude, parallel=self.sequenceInParallel,
                                  fastCenterCalculation=self.fastCenterCalculation)
