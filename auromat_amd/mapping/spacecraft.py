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
    """What metadata.json says about one image: the sequence's entries, overridden by the image's own."""
    if not metadata:
        return None
    merged = dict(metadata['sequence_metadata'])
    merged.update(metadata['image_metadata'][identifier])
    return merged


def _stem(path):
    return os.path.splitext(os.path.basename(path))[0]


def _solved_frames(image_paths, wcs_paths, time_of):
    """DateCatalogue of the frames that have a solution: per ``.wcs`` file its image (same file name without the extension),
    ordered by `time_of(header)`; payload (identifier, image path, wcs path).  A solution without an image is an error."""
    from ..fits import readHeader
    from ._catalogue import DateCatalogue
    image_of = {_stem(p): p for p in image_paths}
    orphans = [_stem(w) for w in wcs_paths if _stem(w) not in image_of]
    if orphans:
        raise ValueError('no image file for the solutions ' + ', '.join(orphans))
    return DateCatalogue(((time_of(readHeader(w)), (_stem(w), image_of[_stem(w)], w), w) for w in wcs_paths), 'the set of .wcs files'), image_of


class SpacecraftMappingProvider(BaseMappingProvider):
    """
    The frames of a folder — image files and, next to them or in a folder of their own, their astrometry.net ``.wcs`` solutions
    — or of two explicit path lists, as mappings by date, by identifier or as a sequence in time order (the reference's provider,
    spacecraft.py:40-248, on :func:`auromat_amd.fits.readHeader` / :func:`getMapping`).  A ``metadata.json`` beside the images
    (``sequence_metadata`` for all, ``image_metadata`` per identifier) is merged into each mapping's metadata.  The camera
    position comes from the header cards (TLE look-up is not part of this package; the parameters are accepted and passed on).
    """

    def __init__(self, imageSequenceFolder, wcsFolder=None, imageFileExtension=None, timeshift=None, noradId=None,
                 tleFolder=None, spacetrack=None, altitude=110, maxTimeOffset=3, sequenceInParallel=False,
                 fastCenterCalculation=False):
        """
        :param imageSequenceFolder: folder path or a list of image file paths
        :param wcsFolder: folder path or a list of wcs file paths; may be left out when the image folder holds the wcs files too
        """
        BaseMappingProvider.__init__(self, maxTimeOffset=maxTimeOffset)
        lists = isinstance(imageSequenceFolder, list), isinstance(wcsFolder, list)
        if wcsFolder is None:
            if lists[0]:
                raise ValueError('wcsFolder is needed when imageSequenceFolder is a list of paths')
            wcsFolder, lists = imageSequenceFolder, (False, False)
        if lists[0] != lists[1]:
            raise ValueError('imageSequenceFolder and wcsFolder must be both path lists or folder paths')
        self._from_lists = lists[0]
        self.imageSequenceFolder, self.wcsFolder = (None, None) if self._from_lists else (imageSequenceFolder, wcsFolder)
        self._given = (list(imageSequenceFolder), list(wcsFolder)) if self._from_lists else None
        self._imageFileExtension = os.path.splitext(self._given[0][0])[1][1:] if self._from_lists else imageFileExtension
        self.timeshift, self.noradId, self.tleFolder, self.spacetrack = timeshift, noradId, tleFolder, spacetrack
        self.altitude, self.fastCenterCalculation = altitude, fastCenterCalculation
        self._sequenceInParallel = sequenceInParallel
        self.reload()
        self.metadata = _load_metadata(os.path.join(os.path.dirname(self.imagePaths[0]), 'metadata.json')) if self.imagePaths else None

    def reload(self):
        """Look at the folders again (path lists: at the lists as given)."""
        if self._from_lists:
            images, solutions = self._given
        else:
            solutions = sorted(os.path.join(self.wcsFolder, f) for f in os.listdir(self.wcsFolder) if f.endswith('.wcs'))
            try:
                suffix = '.' + self.imageFileExtension
                images = sorted(os.path.join(self.imageSequenceFolder, f) for f in os.listdir(self.imageSequenceFolder) if f.endswith(suffix))
            except ValueError:
                images, solutions = [], []            # nothing solved yet: no extension to go by
        self._frames, self._imageOf = _solved_frames(images, solutions, getShiftedPhotoTime)
        self.imagePaths = images
        self.dates = list(self._frames.dates)
        self.ids = [f[0] for f in self._frames.payloads]
        self.wcsPaths = [f[2] for f in self._frames.payloads]

    def __len__(self):
        return len(self._frames)

    @property
    def imageFileExtension(self):
        """e.g. 'jpg'.  When not given: the extension of the one file that shares its name with a ``.wcs`` file."""
        if self._imageFileExtension is None:
            solved = set(_stem(f) for f in os.listdir(self.wcsFolder) if f.endswith('.wcs'))
            partners = {}
            for f in os.listdir(self.imageSequenceFolder):
                base, ext = os.path.splitext(f)
                if base in solved and ext != '.wcs':
                    partners.setdefault(base, []).append(ext[1:])
            for base in sorted(partners):
                if len(partners[base]) > 1:
                    raise ValueError('no image file extension given and %s.* is ambiguous: %s' % (base, sorted(partners[base])))
                self._imageFileExtension = partners[base][0]
                break
            else:
                raise ValueError('no image file extension given and none can be found: there is no .wcs file with an image file '
                                 'of the same name')
        return self._imageFileExtension

    @property
    def range(self):
        return self._frames.span

    @property
    def unsolvedIds(self):
        """Identifiers of the images that have no ``.wcs`` file (yet)."""
        return sorted(set(self._imageOf) - set(self.ids))

    def contains(self, date):
        return self._frames.within(date, self.maxTimeOffset)

    def _mapping(self, frame):
        identifier, image, solution = frame
        return getMapping(image, solution, self.timeshift, self.noradId, self.tleFolder, self.spacetrack, altitude=self.altitude,
                          fastCenterCalculation=self.fastCenterCalculation, metadata=_metadata_of(self.metadata, identifier))

    def get(self, date):
        if not len(self._frames):
            raise ValueError('the provider holds no solved frame')
        return self._mapping(self._frames.pick(date, self.maxTimeOffset))

    def getById(self, identifier):
        """The frame whose identifier contains `identifier`; ValueError unless exactly one does."""
        hits = [f for f in self._frames.payloads if identifier in f[0]]
        if len(hits) != 1:
            raise ValueError('%r names %d frames: %s' % (identifier, len(hits), [f[0] for f in hits][:5]))
        return self._mapping(hits[0])

    def getSequence(self, dateBegin=None, dateEnd=None):
        if dateBegin is not None or dateEnd is not None:
            raise NotImplementedError('sequences of a date range (the reference does not support them either, spacecraft.py:226-228)')
        frames = self._frames.payloads
        metadatas = [_metadata_of(self.metadata, f[0]) for f in frames] if self.metadata else None
        return getMappingSequence([f[1] for f in frames], [f[2] for f in frames], metadatas, self.timeshift, self.noradId,
                                  self.tleFolder, self.spacetrack, altitude=self.altitude, parallel=self._sequenceInParallel,
                                  fastCenterCalculation=self.fastCenterCalculation)


class SpacecraftMappingPathProvider(BaseMappingProvider):
    """Explicit lists of image and ``.wcs`` paths (pairs by position), as a sequence in the order of the ORIGINAL photo times
    only; look-up by date or identifier is not offered (reference spacecraft.py:250-300)."""

    def __init__(self, imagePaths, wcsPaths, metadataPath=None, timeshift=None, noradId=None, tleFolder=None,
                 spacetrack=None, altitude=110, maxTimeOffset=3, sequenceInParallel=False, fastCenterCalculation=False):
        BaseMappingProvider.__init__(self, maxTimeOffset=maxTimeOffset)
        if len(imagePaths) != len(wcsPaths):
            raise ValueError('%d images for %d solutions' % (len(imagePaths), len(wcsPaths)))
        from ..fits import readHeader
        order = sorted(range(len(wcsPaths)), key=lambda k: getPhotoTime(readHeader(wcsPaths[k])))
        self.wcsPaths = [wcsPaths[k] for k in order]
        self.imagePaths = [imagePaths[k] for k in order]
        self.timeshift, self.noradId, self.tleFolder, self.spacetrack = timeshift, noradId, tleFolder, spacetrack
        self.altitude, self.sequenceInParallel, self.fastCenterCalculation = altitude, sequenceInParallel, fastCenterCalculation
        self.metadata = _load_metadata(metadataPath)

    def __len__(self):
        return len(self.wcsPaths)

    @property
    def imageFileExtension(self):
        return os.path.splitext(self.imagePaths[0])[1][1:]

    def _time_of(self, wcs_path):
        from ..fits import readHeader
        header = readHeader(wcs_path)
        return getShiftedPhotoTime(header) if self.timeshift is None else getPhotoTime(header) + self.timeshift

    @property
    def range(self):
        return self._time_of(self.wcsPaths[0]), self._time_of(self.wcsPaths[-1])

    def contains(self, date):
        raise NotImplementedError

    def get(self, date):
        raise NotImplementedError

    def getById(self, identifier):
        raise NotImplementedError

    def getSequence(self, dateBegin=None, dateEnd=None):
        if dateBegin is not None or dateEnd is not None:
            raise NotImplementedError('sequences of a date range')
        metadatas = [_metadata_of(self.metadata, _stem(p)) for p in self.imagePaths] if self.metadata else None
        return getMappingSequence(self.imagePaths, self.wcsPaths, metadatas, self.timeshift, self.noradId, self.tleFolder,
                                  self.spacetrack, altitude=self.altitude, parallel=self.sequenceInParallel,
                                  fastCenterCalculation=self.fastCenterCalculation)
