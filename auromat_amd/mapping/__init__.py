"""
Georeferenced-image ("mapping") classes backed by device-resident arrays.  Mirrors the parts of
the reference's ``auromat.mapping`` package that sit on the georeferencing + resampling path:
``mapping`` (BaseMapping, GenericMapping, BoundingBox, ...), ``astrometry`` (WCS camera mappings)
``spacecraft`` (ArraySpacecraftMapping, getMapping), ``miracle`` (all-sky fisheye mappings) and ``themis``
(altitude reprojection, ThemisMapping).
"""
