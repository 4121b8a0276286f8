"""
Coordinate conversions, ray/ellipsoid intersections and the TAN camera model, executed by the HIP
kernels of libauromat_hip.so.  Mirrors the public surface of the reference's
``auromat.coordinates`` package for the georeferencing path (intersection, transform, wcs, igrf,
and the WGS84 constants of geodesic); like the reference package it does not depend on mapping objects.
"""
