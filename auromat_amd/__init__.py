"""
auromat_amd — MI355X-native georeferencing + resampling of aurora imagery.

Drop-in for the per-pixel hot path of esa/auromat (camera/WCS ray cast, inflated-WGS84
intersection, geodetic and MLat/MLT conversion, elevation, mask rules, histogram-binned
regridding): same Python names and semantics as the reference, executed by hand-written HIP kernels
(gfx950) through the C ABI of include/auromat_hip.h.  There is no CPU fallback.

    from auromat_amd.mapping.spacecraft import getMapping          # reference: auromat.mapping.spacecraft
    from auromat_amd.resample import resample, resampleMLatMLT     # reference: auromat.resample
    m = getMapping(img, wcsHeader, altitude=110, fastCenterCalculation=True).maskedByElevation(10)
    r = resample(m, pxPerDeg=10)

Submodules import lazily so that ``import auromat_amd`` works on a machine that only builds.
"""
__version__ = '0.1.0'

__all__ = ['coordinates', 'mapping', 'resample', 'util', 'pipeline', 'sequence', 'synthetic']
