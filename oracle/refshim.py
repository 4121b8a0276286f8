"""
TEST INFRASTRUCTURE — container-only loader for the *real* reference.

Imports esa/auromat from /root/reference (read-only, never copied) on top of a
few in-memory stand-ins for third-party packages that are absent from this
image (astropy, geographiclib, scikit-image, numpy.core.umath_tests, ...).
Only ``oracle/make_golden.py`` uses it, and only in the build container: /root/reference does not exist on the GPU
box, so nothing under tests/, bench.py or the product may import this file.

What the stand-ins replace and why the arithmetic is unaffected:

* ``numpy.core.umath_tests.matrix_multiply`` -> ``np.matmul`` and ``inner1d`` ->
  ``einsum('...i,...i')`` (same gufuncs, renamed in NumPy >= 1.10).
* ``astropy.time.Time(dt, scale='utc').jd`` -> two-part JD collapsed to one
  double (2451545 + days since 2000-01-01T12:00).  Used once per frame
  (reference transform.py:525-530).  The resulting ``et`` is *recorded in every
  fixture*, so parity never depends on this stand-in.
* ``astropy.coordinates.Angle(x*u.deg).wrap_at(w*u.deg).degree`` -> modular wrap
  into [w-360, w) (reference resample.py:213,218,276-277).
* ``astropy.units`` -> minimal deg/arcsec/rad objects.
* ``geographiclib.constants.Constants`` -> the two WGS84 numbers
  (reference geodesic.py:20-21).
* ``skimage.measure``, ``auromat.util.image``, ``auromat.fits``,
  ``auromat.coordinates.ephem`` -> empty (never reached on the hot path).
* ``auromat/util/histogram.py:262`` ``hist[core]`` -> ``hist[tuple(core)]`` is
  patched *in memory* at import (list-of-slices indexing was removed in NumPy
  1.23); nothing is written anywhere.
"""
import importlib.abc
import importlib.util
import sys
import types
from datetime import datetime

import numpy as np

REFERENCE_ROOT = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Unit(object):
    __array_ufunc__ = None  # let ndarray * unit reach __rmul__

    def __init__(self, name, in_deg):
        self.name = name
        self.in_deg = in_deg

    def __rmul__(self, value):
        return _Quantity(value, self)

    __mul__ = __rmul__


class _Quantity(object):
    def __init__(self, value, unit):
        self._value = value
        self.unit = unit

    def to(self, unit):
        if unit is self.unit:
            return _Quantity(self._value, unit)
        return _Quantity(np.asarray(self._value) * (self.unit.in_deg / unit.in_deg) if np.ndim(self._value)
                         else self._value * (self.unit.in_deg / unit.in_deg), unit)

    @property
    def value(self):
        return self._value


_deg = _Unit('deg', 1.0)
_arcsec = _Unit('arcsec', 1.0 / 3600.0)
_rad = _Unit('rad', 180.0 / np.pi)


class _Angle(object):
    def __init__(self, q):
        self._deg = np.array(q.to(_deg).value, dtype=np.float64, copy=True)

    def wrap_at(self, wrap):
        w = float(wrap.to(_deg).value)
        a = self._deg
        # astropy Angle._wrap_at: same sequence of operations
        wraps = (a - (w - 360.0)) // 360.0
        a = a - wraps * 360.0
        a = np.where(a >= w, a - 360.0, a)
        a = np.where(a < w - 360.0, a + 360.0, a)
        out = _Angle.__new__(_Angle)
        out._deg = a
        return out

    @property
    def degree(self):
        return self._deg if self._deg.ndim else float(self._deg)


class _Time(object):
    def __init__(self, date, scale='utc'):
        assert scale == 'utc'
        self._date = date

    @property
    def jd(self):
        delta = self._date - datetime(2000, 1, 1, 12)
        return 2451545.0 + (delta.days + (delta.seconds + delta.microseconds / 1e6) / 86400.0)


def install_shims():
    if getattr(install_shims, 'done', False):
        return
    if not hasattr(np, 'int'):
        np.int = int
    if not hasattr(np, 'float'):
        np.float = float
    if not hasattr(np, 'bool'):
        np.bool = bool

    _mod('numpy.core.umath_tests', matrix_multiply=np.matmul,
         inner1d=lambda a, b: np.einsum('...i,...i->...', a, b))

    units = _mod('astropy.units', deg=_deg, degree=_deg, arcsec=_arcsec, rad=_rad)
    angles = _mod('astropy.coordinates.angles', Angle=_Angle)
    coords = _mod('astropy.coordinates', Angle=_Angle, angles=angles)
    time_ = _mod('astropy.time', Time=_Time)
    const = _mod('astropy.constants')
    wcswcs = _mod('astropy.wcs.wcs', WCS=None)
    wcs = _mod('astropy.wcs', wcs=wcswcs, WCS=None)
    fits = _mod('astropy.io.fits')
    io = _mod('astropy.io', fits=fits)
    _mod('astropy', __version__='0.4.2', units=units, coordinates=coords, time=time_,
         constants=const, wcs=wcs, io=io)

    class Constants(object):
        WGS84_a = 6378137.0
        WGS84_f = 1 / 298.257223563
    gconst = _mod('geographiclib.constants', Constants=Constants)
    ggeod = _mod('geographiclib.geodesic', Geodesic=None)
    _mod('geographiclib', constants=gconst, geodesic=ggeod)

    sk_measure = _mod('skimage.measure')
    _mod('skimage', measure=sk_measure)

    import numpy.testing
    sys.modules.setdefault('numpy.testing.utils', numpy.testing)

    attrib = _mod('nose.plugins.attrib', attr=lambda *a, **k: (lambda f: f))
    plugins = _mod('nose.plugins', attrib=attrib)
    _mod('nose', plugins=plugins)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import auromat  # noqa: F401  (package __init__ only needs matplotlib)

    _mod('auromat.util.image', loadImage=None)
    _mod('auromat.fits')
    _mod('auromat.coordinates.ephem', EphemerisCalculator=None)

    # in-memory one-token patch of the histogram module (see module docstring)
    path = REFERENCE_ROOT + '/auromat/util/histogram.py'
    with open(path) as fp:
        src = fp.read()
    assert src.count('hist = hist[core]') == 1
    src = src.replace('hist = hist[core]', 'hist = hist[tuple(core)]')
    spec = importlib.util.spec_from_loader('auromat.util.histogram', loader=None, origin=path)
    hmod = importlib.util.module_from_spec(spec)
    hmod.__file__ = path
    exec(compile(src, path, 'exec'), hmod.__dict__)
    sys.modules['auromat.util.histogram'] = hmod
    import auromat.util
    auromat.util.histogram = hmod

    install_shims.done = True


def read_wcs_cards(path):
    """Parse an 80-column FITS header into a dict (numbers as float/int, strings stripped)."""
    with open(path, 'rb') as fp:
        raw = fp.read().decode('ascii', 'replace')
    hdr = {}
    for i in range(0, len(raw), 80):
        card = raw[i:i + 80]
        key = card[:8].strip()
        if key == 'END':
            break
        if card[8:10] != '= ' or key in ('HISTORY', 'COMMENT', ''):
            continue
        body = card[10:]
        if body.lstrip().startswith("'"):
            s = body.lstrip()[1:]
            val = s[:s.index("'")].strip()
        else:
            tok = body.split('/')[0].strip()
            if tok in ('T', 'F'):
                val = tok == 'T'
            else:
                try:
                    val = int(tok)
                except ValueError:
                    val = float(tok)
        hdr[key] = val
    return hdr


def header_time_and_camera(hdr):
    """Mirror of reference spacecraft.py:437-452 / fits.py:365-442 for headers that carry POS* cards."""
    from datetime import timedelta

    def parse(s):
        try:
            return datetime.strptime(s, '%Y-%m-%dT%H:%M:%S.%f')
        except ValueError:
            return datetime.strptime(s, '%Y-%m-%dT%H:%M:%S')
    date = parse(hdr['DATE-OBS'])
    if 'POSXSHIF' in hdr and 'DATESHIF' in hdr:
        cam = np.array([hdr['POSXSHIF'], hdr['POSYSHIF'], hdr['POSZSHIF']], dtype=np.float64)
        return date + timedelta(seconds=hdr['DATESHIF']), cam
    cam = np.array([hdr['POSX'], hdr['POSY'], hdr['POSZ']], dtype=np.float64)
    return date, cam
