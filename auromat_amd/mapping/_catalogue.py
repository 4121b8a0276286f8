"""
What the file-backed mapping providers share (host code, no GPU): a time-ordered catalogue of entries with nearest-date lookup,
and the conversion of an exported image variable back to the unsigned pixels it was written from.  The providers of the
reference (mapping/cdf.py:19-77, mapping/netcdf.py:20-76, mapping/spacecraft.py:40-300) each carry their own copy of the date
lookup; here it is one small class over ``bisect``.
"""
import bisect

import numpy as np
import numpy.ma as ma


class DateCatalogue(object):
    """Entries (date, payload, label) kept sorted by date.  Two entries with the same date are an error (the label says where
    each came from)."""

    def __init__(self, entries, kind='files'):
        rows = sorted(entries, key=lambda e: e[0])
        for a, b in zip(rows, rows[1:]):
            if a[0] == b[0]:
                raise ValueError('%s holds the date %s twice: %s and %s' % (kind, a[0], a[2], b[2]))
        self.dates = [r[0] for r in rows]
        self.payloads = [r[1] for r in rows]
        self.labels = [r[2] for r in rows]

    def __len__(self):
        return len(self.dates)

    @property
    def span(self):
        """(first date, last date)"""
        return self.dates[0], self.dates[-1]

    def nearest(self, date):
        """-> (position of the entry closest in time, its distance in seconds); the earlier one of two equally close."""
        if not self.dates:
            raise ValueError('the catalogue is empty')
        hi = bisect.bisect_left(self.dates, date)
        around = [k for k in (hi - 1, hi) if 0 <= k < len(self.dates)]
        best = min(around, key=lambda k: (abs((self.dates[k] - date).total_seconds()), k))
        return best, abs((self.dates[best] - date).total_seconds())

    def within(self, date, seconds):
        """Is there an entry at most `seconds` away from `date`?"""
        return bool(self.dates) and self.nearest(date)[1] <= seconds

    def pick(self, date, seconds):
        """The payload of the entry closest to `date`; ValueError when that is further away than `seconds`."""
        k, off = self.nearest(date)
        if off > seconds:
            raise ValueError('nothing within %s s of %s (the closest entry, %s, is %.3f s away)' % (seconds, date, self.dates[k], off))
        return self.payloads[k]

    def between(self, first=None, last=None):
        """Payloads in time order from `first` to `last`, both inclusive, either open when None."""
        lo = 0 if first is None else bisect.bisect_left(self.dates, first)
        hi = len(self.dates) if last is None else bisect.bisect_right(self.dates, last)
        return self.payloads[lo:hi]


# an exported image is stored in the next larger SIGNED type when it needs a fill value (export/netcdf.py, export/cdf.py): back
_NARROWER = {np.dtype(np.int16): np.uint8, np.dtype(np.int32): np.uint16, np.dtype(np.int64): np.uint32}


def unsigned_pixels(stored, fill=None):
    """Image variable as read from an exported file -> masked array of the unsigned pixel type it was written from: values equal
    to `fill` masked; an unsigned variable is taken as it is; any other type is refused (NotImplementedError)."""
    stored = np.asarray(stored)
    if stored.dtype.kind == 'u':
        return ma.masked_array(stored)
    target = _NARROWER.get(stored.dtype)
    if target is None:
        raise NotImplementedError('image variables of type %s are not supported' % stored.dtype)
    pixels = ma.masked_array(stored) if fill is None else ma.masked_equal(stored, fill, copy=False)
    seen = pixels.compressed()
    if seen.size and not (0 <= seen.min() and seen.max() <= np.iinfo(target).max):
        raise ValueError('image values outside the range of %s' % np.dtype(target).name)
    return pixels.astype(target)
