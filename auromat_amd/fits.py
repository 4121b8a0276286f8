"""
FITS header helpers under the reference's names (auromat/fits.py) — the part of that module the mapping path uses:
reading / writing the header-only FITS files astrometry.net produces (``.wcs``), photo time and spacecraft position
cards.  astropy is not needed: a FITS header is 80-column ASCII cards in 2880-byte blocks, which is all a ``.wcs``
file holds.  (The star-catalogue and xyls helpers of the reference's module belong to its solving layer and are not
part of this package.)
"""
import json

from .mapping.spacecraft import (getPhotoTime, getShiftedPhotoTime, getShiftedSpacecraftPosition,  # noqa: F401
                                 getSpacecraftPosition)


def _parse_value(val):
    if val.lstrip().startswith("'"):
        # a character string: quotes doubled inside, trailing blanks insignificant
        body = val.lstrip()[1:]
        out, i = [], 0
        while i < len(body):
            if body[i] == "'":
                if i + 1 < len(body) and body[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                break
            out.append(body[i])
            i += 1
        return ''.join(out).rstrip()
    val = val.split('/')[0].strip()
    if val in ('T', 'F'):
        return val == 'T'
    try:
        return int(val)
    except ValueError:
        try:
            return float(val.replace('D', 'E'))
        except ValueError:
            return val


def readHeader(filePath):
    """Return the primary FITS header of `filePath` as a dict (reference fits.py:29-31 returns astropy's Header; the
    mapping code only reads cards by keyword).  A ``.json`` file with the same cards is accepted as well."""
    if filePath.endswith('.json'):
        with open(filePath) as fp:
            return json.load(fp)
    with open(filePath, 'rb') as fp:
        raw = fp.read().decode('ascii', 'replace')
    hdr = {}
    for i in range(0, len(raw) - 79, 80):
        card = raw[i:i + 80]
        key = card[:8].strip()
        if key == 'END':
            break
        if card[8:10] != '= ' or not key:
            continue          # COMMENT / HISTORY / blank cards
        hdr[key] = _parse_value(card[10:])
    return hdr


def _format_card(key, value):
    if isinstance(value, bool):
        val = '%20s' % ('T' if value else 'F')
    elif isinstance(value, str):
        val = ("'%s'" % value.replace("'", "''").ljust(8)).ljust(20)
    elif isinstance(value, int):
        val = '%20d' % value
    else:
        r = repr(float(value))
        val = '%20s' % (r.upper() if 'e' in r else r)
    card = '%-8s= %s' % (key, val)
    if len(card) > 80:
        raise ValueError('FITS card too long: ' + card)
    return card.ljust(80)


def writeHeader(filePath, header, overwrite=False):
    """Create a header-only FITS file from a dict of cards (reference fits.py:33-41): SIMPLE / BITPIX / NAXIS first,
    80-column cards, END, padded to 2880 bytes."""
    import os
    if os.path.exists(filePath) and not overwrite:
        raise IOError(filePath + ' exists')
    cards = [_format_card('SIMPLE', True), _format_card('BITPIX', 8), _format_card('NAXIS', 0)]
    for key, value in header.items():
        if key in ('SIMPLE', 'BITPIX', 'NAXIS', 'END') or value is None:
            continue
        cards.append(_format_card(key, value))
    cards.append('END'.ljust(80))
    raw = ''.join(cards)
    raw += ' ' * (-len(raw) % 2880)
    with open(filePath, 'wb') as fp:
        fp.write(raw.encode('ascii'))


__all__ = ['readHeader', 'writeHeader', 'getPhotoTime', 'getShiftedPhotoTime', 'getSpacecraftPosition',
           'getShiftedSpacecraftPosition']
