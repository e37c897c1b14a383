"""Minimal FITS writer for the uint16 images the pipeline saves.

The reference writes them with astropy (`fits.PrimaryHDU(img, header=hdr).writeto`,
Solex_recon.py:80-82, 137-152; solex_util.py:204-206, 584-587).  For a uint16 array
astropy stores big-endian int16 = value - 32768 with BITPIX=16, BSCALE=1, BZERO=32768,
keeps the extra cards of make_header (BIN1, BIN2, EXPTIME; solex_util.py:147-161) and
overrides its BITPIX/NAXIS*/BZERO/BSCALE.  tests/golden/g9_fits.npz holds astropy's bytes
for a small array; test_host_cpu.py compares this writer against them byte for byte.
"""
import numpy as np

BLOCK = 2880
_STRUCTURAL = ('SIMPLE', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'BSCALE', 'BZERO', 'EXTEND')


class Header(dict):
    """Insertion-ordered card dictionary (the subset of astropy.io.fits.Header the path uses)."""


def make_header(rdr):
    """Same cards as the reference's make_header (solex_util.py:147-161)."""
    hdr = Header()
    hdr['SIMPLE'] = 'T'
    hdr['BITPIX'] = 32
    hdr['NAXIS'] = 2
    hdr['NAXIS1'] = rdr.iw
    hdr['NAXIS2'] = rdr.ih
    hdr['BZERO'] = 0
    hdr['BSCALE'] = 1
    hdr['BIN1'] = 1
    hdr['BIN2'] = 1
    hdr['EXPTIME'] = 0
    return hdr


def _card(key, value, comment=''):
    if isinstance(value, bool):
        text = 'T' if value else 'F'
    elif isinstance(value, (int, np.integer)):
        text = '%d' % int(value)
    elif isinstance(value, (float, np.floating)):
        text = repr(float(value)).upper()
    else:
        text = "'%-8s'" % str(value).replace("'", "''")
        card = '%-8s= %-20s' % (key, text)
        if comment:
            card += ' / ' + comment
        return card[:80].ljust(80)
    card = '%-8s= %20s' % (key, text)
    if comment:
        card += ' / ' + comment
    return card[:80].ljust(80)


def fits_bytes(array, header=None):
    array = np.asarray(array)
    if array.dtype == np.float64 and array.ndim == 2:
        # a de-vignetted frame is float64 in the reference (solex_util.py:654): BITPIX = -64, no scaling
        cards = [_card('SIMPLE', True, 'conforms to FITS standard'), _card('BITPIX', -64, 'array data type'),
                 _card('NAXIS', 2, 'number of array dimensions'), _card('NAXIS1', array.shape[1]),
                 _card('NAXIS2', array.shape[0])]
        for key, value in (header or {}).items():
            if key.upper() not in _STRUCTURAL:
                cards.append(_card(key.upper(), value))
        cards.append('END'.ljust(80))
        head = ''.join(cards).encode('ascii')
        head += b' ' * (-len(head) % BLOCK)
        data = array.astype('>f8').tobytes()
        return head + data + b'\0' * (-len(data) % BLOCK)
    if array.dtype != np.uint16 or array.ndim != 2:
        raise TypeError('fits_bytes writes 2-D uint16 (or float64) images, got %s %s' % (array.dtype, array.shape))
    cards = [_card('SIMPLE', True, 'conforms to FITS standard'), _card('BITPIX', 16, 'array data type'),
             _card('NAXIS', 2, 'number of array dimensions'), _card('NAXIS1', array.shape[1]),
             _card('NAXIS2', array.shape[0])]
    for key, value in (header or {}).items():
        if key.upper() not in _STRUCTURAL:
            cards.append(_card(key.upper(), value))
    cards += [_card('BSCALE', 1), _card('BZERO', 32768), 'END'.ljust(80)]
    head = ''.join(cards).encode('ascii')
    head += b' ' * (-len(head) % BLOCK)
    # value - 32768 as big-endian int16: flip the top bit, swap the bytes (two passes over the image, no int32 detour)
    data = (np.ascontiguousarray(array) ^ np.uint16(0x8000)).byteswap().tobytes()
    return head + data + b'\0' * (-len(data) % BLOCK)


def write_fits(path, array, header=None):
    with open(path, 'wb') as f:
        f.write(fits_bytes(array, header))


def read_fits_u16(path):
    """Inverse of write_fits (tests and the CLI round trip)."""
    raw = open(path, 'rb').read()
    cards = {}
    pos = 0
    while True:
        card = raw[pos:pos + 80].decode('ascii')
        pos += 80
        if card.startswith('END'):
            break
        if '=' in card[:10]:
            cards[card[:8].strip()] = card[10:].split('/')[0].strip()
    pos += -pos % BLOCK
    w, h = int(cards['NAXIS1']), int(cards['NAXIS2'])
    data = np.frombuffer(raw, dtype='>i2', count=w * h, offset=pos).astype(np.int32) + int(float(cards.get('BZERO', 0)))
    return data.reshape(h, w).astype(np.uint16), cards
