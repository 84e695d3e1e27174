"""Reading PIV frames without scikit-image.

The reference's scripts read their frames with ``skimage.io.imread(fn).astype(np.float32)``
(examples/PyHSchunck_Fs3_4_PyrLvls2.py:26,129-130; benchmark_of_methods.py:125-126), which for the single-page greyscale
TIFFs of a PIV camera returns the stored samples unchanged as a (rows, cols) array of the file's own sample type.
``imread`` here returns the same array from whatever reader the machine has: tifffile, Pillow, or - with neither - the
baseline TIFF reader below (strips, 8/16/32-bit unsigned or float32 greyscale, no compression / PackBits / LZW / Deflate,
horizontal predictor, either byte order).  ``compat/skimage/io.py`` offers it under the name the scripts import, for
machines without scikit-image: append ``<this directory>/compat`` to sys.path (append, so that an installed scikit-image
wins).
"""
import struct
import zlib

import numpy as np

_TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8, 17: 8, 18: 8}
_TYPE_FMT = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i", 16: "Q", 17: "q"}


def _unpackbits(data, want):
    out = bytearray()
    i, n = 0, len(data)
    while i < n and len(out) < want:
        c = data[i]
        i += 1
        if c < 128:
            out += data[i:i + c + 1]
            i += c + 1
        elif c > 128:
            out += data[i:i + 1] * (257 - c)
            i += 1
    return bytes(out[:want])


def _unlzw(data, want):
    """TIFF's LZW: codes of 9 to 12 bits, most significant bit first, 256 = clear, 257 = end, widths grow one code early."""
    out = bytearray()
    table = [bytes((i,)) for i in range(256)] + [b"", b""]
    width, prev = 9, None
    acc = nbits = 0
    for byte in data:
        acc = (acc << 8) | byte
        nbits += 8
        while nbits >= width:
            code = (acc >> (nbits - width)) & ((1 << width) - 1)
            nbits -= width
            if code == 257:
                return bytes(out[:want])
            if code == 256:
                table = table[:258]
                width, prev = 9, None
                continue
            if prev is None:
                entry = table[code]
            else:
                if code < len(table):
                    entry = table[code]
                elif code == len(table):
                    entry = prev + prev[:1]
                else:
                    raise ValueError("corrupt LZW stream")
                table.append(prev + entry[:1])
            out += entry
            prev = entry
            if len(table) >= (1 << width) - 1 and width < 12:
                width += 1
        if len(out) >= want:
            break
    return bytes(out[:want])


def read_tiff_baseline(path):
    """The first image of a greyscale TIFF as a (rows, cols) array of its own sample type; ValueError for anything this
    small reader does not cover (tiles, colour, bit depths other than 8/16/32, other compressions)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:2] == b"II":
        bo = "<"
    elif buf[:2] == b"MM":
        bo = ">"
    else:
        raise ValueError("%s: not a TIFF file" % path)
    magic, = struct.unpack(bo + "H", buf[2:4])
    if magic != 42:
        raise ValueError("%s: not a classic TIFF (BigTIFF is not read here)" % path)
    ifd, = struct.unpack(bo + "I", buf[4:8])
    count, = struct.unpack(bo + "H", buf[ifd:ifd + 2])
    tags = {}
    for e in range(count):
        o = ifd + 2 + 12 * e
        tag, typ, n = struct.unpack(bo + "HHI", buf[o:o + 8])
        size = _TYPE_SIZE.get(typ, 1) * n
        at = o + 8 if size <= 4 else struct.unpack(bo + "I", buf[o + 8:o + 12])[0]
        if typ in _TYPE_FMT:
            tags[tag] = struct.unpack(bo + "%d%s" % (n, _TYPE_FMT[typ]), buf[at:at + size])
    def one(tag, default=None):
        v = tags.get(tag)
        return v[0] if v else default
    cols, rows = one(256), one(257)
    if not cols or not rows:
        raise ValueError("%s: no image size" % path)
    if 322 in tags or 324 in tags:
        raise ValueError("%s: tiled TIFFs are not read here" % path)
    if one(277, 1) != 1:
        raise ValueError("%s: %d samples per pixel; greyscale frames only" % (path, one(277)))
    bits, fmt, comp, pred = one(258, 1), one(339, 1), one(259, 1), one(317, 1)
    if (bits, fmt) in ((8, 1), (16, 1), (32, 1)):
        dt = np.dtype("u%d" % (bits // 8))
    elif (bits, fmt) in ((8, 2), (16, 2), (32, 2)):
        dt = np.dtype("i%d" % (bits // 8))
    elif (bits, fmt) == (32, 3):
        dt = np.dtype("f4")
    else:
        raise ValueError("%s: %d-bit samples of format %d are not read here" % (path, bits, fmt))
    if pred not in (1, 2) or (pred == 2 and dt.kind == "f"):
        raise ValueError("%s: predictor %d is not read here" % (path, pred))
    rps = min(one(278, rows), rows)
    offsets, counts = tags.get(273), tags.get(279)
    nstrips = (rows + rps - 1) // rps
    if not offsets or len(offsets) != nstrips:
        raise ValueError("%s: strip table does not match the image" % path)
    if not counts:
        if comp != 1:
            raise ValueError("%s: compressed strips without byte counts" % path)
        counts = [min(rps, rows - s * rps) * cols * dt.itemsize for s in range(nstrips)]
    out = np.empty((rows, cols), dt)
    for s in range(nstrips):
        r0 = s * rps
        nr = min(rps, rows - r0)
        want = nr * cols * dt.itemsize
        raw = buf[offsets[s]:offsets[s] + counts[s]]
        if comp == 1:
            raw = raw[:want]
        elif comp == 32773:
            raw = _unpackbits(raw, want)
        elif comp == 5:
            raw = _unlzw(raw, want)
        elif comp in (8, 32946):
            raw = zlib.decompress(raw)[:want]
        else:
            raise ValueError("%s: compression %d is not read here" % (path, comp))
        if len(raw) != want:
            raise ValueError("%s: strip %d is short (%d of %d bytes)" % (path, s, len(raw), want))
        block = np.frombuffer(raw, dt.newbyteorder(bo)).reshape(nr, cols).astype(dt)
        if pred == 2:
            block = np.cumsum(block, axis=1, dtype=dt)        # differences along a row, modulo the sample width
        out[r0:r0 + nr] = block
    if one(262, 1) == 0 and dt.kind == "u":                     # WhiteIsZero
        out = (np.iinfo(dt).max - out).astype(dt)
    return out


def imread(path, backend=None):
    """The image file as an array, as ``skimage.io.imread`` returns it for a single-page greyscale frame.
    ``backend``: 'tifffile', 'pillow' or 'baseline' to force a reader; by default the first one that is installed
    (the baseline reader for .tif / .tiff only)."""
    order = [backend] if backend else ["tifffile", "pillow", "baseline"]
    last = None
    for b in order:
        try:
            if b == "tifffile":
                import tifffile
                return np.asarray(tifffile.imread(path))
            if b == "pillow":
                from PIL import Image
                with Image.open(path) as im:
                    return np.array(im)
            if b == "baseline":
                return read_tiff_baseline(path)
            raise ValueError("unknown backend %r" % (b,))
        except ImportError as e:
            last = e
    raise ImportError("no image reader available for %s (%s)" % (path, last))


def read_pair(fn1, fn2, range_rule=False):
    """Two frames as float32, the way the scripts prepare them (examples/PyHSchunck_Fs3_4_PyrLvls2.py:129-130);
    range_rule: frames with values above 255 are scaled from the 16-bit to the 8-bit range
    (benchmark_of_methods.py:134-137)."""
    a, b = imread(fn1).astype(np.float32), imread(fn2).astype(np.float32)
    if range_rule and a.max() > 255:
        a = (a / 65535.0 * 255.0).astype(np.float32)
        b = (b / 65535.0 * 255.0).astype(np.float32)
    return a, b
