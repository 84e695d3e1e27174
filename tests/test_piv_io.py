"""The frame reader that stands in for skimage.io.imread (examples/PyHSchunck_Fs3_4_PyrLvls2.py:26,129-130)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_gold

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opticalflow-ri_amd")
REF_PAIR = "/root/reference/examples/testImages/Bits08/Ni06"


def _frames():
    rng = np.random.default_rng(5)
    smooth = (np.add.outer(np.arange(70), np.arange(93)) % 251).astype(np.uint8)      # long runs and repeats for the coders
    return {"noise8": rng.integers(0, 256, (61, 47), dtype=np.uint8), "smooth8": smooth,
            "flat8": np.full((300, 500), 7, np.uint8),                                # LZW past the 9/10/11-bit code widths
            "noise16": rng.integers(0, 65536, (33, 58), dtype=np.uint16),
            "smooth16": (np.add.outer(np.arange(40), np.arange(64)) * 257 % 65536).astype(np.uint16)}


@pytest.mark.parametrize("compression", [None, "packbits", "tiff_lzw", "tiff_adobe_deflate"])
def test_baseline_reader_matches_pillow(tmp_path, compression):
    from PIL import Image
    import piv_io
    for name, a in _frames().items():
        fn = os.path.join(str(tmp_path), "%s_%s.tif" % (name, compression))
        Image.fromarray(a).save(fn, compression=compression)
        got = piv_io.read_tiff_baseline(fn)
        assert got.dtype == a.dtype and got.shape == a.shape, (name, compression)
        np.testing.assert_array_equal(got, a)
        np.testing.assert_array_equal(piv_io.imread(fn, backend="pillow"), a)
        np.testing.assert_array_equal(piv_io.imread(fn), a)


def test_baseline_reader_big_endian_and_predictor(tmp_path):
    """Hand-made files: 'MM' byte order, 16-bit samples, several strips, horizontal differencing with Deflate."""
    import struct
    import zlib
    import piv_io
    a = (np.add.outer(np.arange(10), np.arange(12)) * 1000 % 65536).astype(np.uint16)
    for bo, pred in ((">", 1), ("<", 2), (">", 2)):
        rps, strips = 4, []
        for r0 in range(0, 10, rps):
            blk = a[r0:r0 + rps]
            if pred == 2:
                blk = np.concatenate([blk[:, :1], np.diff(blk.astype(np.int64), axis=1)], axis=1).astype(np.uint16)
            raw = blk.astype(np.dtype("u2").newbyteorder(bo)).tobytes()
            strips.append(zlib.compress(raw) if pred == 2 else raw)
        n = len(strips)
        entries = [(256, 3, 1, 12), (257, 3, 1, 10), (258, 3, 1, 16), (259, 3, 1, 8 if pred == 2 else 1), (262, 3, 1, 1),
                   (273, 4, n, None), (277, 3, 1, 1), (278, 3, 1, rps), (279, 4, n, None), (317, 3, 1, pred)]
        ifd_at = 8
        data_at = ifd_at + 2 + 12 * len(entries) + 4
        off_table, cnt_table = data_at, data_at + 4 * n
        pos = cnt_table + 4 * n
        offs = []
        for s in strips:
            offs.append(pos)
            pos += len(s)
        out = (b"MM" if bo == ">" else b"II") + struct.pack(bo + "HI", 42, ifd_at) + struct.pack(bo + "H", len(entries))
        for tag, typ, cnt, val in entries:
            if val is None:
                out += struct.pack(bo + "HHII", tag, typ, cnt, off_table if tag == 273 else cnt_table)
            elif typ == 3:
                out += struct.pack(bo + "HHIHH", tag, typ, cnt, val, 0)
            else:
                out += struct.pack(bo + "HHII", tag, typ, cnt, val)
        out += struct.pack(bo + "I", 0)
        out += struct.pack(bo + "%dI" % n, *offs) + struct.pack(bo + "%dI" % n, *[len(s) for s in strips]) + b"".join(strips)
        fn = os.path.join(str(tmp_path), "hand_%s_%d.tif" % ("be" if bo == ">" else "le", pred))
        with open(fn, "wb") as f:
            f.write(out)
        np.testing.assert_array_equal(piv_io.read_tiff_baseline(fn), a)
        np.testing.assert_array_equal(piv_io.imread(fn, backend="pillow"), a)


def test_baseline_reader_says_what_it_does_not_read(tmp_path):
    from PIL import Image
    import piv_io
    fn = os.path.join(str(tmp_path), "rgb.tif")
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(fn)
    with pytest.raises(ValueError, match="greyscale"):
        piv_io.read_tiff_baseline(fn)
    bad = os.path.join(str(tmp_path), "bad.tif")
    with open(bad, "wb") as f:
        f.write(b"not a tiff at all")
    with pytest.raises(ValueError, match="not a TIFF"):
        piv_io.read_tiff_baseline(bad)


def test_read_pair_applies_the_16_bit_range_rule(tmp_path):
    from PIL import Image
    import piv_io
    fr = _frames()
    f8, f16 = os.path.join(str(tmp_path), "a8.tif"), os.path.join(str(tmp_path), "a16.tif")
    Image.fromarray(fr["noise8"]).save(f8)
    Image.fromarray(fr["noise16"]).save(f16)
    a, b = piv_io.read_pair(f8, f8, range_rule=True)
    assert a.dtype == np.float32
    np.testing.assert_array_equal(a, fr["noise8"].astype(np.float32))
    a, b = piv_io.read_pair(f16, f16, range_rule=True)
    np.testing.assert_array_equal(a, (fr["noise16"].astype(np.float32) / 65535.0 * 255.0).astype(np.float32))
    a, b = piv_io.read_pair(f16, f16)
    np.testing.assert_array_equal(b, fr["noise16"].astype(np.float32))


def test_scripts_import_line_resolves_without_scikit_image(tmp_path):
    """`from skimage.io import imread` with the compat directory at the end of sys.path, in a fresh interpreter."""
    from PIL import Image
    a = _frames()["smooth8"]
    fn = os.path.join(str(tmp_path), "f.tif")
    Image.fromarray(a).save(fn, compression="packbits")
    code = ("import sys, os\nsys.path.append(os.path.join(%r, 'compat'))\nfrom skimage.io import imread\nimport numpy as np\n"
            "x = imread(%r).astype(np.float32)\nprint(x.shape, x.dtype, float(x.sum()))\n" % (PKG, fn))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[-1] == repr(float(a.astype(np.float64).sum())) and "float32" in out.stdout


@pytest.mark.skipif(not os.path.isdir(REF_PAIR), reason="the reference's bundled frames are only in the build container")
def test_bundled_pair_reads_as_the_committed_fixture():
    import piv_io
    g = load_gold("pair_parabolic01")
    for k, name in (("im0", "parabolic01_0.tif"), ("im1", "parabolic01_1.tif")):
        for backend in ("baseline", "pillow"):
            got = piv_io.imread(os.path.join(REF_PAIR, name), backend=backend)
            assert got.dtype == np.uint8
            np.testing.assert_array_equal(got, g[k])
