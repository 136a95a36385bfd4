"""The file hand-off of the drop-in boundary: the shim's PLY reader + normal cleaning against the
REFERENCE'S OWN reader (S4/io/io.cc + io_ply.h, compiled unmodified into oracle/_ref, followed by
Utils::CleanInvalidNormals as in S4/super4pcs_test.cc:58-89) on files in the layout
pcl::io::savePLYFile writes for PointXYZRGBNormal.  A committed fixture (PLY text + what the
reference reader returned) lets the comparison run where /root/reference is absent."""
import ctypes as C
import os

import numpy as np
import pytest

from _checkers import have_ref, ref_lib
from _dropin import write_png_with_filter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim", "libsuper4pcs.so")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ply_reader.npz")
_f = C.POINTER(C.c_float)

pytestmark = pytest.mark.skipif(not os.path.exists(SHIM), reason="shim/libsuper4pcs.so not built (needs Eigen)")


def ascii_ply(xyz, nrm):
    """PointXYZRGBNormal as pcl::io::savePLYFile writes it (ASCII): x y z r g b nx ny nz curvature."""
    hdr = ("ply\nformat ascii 1.0\ncomment PCL generated\nelement vertex %d\nproperty float x\nproperty float y\n"
           "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\n"
           "property float ny\nproperty float nz\nproperty float curvature\nelement camera 1\n"
           "property float view_px\nend_header\n") % len(xyz)
    body = "".join("%.9g %.9g %.9g %d %d %d %.9g %.9g %.9g 0\n" % (*p, 10 + i % 200, 20, 30, *q)
                   for i, (p, q) in enumerate(zip(xyz, nrm)))
    return (hdr + body + "0\n").encode()


def tricky_cloud(rng, n):
    xyz = (rng.standard_normal((n, 3)) * rng.choice([1e-3, 0.1, 1.0, 37.0], (n, 1))).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    nrm[0] = 0.0                                  # zero normal -> stays zero
    nrm[1] = [1e-3, 0, 0]                         # tiny normal: normalised BEFORE the 0.01 test
    nrm[2] = [0.05, 0.05, 0.0]                    # squared norm 0.005 < 0.01 before normalisation
    nrm[3] *= 1e4                                 # long normal
    nrm[4] = [0, -0.0, 1]
    return xyz, nrm


def read_with(fn, path, cap=100000):
    xyz, nrm = np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32)
    n = fn(path.encode(), xyz.ctypes.data_as(_f), nrm.ctypes.data_as(_f), cap)
    assert n >= 0, "reader failed"
    return xyz[:n].copy(), nrm[:n].copy()


def shim_reader():
    L = C.CDLL(SHIM)
    L.super4pcs_shim_read_cloud.argtypes = [C.c_char_p, _f, _f, C.c_int]
    return L.super4pcs_shim_read_cloud


def test_shim_reader_matches_the_committed_reference_output(tmp_path):
    g = np.load(GOLD)
    for k in range(int(g["n_files"])):
        p = str(tmp_path / f"cloud_{k}.ply")
        with open(p, "wb") as f:
            f.write(g[f"ply_{k}"].tobytes())
        xyz, nrm = read_with(shim_reader(), p)
        assert np.array_equal(xyz, g[f"xyz_{k}"])
        assert np.array_equal(nrm, g[f"nrm_{k}"])


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (build container)")
def test_shim_reader_matches_the_reference_reader_live(tmp_path):
    L = ref_lib()
    L.ref_read_cloud.argtypes = [C.c_char_p, _f, _f, C.c_int]
    rng = np.random.default_rng(17)
    for k, n in enumerate((1, 5, 64, 2000)):
        xyz, nrm = tricky_cloud(rng, max(n, 5))
        xyz, nrm = xyz[:n], nrm[:n]
        p = str(tmp_path / f"c{k}.ply")
        with open(p, "wb") as f:
            f.write(ascii_ply(xyz, nrm))
        x_ref, n_ref = read_with(L.ref_read_cloud, p)
        x_shim, n_shim = read_with(shim_reader(), p)
        assert len(x_ref) == n
        assert np.array_equal(x_shim, x_ref)
        assert np.array_equal(n_shim, n_ref)
        assert np.array_equal(x_ref, xyz)          # %.9g round-trips float32


def test_shim_png_decoder_matches_pil(tmp_path):
    """The probability image (cv::imread of a CV_16UC1 PNG at base.cc:317; OpenCV is not in this image)
    is decoded by the shim's own inflate + unfilter code: compare with PIL on 16- and 8-bit greyscale
    images whose rows make the encoder pick different scanline filters."""
    from PIL import Image
    L = C.CDLL(SHIM)
    L.super4pcs_shim_read_png.argtypes = [C.c_char_p, C.POINTER(C.c_ushort), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:97, 0:131]
    images = [
        rng.integers(0, 65536, (48, 64)).astype(np.uint16),                       # noise
        ((xx * 500 + yy * 37) % 65536).astype(np.uint16),                         # ramps (Sub / Up / Paeth rows)
        np.where((xx // 16 + yy // 16) % 2 == 0, 10000, 0).astype(np.uint16),     # flat blocks
        np.zeros((1, 1), np.uint16),
        rng.integers(0, 256, (33, 21)).astype(np.uint8),                          # 8-bit
        (np.clip(10000 * np.exp(-((xx - 60) ** 2 + (yy - 40) ** 2) / 900.0), 0, 10000)).astype(np.uint16),
    ]
    for k, img in enumerate(images):
        p = str(tmp_path / f"img_{k}.png")
        Image.fromarray(img).save(p)
        want = np.array(Image.open(p)).astype(np.uint16)
        out = np.zeros(img.size, np.uint16)
        rows, cols = C.c_int(0), C.c_int(0)
        rc = L.super4pcs_shim_read_png(p.encode(), out.ctypes.data_as(C.POINTER(C.c_ushort)), out.size,
                                       C.byref(rows), C.byref(cols))
        assert rc == 0 and (rows.value, cols.value) == img.shape
        assert np.array_equal(out.reshape(img.shape), want)
    assert L.super4pcs_shim_read_png(str(tmp_path / "missing.png").encode(), None, 0, C.byref(rows), C.byref(cols)) == -1


def test_shim_png_decoder_every_scanline_filter(tmp_path):
    """Each of the five scanline filters on every row (Sub is what cv::imwrite writes, base.cc:317 reads it back) and a
    mix that puts every filter above every other: 16- and 8-bit, widths that end inside, on and just past the 16-byte
    blocks the Sub / Up / byte-swap loops take."""
    L = C.CDLL(SHIM)
    L.super4pcs_shim_read_png.argtypes = [C.c_char_p, C.POINTER(C.c_ushort), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rng = np.random.default_rng(23)
    mix = [0, 1, 2, 3, 4, 0, 2, 4, 1, 3, 0, 3, 1, 4, 2, 0, 4, 3, 2, 1, 1, 0]
    k = 0
    for dtype, hi in ((np.uint16, 65536), (np.uint8, 256)):
        for w in (1, 2, 7, 8, 9, 15, 16, 17, 23, 33, 64, 131):
            h = 23
            img = rng.integers(0, hi, (h, w)).astype(dtype)
            img[::3] = (np.arange(w) * 301 % hi).astype(dtype)          # rows with structure too
            for ftype in (0, 1, 2, 3, 4, mix):
                p = str(tmp_path / f"f_{k}.png")
                k += 1
                write_png_with_filter(img, ftype, p)
                out = np.zeros(img.size, np.uint16)
                rows, cols = C.c_int(0), C.c_int(0)
                rc = L.super4pcs_shim_read_png(p.encode(), out.ctypes.data_as(C.POINTER(C.c_ushort)), out.size, C.byref(rows), C.byref(cols))
                assert rc == 0 and (rows.value, cols.value) == img.shape, (dtype, w, ftype)
                assert np.array_equal(out.reshape(img.shape), img.astype(np.uint16)), (dtype, w, ftype)
    # the writer above against an independent decoder, once
    from PIL import Image
    img = rng.integers(0, 65536, (9, 21)).astype(np.uint16)
    p = str(tmp_path / "pil_check.png")
    write_png_with_filter(img, mix, p)
    assert np.array_equal(np.array(Image.open(p)).astype(np.uint16), img)


def test_shim_png_decoder_stops_after_the_last_row_needed(tmp_path):
    """The file hand-off publishes the last image row the segment's points fall on (pgp_image_rows_needed) and the
    decoder stops after the band of rows that holds it: every row up to it equals PIL's, later bands stay zero."""
    from PIL import Image
    L = C.CDLL(SHIM)
    L.super4pcs_shim_read_png_rows.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_ushort), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    yy, xx = np.mgrid[0:200, 0:150]
    rng = np.random.default_rng(9)
    for k, img in enumerate((((xx * 300 + yy * 91) % 65536).astype(np.uint16), rng.integers(1, 65536, (200, 150)).astype(np.uint16))):
        p = str(tmp_path / f"rows_{k}.png")
        Image.fromarray(img).save(p)
        want = np.array(Image.open(p)).astype(np.uint16)
        for last in (-1, 0, 31, 32, 100, 199, 5000):
            out = np.zeros(img.size, np.uint16)
            rows, cols = C.c_int(0), C.c_int(0)
            rc = L.super4pcs_shim_read_png_rows(p.encode(), last, out.ctypes.data_as(C.POINTER(C.c_ushort)), out.size, C.byref(rows), C.byref(cols))
            assert rc == 0 and (rows.value, cols.value) == img.shape
            got = out.reshape(img.shape)
            n = min(200, max(last, 0) + 1)
            assert np.array_equal(got[:n], want[:n])
            done = min(200, (max(last, 0) // 32 + 1) * 32)          # whole bands of 32 rows
            assert np.array_equal(got[:done], want[:done]) and not got[done:].any()


def test_number_parser_rounds_like_the_references_fscanf():
    """The reader's decimal path (<= 15 digits, |exponent| <= 22: one exact multiply / divide to a double, a cast
    to float unless the double sits at a float midpoint) and its strtof fallback against the C library's strtof
    = what fscanf("%f", &float) stores (S4/io/io_ply.h:272-296): ONE rounding, decimal -> float.  Includes
    tokens placed on and next to float midpoints, where rounding through a double goes wrong."""
    L = C.CDLL(SHIM)
    L.super4pcs_shim_parse_numbers.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int]
    libc = C.CDLL("libc.so.6")
    libc.strtof.restype = C.c_float
    libc.strtof.argtypes = [C.c_char_p, C.c_void_p]
    rng = np.random.default_rng(5)
    toks = []
    for _ in range(20000):
        nd = int(rng.integers(1, 20))
        digits = "".join(rng.choice(list("0123456789"), nd))
        cut = int(rng.integers(0, nd + 1))
        tok = ("-" if rng.random() < 0.4 else "") + (digits[:cut] or "0") + ("." + digits[cut:] if cut < nd else "")
        if rng.random() < 0.3:
            tok += "e%+d" % int(rng.integers(-30, 31))
        toks.append(tok)
    # decimals at / around the midpoint of two adjacent floats, written with 9 .. 17 significant digits
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    for _ in range(4000):
        f = np.float32(rng.uniform(-1, 1) * 10.0 ** rng.integers(-6, 3))
        g = np.nextafter(f, np.float32(np.inf))
        mid = (Decimal(float(f)) + Decimal(float(g))) / 2
        for nd in (9, 12, 15, 17):
            q = mid.quantize(Decimal(1).scaleb(mid.adjusted() - nd + 1))
            toks.append(format(q, "f"))
            toks.append(format(q + Decimal(1).scaleb(mid.adjusted() - nd + 1), "f"))
        toks.append(format(mid, "f"))                       # the exact midpoint (many digits): ties-to-even
    toks += ["0", "-0", "0.0", "1e22", "1e23", "9007199254740993", "0.1", "123456789012345", "1234567890123456",
             "4.9e-324", "1.7976931348623157e308", "00012.5000", ".5", "5.", "+3.25", "1E5", "1e-22", "1e-23",
             "1e-40", "3.4028235e38", "3.5e38", "1.17549435e-38", "1e-45", "7e-46"]
    out = np.zeros(len(toks), np.float32)
    n = L.super4pcs_shim_parse_numbers((" ".join(toks) + "\n").encode(), out.ctypes.data_as(C.POINTER(C.c_float)), len(toks))
    assert n == len(toks)
    want = np.array([libc.strtof(t.encode(), None) for t in toks], np.float32)
    bad = np.flatnonzero(out.view(np.uint32) != want.view(np.uint32))
    assert len(bad) == 0, [(toks[i], out[i], want[i]) for i in bad[:5]]
    # and the double-rounded value really does differ on some of these tokens: the test has teeth
    with np.errstate(over="ignore"):
        twice = np.array([np.float32(float(t)) for t in toks], np.float32)
    assert (twice.view(np.uint32) != want.view(np.uint32)).sum() > 0


def test_shim_inflater_matches_zlib():
    """shim/fast_inflate.h (the probability PNG's inflate): stored, fixed-Huffman and dynamic blocks, long and overlapping
    matches, run lengths of 16-bit values, incompressible data, empty input -- byte for byte what zlib produces, in one
    run and resumed every few hundred bytes; damaged streams are refused or fail the checksum, never crash."""
    import zlib
    L = C.CDLL(SHIM)
    L.super4pcs_shim_inflate.restype = C.c_longlong
    L.super4pcs_shim_inflate.argtypes = [C.c_char_p, C.c_longlong, C.POINTER(C.c_ubyte), C.c_longlong, C.c_longlong]
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:120, 0:160]
    blob = (np.clip(10000 * np.exp(-((xx - 80) ** 2 + (yy - 60) ** 2) / 900.0), 0, 10000)).astype(">u2").tobytes()
    datas = [
        b"", b"a", b"abc" * 1000, bytes(70000), blob,
        rng.integers(0, 256, 50000, dtype=np.uint8).tobytes(),                       # incompressible
        rng.integers(0, 4, 80000, dtype=np.uint8).tobytes(),                         # short codes, many matches
        np.repeat(rng.integers(0, 65536, 300).astype("<u2"), rng.integers(1, 700, 300)).tobytes(),   # runs of 16-bit values
        bytes(rng.integers(0, 256, 7, dtype=np.uint8)) * 9000,                       # period 7: overlapping byte copies
        (b"0123456789abcdef" * 4096) + rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(),
        " ".join(str(v) for v in rng.normal(size=20000)).encode(),                   # text: long dynamic tables
    ]

    def streams(d):
        for level in (0, 1, 6, 9):
            yield zlib.compress(d, level)
        for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, strategy)
            yield c.compress(d) + c.flush()
        c = zlib.compressobj(6, zlib.DEFLATED, 9)      # a small window
        yield c.compress(d) + c.flush()
        c = zlib.compressobj(6)                        # several blocks, an empty stored block between them
        half = len(d) // 2
        yield c.compress(d[:half]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(d[half:]) + c.flush()

    n_streams = 0
    for d in datas:
        for z in streams(d):
            assert zlib.decompress(z) == d
            for step in (0, 1, 257, 4096):
                out = np.zeros(max(len(d), 1) + 16, np.uint8)
                n = L.super4pcs_shim_inflate(z, len(z), out.ctypes.data_as(C.POINTER(C.c_ubyte)), len(d), step)
                assert n == len(d), (len(d), len(z), step, n)
                assert out[:len(d)].tobytes() == d
            n_streams += 1
    assert n_streams == len(datas) * 10
    # damage: every refusal is an error code (the reader then falls back to zlib), never a wrong answer accepted
    z = zlib.compress(datas[4], 6)
    out = np.zeros(len(datas[4]) + 16, np.uint8)
    po = out.ctypes.data_as(C.POINTER(C.c_ubyte))
    assert L.super4pcs_shim_inflate(z[:-5], len(z) - 5, po, len(datas[4]), 0) < 0                     # truncated
    assert L.super4pcs_shim_inflate(z, len(z), po, len(datas[4]) - 1, 0) < 0                           # output too small
    for k in rng.integers(2, len(z) - 4, 200):
        bad = bytearray(z)
        bad[k] ^= 1 << int(rng.integers(0, 8))
        n = L.super4pcs_shim_inflate(bytes(bad), len(bad), po, len(datas[4]), 0)
        assert n < 0 or out[:len(datas[4])].tobytes() == datas[4]       # (a flipped bit in unused header bits may be harmless)


def test_shim_png_reader_takes_zlib_when_told(tmp_path, monkeypatch):
    """PGP_SHIM_ZLIB=1 (the A/B knob, also the path a refused stream takes) decodes the same pixels."""
    from PIL import Image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 65536, (70, 90)).astype(np.uint16)
    p = str(tmp_path / "z.png")
    Image.fromarray(img).save(p)
    import subprocess, sys
    code = ("import ctypes as C, numpy as np, sys\n"
            "L = C.CDLL(sys.argv[1]); out = np.zeros(6300, np.uint16); r = C.c_int(0); c = C.c_int(0)\n"
            "L.super4pcs_shim_read_png.argtypes = [C.c_char_p, C.POINTER(C.c_ushort), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]\n"
            "assert L.super4pcs_shim_read_png(sys.argv[2].encode(), out.ctypes.data_as(C.POINTER(C.c_ushort)), 6300, C.byref(r), C.byref(c)) == 0\n"
            "sys.stdout.write(out.tobytes().hex())\n")
    outs = []
    for env in ({}, {"PGP_SHIM_ZLIB": "1"}):
        r = subprocess.run([sys.executable, "-c", code, SHIM, p], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1] == img.tobytes().hex()

