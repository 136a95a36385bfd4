"""The drop-in's file readers under AddressSanitizer + UBSan: shim/file_readers.h (PLY, PNG) and
shim/fast_inflate.h are built on their own (`make -C shim asan`: no Eigen, no libpgp, CPU only) and fed seeded mutations
of valid files plus the crafted headers that used to be trusted -- `element vertex 2000000000` over a 100 KB file, an
IHDR of 50 000 x 50 000 pixels, an IHDR chunk shorter than 13 bytes.  Failure mode: refused (the drop-in then answers
identity / score 0 where the reference calls exit(-1), super4pcs_test.cc:58-80); never a sanitizer report, never an
allocation beyond 256 MB for files of ~100 KB (ASAN_OPTIONS=max_allocation_size_mb aborts the run otherwise)."""
import os
import shutil
import struct
import subprocess
import zlib

import numpy as np
import pytest

from _dropin import write_png_like_opencv, write_png_with_filter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim")
EXE = os.path.join(SHIM, "test_parsers_asan")
ENV = dict(os.environ, ASAN_OPTIONS="max_allocation_size_mb=256:allocator_may_return_null=0:detect_leaks=1:abort_on_error=0",
           UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")

pytestmark = pytest.mark.skipif(not shutil.which("g++"), reason="no g++")


@pytest.fixture(scope="module")
def exe():
    r = subprocess.run(["make", "-C", SHIM, "asan"], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr + r.stdout):
        pytest.skip("this g++ has no sanitizer runtime")
    assert r.returncode == 0, r.stdout + r.stderr
    return EXE


def _png(path, img, depth=16):
    rows, cols = img.shape
    raw = b"".join(b"\x00" + (img[r].astype(">u2").tobytes() if depth == 16 else img[r].astype(np.uint8).tobytes()) for r in range(rows))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    z = zlib.compress(raw, 6)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", cols, rows, depth, 0, 0, 0, 0))
                + chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))


def _files(tmp):
    rng = np.random.default_rng(5)
    n = 1200
    xyz, nrm = rng.normal(0, 0.2, (n, 3)), rng.normal(0, 1, (n, 3))
    rgb = rng.integers(0, 255, (n, 3))
    head = ("ply\nformat {fmt} 1.0\ncomment PCL generated\nelement vertex %d\nproperty float x\nproperty float y\n"
            "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\n"
            "property float ny\nproperty float nz\nproperty float curvature\nelement camera 1\nproperty float view_px\n"
            "end_header\n") % n
    a = os.path.join(tmp, "ascii.ply")
    with open(a, "w") as f:
        f.write(head.format(fmt="ascii"))
        for p, c, q in zip(xyz, rgb, nrm):
            f.write("%.6g %.6g %.6g %d %d %d %.6g %.6g %.6g 0\n" % (*p, *c, *q))
        f.write("0\n")
    b = os.path.join(tmp, "binary.ply")
    with open(b, "wb") as f:
        f.write(head.format(fmt="binary_little_endian").encode())
        for p, c, q in zip(xyz, rgb, nrm):
            f.write(struct.pack("<3f3B4f", *p, *c, *q, 0.0))
        f.write(struct.pack("<f", 0.0))
    img = (rng.random((120, 160)) * 10000).astype(np.uint16)
    img[30:80, 40:100] = 10000                          # long matches and literals both
    p16, p8 = os.path.join(tmp, "prob16.png"), os.path.join(tmp, "prob8.png")
    _png(p16, img, 16)
    _png(p8, img >> 8, 8)
    # cv::imwrite's form (Sub on every row, byte runs only, several IDAT chunks) and every filter above every other: the
    # 16-bytes-at-a-time unfilter loops and the decoder's run / literal paths under the sanitizers
    pcv, pmix = os.path.join(tmp, "prob_cv.png"), os.path.join(tmp, "prob_mix.png")
    write_png_like_opencv(pcv, img)
    write_png_with_filter(img[:40, :77], [0, 1, 2, 3, 4, 0, 2, 4, 1, 3, 0, 3, 1, 4, 2, 0, 4, 3, 2, 1, 1, 0], pmix)
    return a, b, p16, p8, pcv, pmix


def _run(exe, *args):
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, env=ENV, timeout=900)
    assert r.returncode == 0, (args, r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def test_valid_files_are_read(exe, tmp_path):
    for f, kind in zip(_files(str(tmp_path)), ("ply", "ply", "png", "png", "png", "png")):
        assert "READ" in _run(exe, "file", kind, f), f


def test_seeded_mutations_of_ply_png_and_zlib_streams(exe, tmp_path):
    a, b, p16, p8, pcv, pmix = _files(str(tmp_path))
    scratch = os.path.join(str(tmp_path), "mutated.bin")
    total = 0
    for kind, f, n, seed in (("ply", a, 700, 1), ("ply", b, 700, 2), ("png", p16, 700, 3), ("png", p8, 400, 4),
                             ("png", pcv, 500, 7), ("png", pmix, 400, 8),
                             ("zlib", p16, 500, 5), ("zlib", p8, 300, 6), ("zlib", pcv, 400, 9)):
        out = _run(exe, kind, f, n, seed, scratch if kind != "zlib" else "-")
        assert out.strip().endswith("OK"), out
        read, refused = (int(out.split(w)[0].split()[-1]) for w in (" read", " refused"))
        assert read + refused == n and refused > 0          # the damage is real: some files are refused ...
        assert read > 0 or kind != "ply"                    # ... and some survive it and are read consistently (a PNG's
        #                                                     Adler-32 refuses nearly every damaged stream)
        total += n
    assert total >= 3000


def test_fast_adler32_equals_zlibs(exe):
    """shimio::adler32_of -- 32 bytes at a time where the CPU has AVX2, zlib's otherwise -- is what decides whether a
    decoded probability image is accepted: against zlib's adler32 over ~800 seeded buffers (all short lengths, lengths
    around the vector loop's block size, all-0xFF contents, unaligned starts), under the sanitizers."""
    r = subprocess.run([exe, "adler", "7"], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 differ" in r.stdout


def test_crafted_headers_are_refused_without_the_memory_they_ask_for(exe, tmp_path):
    a, b, p16 = _files(str(tmp_path))[:3]
    tmp = str(tmp_path)
    cases = []
    for name, src in (("ascii", a), ("binary", b)):
        data = open(src, "rb").read()
        for claim in (b"2000000000", b"2147483648", b"99999999999999", b"-7", b"x"):
            f = os.path.join(tmp, f"{name}_{claim.decode()}.ply")
            open(f, "wb").write(data.replace(b"element vertex 1200", b"element vertex " + claim))
            cases.append(("ply", f))
    png = open(p16, "rb").read()
    ihdr = png.index(b"IHDR")

    def with_ihdr(cols, rows, name):
        d = bytearray(png)
        d[ihdr + 4:ihdr + 12] = struct.pack(">II", cols, rows)
        f = os.path.join(tmp, name)
        open(f, "wb").write(bytes(d))
        cases.append(("png", f))
    with_ihdr(50000, 50000, "huge.png")           # 5 GB of pixels over 30 KB of data
    with_ihdr(16384, 16384, "big_but_allowed_size.png")   # within the size limit, beyond what the IDAT can inflate to
    with_ihdr(0x7FFFFFFF, 3, "wide.png")
    with_ihdr(160, 0, "no_rows.png")
    short = bytearray(png)                         # an IHDR chunk of 5 bytes followed by the rest of the file
    short[ihdr - 4:ihdr] = struct.pack(">I", 5)
    f = os.path.join(tmp, "short_ihdr.png")
    open(f, "wb").write(bytes(short[:ihdr + 4 + 5 + 4]) + png[ihdr + 4 + 13 + 4:])
    cases.append(("png", f))
    f = os.path.join(tmp, "truncated_ihdr.png")
    open(f, "wb").write(png[:ihdr + 4 + 6])
    cases.append(("png", f))
    for kind, f in cases:
        assert "REFUSED" in _run(exe, "file", kind, f), f
