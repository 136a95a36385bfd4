"""The node side of the drop-in boundary as INPUT FILES (shared by tests/test_shim_gpu.py and bench.py's
drop_in row): three PLY clouds in pcl::io::savePLYFile's PointXYZRGBNormal layout, a 16-bit
probability PNG, a PPFMap.txt (PPE/data_layer/Objects.cpp:31-49) and camera intrinsics for one
synthetic segment -- what ObjectPoseCandidateSet.cpp:53-68 hands to getProbableTransformsSuper4PCS."""
import os

import numpy as np

from physimglobalpose_amd import synth


def write_ply(path, xyz, nrm, binary):
    """PointXYZRGBNormal layout of pcl::io::savePLYFile: x y z red green blue nx ny nz curvature."""
    n = len(xyz)
    hdr = ("ply\nformat %s 1.0\ncomment PCL generated\nelement vertex %d\nproperty float x\nproperty float y\n"
           "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\n"
           "property float ny\nproperty float nz\nproperty float curvature\nelement camera 1\n"
           "property float view_px\nend_header\n") % ("binary_little_endian" if binary else "ascii", n)
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if binary:
            rec = np.zeros(n, dtype=[("p", "<f4", 3), ("c", "u1", 3), ("n", "<f4", 3), ("k", "<f4")])
            rec["p"], rec["n"], rec["c"] = xyz, nrm, 128
            f.write(rec.tobytes())
            f.write(np.zeros(1, "<f4").tobytes())
        else:
            for p, q in zip(xyz, nrm):
                f.write(("%.9g %.9g %.9g 128 128 128 %.9g %.9g %.9g 0\n" % (*p, *q)).encode())
            f.write(b"0\n")


def ppf_map(P, N, trans_disc=5, rot_disc=10):
    """Model pair-feature table in the layout of PPFMap.txt (PPE/data_layer/Objects.cpp:31-49), with
    the features of Match4PCSBase::computePPF (base.cc:582-598) for every ordered pair."""
    P, N = P.astype(np.float32), N.astype(np.float32)
    n = len(P)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    m = i != j
    i, j = i[m], j[m]
    u = P[i] - P[j]

    def ang(a, b):
        return (np.arctan2(np.linalg.norm(np.cross(a, b), axis=1).astype(np.float32),
                           np.einsum("ij,ij->i", a, b).astype(np.float32)) * np.float32(180) / np.pi).astype(np.int64)

    def abin(v, d):
        lo = v - v % d
        return np.where(v - lo < lo + d - v, lo, lo + d)

    f = np.stack([abin((np.linalg.norm(u, axis=1).astype(np.float32) * np.float32(1000)).astype(np.int64), trans_disc),
                  abin(ang(N[i], u), rot_disc), abin(ang(N[j], u), rot_disc), abin(ang(N[i], N[j]), rot_disc)], 1)
    table = {}
    for key, a, b in zip(map(tuple, f.tolist()), i.tolist(), j.tolist()):
        table.setdefault(key, []).append((a, b))
    return table



def write_png_with_filter(img, ftype, path):
    """A greyscale PNG of `img` (uint8 / uint16) whose every row carries scanline filter `ftype` (a list gives one per
    row, cycled) -- cv::imwrite writes Sub on every row, PIL picks per row by heuristic and rarely leaves one alone."""
    import struct, zlib
    bpp = img.dtype.itemsize
    raw = img.astype(">u2" if bpp == 2 else np.uint8).tobytes()
    stride = img.shape[1] * bpp
    types = ftype if isinstance(ftype, (list, tuple)) else [ftype]
    body = bytearray()
    prev = bytes(stride)
    for r in range(img.shape[0]):
        cur = raw[r * stride:(r + 1) * stride]
        f = types[r % len(types)]
        line = bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0: pred = 0
            elif f == 1: pred = a
            elif f == 2: pred = b
            elif f == 3: pred = (a + b) >> 1
            else:
                q = a + b - c
                pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            line[i] = (cur[i] - pred) & 0xFF
        body += bytes([f]) + line
        prev = cur
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", img.shape[1], img.shape[0], 8 * bpp, 0, 0, 0, 0))
                 + chunk(b"IDAT", zlib.compress(bytes(body), 6)) + chunk(b"IEND", b""))


def write_png_like_opencv(path, img):
    """16-bit greyscale PNG in the form cv::imwrite gives a CV_16UC1 image with its defaults (OpenCV is not in this
    image; its PNG encoder documents strategy IMWRITE_PNG_STRATEGY_RLE and a low compression level as the defaults
    and applies filter Sub to every row): what the reference node writes and base.cc:317 reads back."""
    import struct, zlib
    be = img.astype(">u2").view(np.uint8).reshape(img.shape[0], -1)
    sub = be.copy()
    sub[:, 2:] = be[:, 2:] - be[:, :-2]
    body = np.concatenate([np.ones((img.shape[0], 1), np.uint8), sub], axis=1).tobytes()
    z = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
    data = z.compress(body) + z.flush()
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", img.shape[1], img.shape[0], 16, 0, 0, 0, 0)))
        for k in range(0, len(data), 8192):                      # libpng's default IDAT size
            f.write(chunk(b"IDAT", data[k:k + 8192]))
        f.write(chunk(b"IEND", b""))


def make_dropin_case(tmp, binary=False, n_scene=8000, n_model=1500, n_search=800, config_id=91, png="pil"):
    """Writes the files under `tmp`; returns (argv for shim/test_shim, info dict with the workload).
    png="opencv": the probability image as the node's network writes it -- a dense map (class probability over the whole
    frame, the same values as the sparse image at every pixel a segment point falls on, so the match is the same) in
    cv::imwrite's encoding; "opencv_noisy": the same with noise in every pixel (the decoder's worst case); "pil": the
    points' pixels only, PIL's encoding (the r1-r4 form)."""
    from PIL import Image
    w = synth.make_workload(n_scene, n_model, 4, config_id=config_id, n_search=n_search)
    # a segment as the node produces it: mostly the object, some clutter around it
    rng = np.random.default_rng(0)
    obj = np.flatnonzero(w.P_w == 1.0)
    clutter = rng.choice(np.flatnonzero(w.P_w < 1.0), min(1500, int((w.P_w < 1.0).sum())), replace=False)
    keep = np.sort(np.concatenate([obj, clutter]))
    w.P_xyz, w.P_nrm, w.P_w = w.P_xyz[keep], w.P_nrm[keep], w.P_w[keep]
    # world-frame clouds, as the node hands them over
    P = w.P_xyz + w.centroid_P
    Qv = w.Q_xyz + w.centroid_Q
    Qs = w.Qs_xyz + w.centroid_Q
    seg, val, search = (os.path.join(str(tmp), n) for n in ("pclSegment.ply", "pclModel.ply", "pclModelSampled.ply"))
    write_ply(seg, P, w.P_nrm, binary)
    write_ply(val, Qv, w.Q_nrm, binary)
    write_ply(search, Qs, w.Qs_nrm, binary)
    fx = fy = 600.0
    cx, cy = 320.0, 240.0
    img = np.zeros((480, 640), np.uint16)
    col = (fx * P[:, 0] / P[:, 2] + cx).astype(int)
    row = (fy * P[:, 1] / P[:, 2] + cy).astype(int)
    ok = (row >= 0) & (row < 480) & (col >= 0) & (col < 640)
    order = np.argsort(w.P_w[ok])                       # object pixels (w = 1) are written last
    img[row[ok][order], col[ok][order]] = np.round(w.P_w[ok][order] * 10000).astype(np.uint16)
    png_path = os.path.join(str(tmp), "prob.png")
    if png in ("opencv", "opencv_noisy"):
        yy, xx = np.mgrid[0:480, 0:640]
        cr, cc = (row[ok].mean(), col[ok].mean()) if ok.any() else (240.0, 320.0)
        dist = np.sqrt((yy - cr) ** 2 + (xx - cc) ** 2)
        noise = np.random.default_rng(3).standard_normal(dist.shape)
        if png == "opencv":
            # a confident network: 1 inside the object's blob, 0 outside, an uncertain band between them
            prob = 1.0 / (1.0 + np.exp((dist - 110.0) / 6.0))
            dense = 10000.0 * prob + 600.0 * prob * (1.0 - prob) * noise
        else:
            # no flat region anywhere (the low byte of every pixel is noise): the encoder finds nothing to match and the
            # decoder's time is all literals -- the slowest image of this size there is
            dense = 10000.0 * np.exp(-dist ** 2 / (2 * 90.0 ** 2)) + 40.0 * noise
        dense = np.round(dense).clip(0, 10000).astype(np.uint16)
        hit = np.zeros(img.shape, bool)
        hit[row[ok], col[ok]] = True
        img = np.where(hit, img, dense)
        write_png_like_opencv(png_path, img)
    else:
        Image.fromarray(img).save(png_path)
    png = png_path
    table = ppf_map(Qs, w.Qs_nrm)
    ppf = os.path.join(str(tmp), "PPFMap.txt")
    with open(ppf, "w") as f:
        for key, pairs in table.items():
            f.write("%d %d %d %d %d %s\n" % (*key, len(pairs), " ".join("%d %d" % p for p in pairs)))
    argv = [seg, val, search, png, ppf, str(fx), str(fy), str(cx), str(cy)]
    info = {"n_segment": int(len(P)), "n_model_validation": int(len(Qv)), "n_model_search": int(len(Qs)),
            "ppf_keys": len(table)}
    return argv, {"info": info, "w": w, "P": P, "Qv": Qv, "Qs": Qs, "table": table}
