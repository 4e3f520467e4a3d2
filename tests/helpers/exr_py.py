"""An EXR scan-line reader that shares nothing with host/imageio.cpp: struct + zlib + numpy, written from the OpenEXR file-layout description
(magic, version, attribute list, line-offset table, chunks; ZIP / ZIPS chunks = zlib, then the byte predictor, then the two-halves interleave).
Test infrastructure: the independent pin of `lmc_image_read`."""
import struct
import zlib

import numpy as np


def _cstr(b, p):
    e = b.index(b"\0", p)
    return b[p:e].decode("latin-1"), e + 1


def read_exr(path):
    """-> (channel names in file order, {name: float32 [H, W]})"""
    b = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", b, 0)
    assert magic == 20000630 and not (version & 0x1A00), "scan-line single-part files only"
    p = 8
    attrs = {}
    while True:
        name, p = _cstr(b, p)
        if not name:
            break
        typ, p = _cstr(b, p)
        (size,) = struct.unpack_from("<i", b, p)
        p += 4
        attrs[name] = (typ, b[p : p + size])
        p += size
    chans = []  # (name, pixel type)
    c = attrs["channels"][1]
    q = 0
    while c[q] != 0:
        name, q = _cstr(c, q)
        ptype, _plinear, xs, ys = struct.unpack_from("<iIii", c, q)
        q += 16
        assert xs == 1 and ys == 1
        chans.append((name, ptype))
    comp = attrs["compression"][1][0]
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    lines_per_block = {0: 1, 2: 1, 3: 16}[comp]
    nblocks = (H + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from("<%dQ" % nblocks, b, p)
    dt = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}
    line_bytes = sum(W * dt[t].itemsize for _, t in chans)
    out = {name: np.zeros((H, W), np.float32) for name, _ in chans}
    for off in offsets:
        y, dsize = struct.unpack_from("<ii", b, off)
        y -= y0
        nl = min(lines_per_block, H - y)
        usize = line_bytes * nl
        data = b[off + 8 : off + 8 + dsize]
        if comp != 0 and dsize < usize:
            t = np.frombuffer(zlib.decompress(data), np.uint8)
            assert t.size == usize
            # predictor: t[i] = t[i-1] + t[i] - 128 (mod 256) == running sum of (t - 128) with t[0] kept
            d = t.astype(np.int64)
            d[1:] -= 128
            t = (np.cumsum(d) & 0xFF).astype(np.uint8)
            half = (usize + 1) // 2
            raw = np.empty(usize, np.uint8)
            raw[0::2] = t[:half]
            raw[1::2] = t[half:]
            raw = raw.tobytes()
        else:
            raw = data[:usize]
        q = 0
        for l in range(nl):
            for name, t in chans:  # inside a scan line the channels follow each other in file (alphabetical) order
                n = W * dt[t].itemsize
                out[name][y + l] = np.frombuffer(raw, dt[t], W, q).astype(np.float32)
                q += n
    return [n for n, _ in chans], out


def read_exr_rgb(path):
    names, ch = read_exr(path)
    return np.stack([ch["R"], ch["G"], ch["B"]], axis=-1)
