"""Readers for the HDR image formats on either side of the hot path: OpenEXR scan-line files
(`*_diffColor.exr` of the TensoSDF test split, dataset/database.py:530-535; lat-long environment maps of the relighting scripts)
and Radiance `.hdr` (RGBE) lat-long maps (EnvLight.load, network/light.py:39-49, which goes through imageio).

No OpenEXR / imageio / cv2 in this image, so both decoders are written out from the published file formats:

* OpenEXR 2 single-part scan-line files, compression NONE / ZIPS / ZIP (what Blender writes by default), channels HALF / FLOAT /
  UINT, any channel set (returned in R, G, B, A order when those names exist, alphabetical otherwise).  Tiled, multi-part, deep and
  the lossy codecs (PIZ, PXR24, B44, DWA) raise NotImplementedError.
* Radiance RGBE: header up to the blank line, `-Y h +X w` resolution line, new-style run-length encoded or flat scan lines.

`write_exr` (NONE / ZIP, HALF / FLOAT) exists for the material export path and for the round-trip tests.  PARITY UNPINNED: there is
no second decoder in the image to compare with and the reference holds no EXR / HDR fixture; the tests check round trips and a
hand-assembled file.
"""
import struct
import zlib

import numpy as np

_EXR_MAGIC = 20000630
_PIX = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}


def _cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _zip_reconstruct(raw):
    """Inverse of OpenEXR's ZIP pre-processing: byte delta predictor, then de-interleave of the two half streams."""
    t = np.frombuffer(raw, dtype=np.uint8).astype(np.int64)
    t[1:] -= 128
    t = (np.cumsum(t) & 0xFF).astype(np.uint8)
    n = t.size
    out = np.empty(n, np.uint8)
    half = (n + 1) // 2
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _zip_prepare(raw):
    a = np.frombuffer(raw, dtype=np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int64)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) & 0xFF
    return d.astype(np.uint8).tobytes()


def read_exr(path):
    """-> float32 array [H, W, C] (C in R, G, B, A order when present) and the list of channel names in that order."""
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<iI", buf, 0)
    if magic != _EXR_MAGIC:
        raise ValueError(f"{path}: not an OpenEXR file")
    if version & 0x1A00:        # tiled (0x200), deep (0x800), multi-part (0x1000)
        raise NotImplementedError(f"{path}: tiled / deep / multi-part OpenEXR files are not read")
    pos, attrs = 8, {}
    while buf[pos] != 0:
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        (size,) = struct.unpack_from("<i", buf, pos)
        attrs[name] = (typ, buf[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    chans, c, cp = [], attrs["channels"][1], 0
    while c[cp] != 0:
        name, cp = _cstr(c, cp)
        ptype, _lin, xs, ys = struct.unpack_from("<iB3xii", c, cp)
        cp += 16
        if xs != 1 or ys != 1:
            raise NotImplementedError(f"{path}: sub-sampled channel {name}")
        chans.append((name, _PIX[ptype]))
    comp = attrs["compression"][1][0]
    if comp not in (0, 2, 3):
        raise NotImplementedError(f"{path}: OpenEXR compression {comp} (only NONE / ZIPS / ZIP are read)")
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = xmax - xmin + 1, ymax - ymin + 1
    lines = 16 if comp == 3 else 1
    n_chunks = (H + lines - 1) // lines
    offsets = struct.unpack_from(f"<{n_chunks}Q", buf, pos)
    row_bytes = sum(dt.itemsize for _, dt in chans) * W
    planes = {name: np.empty((H, W), np.float32) for name, _ in chans}
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        nl = min(lines, ymax + 1 - y)
        want = row_bytes * nl
        if comp != 0 and size < want:
            data = _zip_reconstruct(zlib.decompress(data))
        if len(data) != want:
            raise ValueError(f"{path}: scan-line block at y={y} has {len(data)} bytes, expected {want}")
        p = 0
        for r in range(nl):
            for name, dt in chans:          # within a scan line: channels in file (alphabetical) order, each a run of W samples
                planes[name][y - ymin + r] = np.frombuffer(data, dtype=dt, count=W, offset=p).astype(np.float32)
                p += dt.itemsize * W
    names = [n for n, _ in chans]
    order = [n for n in ("R", "G", "B", "A") if n in names] + [n for n in names if n not in ("R", "G", "B", "A")]
    return np.stack([planes[n] for n in order], -1), order


def write_exr(path, img, channels=None, half=False, compress=True):
    """float array [H, W, C] -> single-part scan-line OpenEXR (ZIP or uncompressed; HALF or FLOAT samples)."""
    img = np.asarray(img, dtype=np.float32)
    H, W, C = img.shape
    channels = list(channels) if channels else list("RGBA"[:C]) if C <= 4 else [f"C{i}" for i in range(C)]
    order = sorted(range(C), key=lambda i: channels[i])      # the format wants the channel list sorted by name
    dt, ptype = (np.dtype("<f2"), 1) if half else (np.dtype("<f4"), 2)
    chlist = b"".join(channels[i].encode("latin-1") + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for i in order) + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val

    head = struct.pack("<iI", _EXR_MAGIC, 2)
    head += attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([3 if compress else 0]))
    head += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    head += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0))
    head += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = 16 if compress else 1
    chunks = []
    for y in range(0, H, lines):
        raw = b"".join(img[r, :, i].astype(dt).tobytes() for r in range(y, min(y + lines, H)) for i in order)
        data = raw
        if compress:
            z = zlib.compress(_zip_prepare(raw))
            if len(z) < len(raw):
                data = z
        chunks.append(struct.pack("<ii", y, len(data)) + data)
    table_at = len(head)
    off, offsets = table_at + 8 * len(chunks), []
    for ch in chunks:
        offsets.append(off)
        off += len(ch)
    with open(path, "wb") as fp:
        fp.write(head + struct.pack(f"<{len(chunks)}Q", *offsets) + b"".join(chunks))


def read_hdr(path):
    """Radiance RGBE picture -> float32 [H, W, 3]."""
    buf = open(path, "rb").read()
    if not (buf.startswith(b"#?RADIANCE") or buf.startswith(b"#?RGBE")):
        raise ValueError(f"{path}: not a Radiance HDR file")
    end = buf.index(b"\n\n")
    if b"FORMAT=32-bit_rle_rgbe" not in buf[:end]:
        raise NotImplementedError(f"{path}: only FORMAT=32-bit_rle_rgbe is read")
    eol = buf.index(b"\n", end + 2)
    res = buf[end + 2:eol].split()
    if len(res) != 4 or res[0] != b"-Y" or res[2] != b"+X":
        raise NotImplementedError(f"{path}: resolution line {buf[end + 2:eol]!r} (only '-Y h +X w' is read)")
    H, W = int(res[1]), int(res[3])
    pos = eol + 1
    rgbe = np.empty((H, W, 4), np.uint8)
    for y in range(H):
        if 8 <= W < 32768 and buf[pos] == 2 and buf[pos + 1] == 2 and ((buf[pos + 2] << 8) | buf[pos + 3]) == W:
            pos += 4                                       # new-style RLE: the four components run-length coded one after the other
            for c in range(4):
                x = 0
                while x < W:
                    n = buf[pos]
                    if n > 128:
                        rgbe[y, x:x + n - 128, c] = buf[pos + 1]
                        x += n - 128
                        pos += 2
                    else:
                        rgbe[y, x:x + n, c] = np.frombuffer(buf, np.uint8, n, pos + 1)
                        x += n
                        pos += 1 + n
        else:
            rgbe[y] = np.frombuffer(buf, np.uint8, 4 * W, pos).reshape(W, 4)
            pos += 4 * W
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)       # 2^(e-128) / 256
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


def imread_float(path):
    """An image as float32 [H, W, C]: .exr / .hdr (linear), .npy, or an 8-bit file through PIL scaled to [0,1] (imageio.imread +
    the `/ 255` of EnvLight.load, network/light.py:41-43)."""
    p = path.lower()
    if p.endswith(".exr"):
        return read_exr(path)[0]
    if p.endswith(".hdr"):
        return read_hdr(path)
    if p.endswith(".npy"):
        return np.load(path).astype(np.float32)
    from PIL import Image
    return np.asarray(Image.open(path)).astype(np.float32) / 255.0
