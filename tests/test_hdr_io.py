"""HDR readers and the data layouts around the hot path (tensoflow_amd/hdr_io.py, dataset.ORBDatabase, EnvLight.load's lat-long
resampling), on CPU.  No second EXR / HDR decoder exists in the image: round trips, a hand-assembled file and closed-form maps."""
import json
import math
import os
import struct
import zlib

import numpy as np
import pytest
import torch


def test_exr_round_trip_and_hand_built_file(tmp_path):
    from tensoflow_amd.hdr_io import read_exr, write_exr
    rng = np.random.default_rng(0)
    img = rng.standard_normal((37, 29, 4)).astype(np.float32) * 3          # 37 rows: three ZIP blocks, the last one partial
    for half, compress in ((False, False), (False, True), (True, True)):
        p = str(tmp_path / f"a_{half}_{compress}.exr")
        write_exr(p, img, half=half, compress=compress)
        got, names = read_exr(p)
        assert names == ["R", "G", "B", "A"] and got.shape == img.shape
        want = img.astype(np.float16).astype(np.float32) if half else img
        assert np.array_equal(got, want)
    # a file assembled by hand from the format description: 2 x 3 pixels, channels B (half) and Y (float), ZIPS (one line per block),
    # data window starting at (5, 7)
    B = np.array([[0.5, 1.0, 2.0], [-1.0, 0.25, 8.0]], np.float16)
    Y = np.array([[1e-3, 2.0, 3.5], [4.0, 5.0, 6e3]], np.float32)
    ch = b"B\0" + struct.pack("<iB3xii", 1, 0, 1, 1) + b"Y\0" + struct.pack("<iB3xii", 2, 0, 1, 1) + b"\0"
    attr = lambda n, t, v: n + b"\0" + t + b"\0" + struct.pack("<i", len(v)) + v
    head = struct.pack("<iI", 20000630, 2) + attr(b"channels", b"chlist", ch) + attr(b"compression", b"compression", b"\x02")
    head += attr(b"dataWindow", b"box2i", struct.pack("<4i", 5, 7, 7, 8)) + attr(b"lineOrder", b"lineOrder", b"\0") + b"\0"
    blocks = []
    for r in range(2):
        raw = B[r].tobytes() + Y[r].tobytes()
        a = np.frombuffer(raw, np.uint8)
        t = np.concatenate([a[0::2], a[1::2]]).astype(np.int64)
        d = t.copy(); d[1:] = (t[1:] - t[:-1] + 384) & 255
        z = zlib.compress(d.astype(np.uint8).tobytes())
        data = z if len(z) < len(raw) else raw
        blocks.append(struct.pack("<ii", 7 + r, len(data)) + data)
    off0 = len(head) + 16
    table = struct.pack("<2Q", off0, off0 + len(blocks[0]))
    p = str(tmp_path / "hand.exr")
    open(p, "wb").write(head + table + b"".join(blocks))
    got, names = read_exr(p)
    assert names == ["B", "Y"] and np.array_equal(got[..., 0], B.astype(np.float32)) and np.array_equal(got[..., 1], Y)
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad.exr"), "wb").write(b"not an exr file at all")
        read_exr(str(tmp_path / "bad.exr"))


def test_radiance_hdr_flat_and_rle(tmp_path):
    from tensoflow_amd.hdr_io import read_hdr
    rng = np.random.default_rng(1)
    H, W = 5, 40
    rgbe = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    rgbe[..., 3] = rng.integers(120, 136, (H, W))
    rgbe[0, :10] = rgbe[0, 0]                                   # a run for the RLE coder
    rgbe[1, 3, 3] = 0                                           # exponent 0 = black
    want = rgbe[..., :3].astype(np.float32) * np.where(rgbe[..., 3:] > 0, np.ldexp(1.0, rgbe[..., 3:].astype(np.int32) - 136), 0.0).astype(np.float32)
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (H, W)
    open(str(tmp_path / "flat.hdr"), "wb").write(head + rgbe.tobytes())
    assert np.array_equal(read_hdr(str(tmp_path / "flat.hdr")), want)
    body = b""
    for y in range(H):                                          # new-style RLE, runs of equal bytes coded as runs, the rest as literals
        body += bytes([2, 2, W >> 8, W & 255])
        for c in range(4):
            row, x = rgbe[y, :, c], 0
            while x < W:
                run = 1
                while x + run < W and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 4:
                    body += bytes([128 + run, int(row[x])]); x += run
                else:
                    n = min(W - x, 7)
                    body += bytes([n]) + row[x:x + n].tobytes(); x += n
    open(str(tmp_path / "rle.hdr"), "wb").write(head + body)
    assert np.array_equal(read_hdr(str(tmp_path / "rle.hdr")), want)


def test_latlong_to_cubemap_and_envlight_load(tmp_path):
    from tensoflow_amd.network.light import EnvLight, _texel_centre_dirs, latlong_to_cubemap
    H, W, res = 256, 512, 16
    # a map that is a smooth function of the direction its texel looks at: u = atan2(x, -z) / 2pi + 1/2, v = acos(y) / pi
    v, u = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
    th, ph = v * np.pi, (u - 0.5) * 2 * np.pi
    d = np.stack([np.sin(th) * np.sin(ph), np.cos(th), -np.sin(th) * np.cos(ph)], -1)
    img = (0.5 + 0.25 * d + 0.1 * d[..., ::-1] ** 2).astype(np.float32)
    cube = latlong_to_cubemap(img, res, "cpu")
    dirs = _texel_centre_dirs(res, "cpu").numpy().astype(np.float64)
    want = 0.5 + 0.25 * dirs + 0.1 * dirs[..., ::-1] ** 2
    assert cube.shape == (6, res, res, 3) and float(np.abs(cube.reshape(-1, 3).numpy() - want).max()) < 2e-3     # bilinear error at 256 x 512
    assert torch.allclose(latlong_to_cubemap(np.full((8, 16, 3), 0.7, np.float32), 4, "cpu"), torch.full((6, 4, 4, 3), 0.7))
    # the seam u = 0 | 1 (direction -z, x -> 0) wraps: a map constant along u stays exact across it
    ramp = np.repeat(np.linspace(0, 1, H, dtype=np.float32)[:, None, None], W, 1).repeat(3, 2)
    c2 = latlong_to_cubemap(ramp, res, "cpu").reshape(-1, 3).numpy()
    tv = np.arccos(np.clip(dirs[:, 1], -1, 1)) / np.pi
    assert float(np.abs(c2[:, 0] - np.clip((tv * H - 0.5) / (H - 1), 0, 1)).max()) < 1e-5
    from tensoflow_amd.hdr_io import write_exr
    write_exr(str(tmp_path / "env.exr"), img)
    env = EnvLight(path=str(tmp_path / "env.exr"), device="cpu", scale=2.0, max_res=res, min_res=4)
    assert torch.allclose(env.base.data, cube * 2.0, atol=1e-6)


def test_orb_database_and_test_split_extras(tmp_path):
    from PIL import Image
    from tensoflow_amd.dataset import ORBDatabase, TensoSDFSynDatabase, parse_database_name
    from tensoflow_amd.hdr_io import write_exr
    rng = np.random.default_rng(2)
    root = str(tmp_path / "data" / "teapot")
    frames = []
    for split in ("train", "test"):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        os.makedirs(os.path.join(root, split + "_mask"), exist_ok=True)
    imgs, masks = [], []
    for k in range(2):
        img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
        m = rng.integers(0, 2, (5, 7), dtype=np.uint8) * 255
        Image.fromarray(img, "RGB").save(os.path.join(root, "train", f"{k:04d}.png"))
        Image.fromarray(m, "L").save(os.path.join(root, "train_mask", f"{k:04d}.png"))
        T = np.eye(4); T[:3, 3] = [0.1 * k, 0.2, 3.0]
        frames.append({"file_path": f"train/{k:04d}", "transform_matrix": T.tolist()})
        imgs.append(img); masks.append(m)
    json.dump({"camera_angle_x": 0.5, "frames": frames}, open(os.path.join(root, "transforms_train.json"), "w"))
    db = parse_database_name("orb/teapot", str(tmp_path / "data"), is_test=False, white_bg=True)
    assert isinstance(db, ORBDatabase) and db.scale_factor == 1.0 and (db.H, db.W) == (5, 7) and len(db.img_ids) == 2
    a, m = imgs[1].astype(np.float32) / 255.0, (masks[1].astype(np.float32) / 255.0)[..., None]
    assert np.array_equal(db.get_image(1), ((a * m + (1 - m)) * 255.0).astype(np.uint8)) and np.array_equal(db.get_mask(1), m[..., 0])
    assert np.allclose(db.get_pose(1)[:3, 3], [0.1, 0.2, 3.0])                       # no rescaling
    assert db.focal == pytest.approx(0.5 * 7 / math.tan(0.25))
    info = db.imgs_info()
    assert info["imgs"].shape == (2, 3, 5, 7) and info["masks"].shape == (2, 1, 5, 7)
    with pytest.raises(NotImplementedError):
        parse_database_name("real/bear", str(tmp_path / "data"))
    # TensoSDF test split: normals and diffuse colour
    r2 = str(tmp_path / "data" / "compressor")
    os.makedirs(os.path.join(r2, "test"), exist_ok=True)
    rgba = rng.integers(0, 256, (4, 6, 4), dtype=np.uint8); rgba[..., 3] = rng.integers(0, 2, (4, 6)) * 255
    nrm = rng.integers(0, 256, (4, 6, 3), dtype=np.uint8)
    dc = rng.random((4, 6, 4)).astype(np.float32)
    Image.fromarray(rgba, "RGBA").save(os.path.join(r2, "test", "r_0.png"))
    Image.fromarray(nrm, "RGB").save(os.path.join(r2, "test", "r_0_normal.png"))
    write_exr(os.path.join(r2, "test", "r_0_diffColor.exr"), dc, half=True)
    json.dump({"camera_angle_x": 0.6, "frames": [{"file_path": "./test/r_0", "transform_matrix": np.eye(4).tolist()}]},
              open(os.path.join(r2, "transforms_test.json"), "w"))
    t = parse_database_name("tensoSDF/compressor", str(tmp_path / "data"), is_test=True, white_bg=True)
    assert isinstance(t, TensoSDFSynDatabase) and len(t.img_ids) == 1
    al = (rgba[..., 3:].astype(np.float32) / 255.0)
    want_n = ((nrm / 255 - 0.5) * 2.0) * al + (1 - al) * np.array([0, 0, 1])
    assert np.allclose(t.get_normal(0), want_n)
    d16 = dc.astype(np.float16).astype(np.float32)
    assert np.allclose(t.get_albedo(0), d16[..., :3] * d16[..., 3:])
