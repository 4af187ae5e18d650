"""Reader side of the shape stage (tensoflow_amd/dataset.py): scene files -> ray table, on CPU."""
import json
import math
import os

import numpy as np
import pytest
import torch


def _write_scene(root, n=3, h=6, w=8, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "train"), exist_ok=True)
    frames, raw = [], []
    for k in range(n):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        img[0, 0, 3], img[0, 1, 3] = 0, 255
        Image.fromarray(img, "RGBA").save(os.path.join(root, "train", f"r_{k}.png"))
        a, b = 0.3 + k, 0.2 * k
        Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
        T = np.eye(4)
        T[:3, :3] = Rz @ Rx
        T[:3, 3] = T[:3, :3] @ np.array([0, 0, 4.0])
        frames.append({"file_path": f"./train/r_{k}", "transform_matrix": T.tolist()})
        raw.append(img)
    for split in ("train", "val"):
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as fp:
            json.dump({"camera_angle_x": 0.6911, "frames": frames if split == "train" else frames[:1]}, fp)
    return raw, frames


def test_database_and_ray_table(tmp_path):
    from tensoflow_amd.dataset import RayTable, TensoSDFSynDatabase, construct_ray_batch_nerf
    root = str(tmp_path / "scene")
    raw, frames = _write_scene(root)
    db = TensoSDFSynDatabase(root)
    assert len(db.get_img_ids()) == 4 and (db.H, db.W) == (6, 8)                       # train (3) + val (1)
    assert db.focal == pytest.approx(0.5 * 8 / math.tan(0.5 * 0.6911)) and db.K[0, 2] == 4 and db.K[1, 2] == 3
    a = raw[0].astype(np.float32) / 255.0
    want = ((a[..., :3] * a[..., 3:] + (1 - a[..., 3:])) * 255.0).astype(np.uint8)      # composited over white, truncated to 8 bits
    assert np.array_equal(db.get_image(0), want) and (db.get_image(0)[0, 0] == 255).all() and np.array_equal(db.get_image(0)[0, 1], raw[0][0, 1, :3])
    assert np.allclose(db.get_mask(0), a[..., 3]) and db.get_mask(0).shape == (6, 8)
    black = TensoSDFSynDatabase(root, splits=("train",), white_bg=False)
    assert (black.get_image(0)[0, 0] == 0).all() and len(black.img_ids) == 3
    T = np.array(frames[1]["transform_matrix"])
    assert np.allclose(db.get_pose(1)[:3, 3], 0.5 * T[:3, 3]) and np.allclose(db.get_pose(1)[:3, :3], T[:3, :3])
    assert np.allclose(db.pose_all[1], T)                                              # get_pose works on a copy

    info = db.imgs_info()
    batch, rn, h, w = construct_ray_batch_nerf(info)
    assert rn == 4 * 48 and (h, w) == (6, 8) and set(batch) == {"dirs", "rays_d", "rays_o", "radiis", "rays_cos", "rgbs", "human_poses", "masks"}
    # ray of image 2, pixel (row 4, col 5): through the pixel centre, camera looks down -z, y up
    k, r, c = 2, 4, 5
    row = k * 48 + r * 8 + c
    pose = torch.from_numpy(db.get_pose(k)).float()
    cam = torch.tensor([(c - 4 + 0.5) / db.focal, -(r - 3 + 0.5) / db.focal, -1.0], dtype=torch.float32)
    assert torch.allclose(batch["rays_d"][row], pose[:3, :3] @ cam, atol=1e-6) and torch.allclose(batch["rays_o"][row], pose[:3, 3])
    assert torch.allclose(batch["dirs"][row], torch.nn.functional.normalize(pose[:3, :3] @ cam, dim=0), atol=1e-6)
    assert float(batch["rays_cos"][row]) == pytest.approx(1 / float(cam.norm()), rel=1e-6)
    assert float(batch["radiis"][row]) == pytest.approx(math.sqrt((1 / db.focal) ** 2 / math.pi), rel=1e-5)       # pixel footprint disc
    assert torch.allclose(batch["rgbs"][row], torch.from_numpy(db.get_image(k)[r, c].astype(np.float32) / 255))
    assert float(batch["masks"][row]) == pytest.approx(float(db.get_mask(k)[r, c])) and batch["human_poses"].shape == (rn, 3, 4)
    test_batch, _, _, _ = construct_ray_batch_nerf(info, is_train=False)
    assert "masks" not in test_batch

    # the shuffled table: ranks of a 2-process run split every batch without overlap; one process sees the same rows in order
    one = RayTable(dict(batch), device="cpu")
    r0, r1 = RayTable(dict(batch), 0, 2, device="cpu"), RayTable(dict(batch), 1, 2, device="cpu")
    for _ in range(5):                                                                  # crosses a reshuffle (192 rows, 64 per batch)
        full, a0, a1 = one.next_batch(64), r0.next_batch(64), r1.next_batch(64)
        assert a0["rays_o"].shape[0] == a1["rays_o"].shape[0] == 32
        assert torch.equal(full["rgbs"][0::2], a0["rgbs"]) and torch.equal(full["rgbs"][1::2], a1["rgbs"])
    assert one.i == r0.i == r1.i
    seen = torch.cat([RayTable(dict(batch), device="cpu").next_batch(192 - 1)["rays_d"]])
    assert seen.shape[0] == 191


def test_tensoir_and_nerf_synthetic_layouts(tmp_path):
    """TensoIR (a directory per frame: the lego / armadillo / horse configs) and NeRF-synthetic (Blender) layouts,
    dataset/database.py:288-477, through parse_database_name."""
    from PIL import Image
    from tensoflow_amd.dataset import NeRFSynDatabase, TensoIRDatabase, construct_ray_batch_nerf, parse_database_name
    rng = np.random.default_rng(5)
    root = tmp_path / "data" / "lego"
    poses = {}
    for split, n in (("train", 3), ("val", 2), ("test", 2)):
        for k in reversed(range(n)):                                  # written out of order: the reader sorts by directory name
            d = root / f"{split}_{k:03d}"
            os.makedirs(d)
            T = np.eye(4)
            T[:3, 3] = [0.1 * k, -0.2, 4.0 + k]
            poses[(split, k)] = T
            json.dump({"cam_transform_mat": ",".join(repr(float(v)) for v in T.reshape(-1)), "cam_angle_x": 0.6911, "imh": 5, "imw": 7},
                      open(d / "metadata.json", "w"))
            img = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
            img[0, 0, 3] = 0
            Image.fromarray(img, "RGBA").save(d / "rgba_sunset_000.png")
            if split == "test":
                nrm = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
                nrm[0, 0, 3] = 0
                Image.fromarray(nrm, "RGBA").save(d / "normal.png")
                Image.fromarray(rng.integers(0, 256, (5, 7, 4), dtype=np.uint8), "RGBA").save(d / "albedo.png")
                poses[("test_nrm", k)] = nrm
    db = parse_database_name("tensoIR/lego", str(tmp_path / "data"), is_test=False, white_bg=True)
    assert isinstance(db, TensoIRDatabase) and len(db.img_ids) == 5 and (db.H, db.W) == (5.0, 7.0) and db.scale_factor == 0.5
    assert np.allclose(db.pose_all[1], poses[("train", 1)]) and np.allclose(db.pose_all[3], poses[("val", 0)])       # train_000..002, val_000..001
    assert np.allclose(db.get_pose(2)[:3, 3], 0.5 * poses[("train", 2)][:3, 3]) and (db.get_image(0)[0, 0] == 255).all()
    assert db.focal == pytest.approx(0.5 * 7 / math.tan(0.5 * 0.6911))
    batch, rn, h, w = construct_ray_batch_nerf(db.imgs_info())
    assert rn == 5 * 35 and (h, w) == (5, 7)
    te = parse_database_name("tensoIR/lego", str(tmp_path / "data"), is_test=True, white_bg=False)
    assert len(te.img_ids) == 2 and (te.get_image(0)[0, 0] == 0).all()
    n0 = poses[("test_nrm", 0)]
    assert np.allclose(te.get_normal(0)[0, 0], [0, 0, 1]) and np.allclose(te.get_normal(0)[2, 3], (n0[2, 3, :3] / 255 - 0.5) * 2 * (n0[2, 3, 3] / 255)
                                                                          + (1 - n0[2, 3, 3] / 255) * np.array([0, 0, 1]))
    assert te.get_albedo(1).shape == (5, 7, 3) and float(te.get_albedo(1).max()) <= 1.0
    # Blender layout: the TensoSDF reader with train + test splits and unscaled poses
    broot = str(tmp_path / "data" / "chair")
    raw, frames = _write_scene(broot)
    json.dump(json.load(open(os.path.join(broot, "transforms_val.json"))), open(os.path.join(broot, "transforms_test.json"), "w"))
    nb = parse_database_name("nerf/chair", str(tmp_path / "data"), is_test=False, white_bg=True)
    assert isinstance(nb, NeRFSynDatabase) and len(nb.img_ids) == 4 and nb.scale_factor == 1.0
    assert np.allclose(nb.get_pose(1)[:3, 3], np.array(frames[1]["transform_matrix"])[:3, 3])
    with pytest.raises(NotImplementedError, match="custom"):
        parse_database_name("custom/shoe", str(tmp_path / "data"))


def test_database_refuses_rgb_files(tmp_path):
    from PIL import Image
    from tensoflow_amd.dataset import TensoSDFSynDatabase
    root = str(tmp_path / "s")
    os.makedirs(os.path.join(root, "train"))
    Image.fromarray(np.zeros((4, 4, 3), np.uint8), "RGB").save(os.path.join(root, "train", "r_0.png"))
    for s in ("train", "val"):
        with open(os.path.join(root, f"transforms_{s}.json"), "w") as fp:
            json.dump({"camera_angle_x": 0.7, "frames": [{"file_path": "./train/r_0", "transform_matrix": np.eye(4).tolist()}]}, fp)
    with pytest.raises(ValueError):
        TensoSDFSynDatabase(root)


@pytest.mark.gpu
def test_shape_renderer_dataset_side(tmp_path):
    """ShapeRenderer(cfg, training=True) on a (tiny) scene in the TensoSDF synthetic layout: forward({'step'}) is a training
    iteration with the reference's loss keys, forward({'eval', 'index', 'step'}) renders a validation image."""
    from tensoflow_amd.network.shapeRenderer import ShapeRenderer
    root = str(tmp_path / "data" / "toy")
    _write_scene(root, n=5, h=16, w=16)
    cfg = dict(gridSize=[32, 32, 32], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False, device="cuda",
               nerfDataType=True, clip_sample_variance=False, apply_occ_loss=False, database_name="tensoSDF/toy",
               dataset_dir=str(tmp_path / "data"), apply_mask_loss=True, train_ray_num=256, test_ray_num=128, downsample_ratio=0.5)
    torch.manual_seed(0)
    r = ShapeRenderer(cfg, training=True).cuda()
    assert r.train_num == 5 and r.test_num == 1 and r.tbn == 5 * 256                    # val split holds 1 frame; default split: first image tests
    r.train()
    out = r({"step": 10})
    assert {"ray_rgb", "loss_rgb", "psnr", "loss_mask", "gradient_error", "acc", "loss_sparse", "loss_tv_sdf"} <= set(out)
    assert out["loss_rgb"].shape == (256,) and torch.isfinite(out["loss_rgb"]).all() and out["ray_rgb"].requires_grad
    (out["loss_rgb"].mean() + out["loss_mask"] + 0.1 * out["gradient_error"].mean()).backward()
    assert r.sdf_network.sdf_plane[0].grad is not None and float(r.sdf_network.sdf_plane[0].grad.abs().sum()) > 0
    r.eval()
    ev = r({"eval": True, "index": 0, "step": 10})
    assert ev["ray_rgb"].shape == (8, 8, 3) and ev["gt_rgb"].shape == (8, 8, 3) and ev["loss_rgb"].shape == (64,)      # downsampled by 2
    assert torch.isfinite(ev["ray_rgb"]).all() and ev["gt_mask"].shape == (8, 8, 1)
    # gt of the downsampled image = 2x2 box average of the file's composited pixels
    want = torch.from_numpy(r.database.get_image(r.test_ids[0]).astype(np.float32) / 255).reshape(8, 2, 8, 2, 3).mean((1, 3))
    assert torch.allclose(ev["gt_rgb"].cpu(), want, atol=1e-6)
    # the other Blender-convention layouts go through the same door (configs/shape/syn/lego.yaml: tensoIR/lego, split_manul)
    from PIL import Image
    rng = np.random.default_rng(2)
    for k in range(4):
        d = tmp_path / "data" / "lego" / f"train_{k:03d}"
        os.makedirs(d)
        T = np.eye(4)
        T[:3, 3] = [0.0, 0.0, 4.0]
        json.dump({"cam_transform_mat": ",".join(repr(float(v)) for v in T.reshape(-1)), "cam_angle_x": 0.6911, "imh": 16, "imw": 16}, open(d / "metadata.json", "w"))
        Image.fromarray(rng.integers(0, 256, (16, 16, 4), dtype=np.uint8), "RGBA").save(d / "rgba_sunset_000.png")
    r2 = ShapeRenderer({**cfg, "database_name": "tensoIR/lego", "split_manul": True, "split_borderline": 3}, training=True).cuda()
    assert r2.train_num == 3 and r2.test_num == 1
    r2.train()
    assert torch.isfinite(r2({"step": 3})["loss_rgb"]).all()
    with pytest.raises(NotImplementedError):
        ShapeRenderer({**cfg, "database_name": "custom/shoe"}, training=True)
    with pytest.raises(NotImplementedError):
        ShapeRenderer(cfg, training=False)({"step": 0})
