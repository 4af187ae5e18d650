"""The two stages chained as a user of the reference would run them (run_training.py shape -> extract_mesh.py -> run_training.py
material -> eval_mat.py), on a synthetic object: every hand-over goes through the files the reference uses (TrainerInv checkpoint,
PLY mesh, .npy materials)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_shape_stage_to_material_stage(tmp_path):
    from tensoflow_amd.mesh import extract_mesh
    from tensoflow_amd.network.materialRenderer import MaterialRenderer
    from tensoflow_amd.network.shapeRenderer import ShapeRenderer
    from tensoflow_amd.synth import pinhole_rays, random_sdf_state
    from tensoflow_amd.trainer import MaterialTrainer, ShapeTrainer
    if not torch.cuda.is_available():
        pytest.fail("GPU test collected but no GPU is visible")
    dev = torch.device("cuda:0")
    torch.manual_seed(6033)
    R = 48

    def make(grid, max_levels=3):
        r = ShapeRenderer(dict(gridSize=list(grid), max_levels=max_levels, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False,
                               device="cuda", nerfDataType=True, clip_sample_variance=False, apply_occ_loss=False, blend_ratio=0.2),
                          training=False).to(dev)
        return r

    # ---- shape stage: a few optimisation steps from an object-sized initial surface, then the checkpoint
    st = ShapeTrainer(make, dict(total_step=50, N_voxel_init=R ** 3, N_voxel_final=R ** 3))
    sd = random_sdf_state(seed=1, R=st.net.gridSize.tolist()[0])
    sd["sdf_mat.2.bias"][0] -= 0.35
    st.net.load_state_dict({"sdf_network." + k: v for k, v in sd.items()}, strict=False)
    o, d, radii, cos = [torch.from_numpy(a).to(dev) for a in pinhole_rays(512, seed=2, h=64, w=64, focal=90.0)]
    batch = {"rays_o": o, "rays_d": d, "dirs": d, "radiis": radii, "rays_cos": cos, "rgbs": torch.rand(512, 3, device=dev),
             "masks": torch.ones(512, device=dev)}
    for _ in range(3):
        assert torch.isfinite(st.train_step(batch)["loss"])
    ckpt = str(tmp_path / "shape.pth")
    st.save(ckpt)
    # ---- mesh hand-over
    ply = str(tmp_path / "shape.ply")
    v, f = extract_mesh(st.net, 64, ply)
    assert f.shape[0] > 500
    # ---- material stage on the files
    shader_cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=64, nis_diffuse_sample_num=32,
                      nis_specular_sample_num=32, outer_light_version="envlight")
    mat = MaterialRenderer({"mesh": ply, "geo_model_path": ckpt, "shader_cfg": shader_cfg}, training=False, nvs=True)
    assert mat.gridSize.tolist() == st.net.gridSize.tolist()
    o2, d2, _, _ = [torch.from_numpy(a).to(dev) for a in pinhole_rays(2048, seed=3, h=64, w=64, focal=90.0)]
    pts, nrm, depth, hit = mat.trace_sdf_with_mesh(o2, d2)
    hit = hit.reshape(-1)
    assert int(hit.sum()) > 50
    # refined points sit near the zero set (a NeuS-weighted mean depth at inv_s = e^3, not a root), normals face the camera
    with torch.no_grad():
        s = st.net.sdf_network.sdf(pts[hit].contiguous(), None)[:, 0]
    assert float(s.abs().max()) < 4e-2 and float((nrm[hit] * d2[hit]).sum(-1).max()) < 0.2
    mt = MaterialTrainer(mat.shader_network, dict(total_step=50, nis_loss_iter=1))
    P, V, Nn = pts[hit].contiguous(), (-d2[hit]).contiguous(), nrm[hit].contiguous()
    target = torch.sigmoid(3.0 * P)
    losses = [float(mt.train_step(P, V, Nn, target)["loss_rgb"]) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    # ---- the module's own dataset side: a scene on disk -> traced surface-point table -> forward({'step'}) / forward({'eval'})
    from test_dataset import _write_scene
    _write_scene(str(tmp_path / "data" / "toy"), n=4, h=24, w=24)
    md = MaterialRenderer({"mesh": ply, "geo_model_path": ckpt, "shader_cfg": shader_cfg, "nerfDataType": True, "database_name": "tensoSDF/toy",
                           "dataset_dir": str(tmp_path / "data"), "train_ray_num": 128}, training=True)
    assert md.train_num == 4 and 0.02 < md.ray_mask_ratio < 0.98 and md.tbn == md.train_batch["inters"].shape[0]
    assert {"rays_o", "rays_d", "rgb", "inters", "normals", "depth", "human_poses"} <= set(md.train_batch)
    with torch.no_grad():
        s_tab = st.net.sdf_network.sdf(md.train_batch["inters"][:512].to(dev).contiguous(), None)[:, 0]
    assert float(s_tab.abs().max()) < 4e-2                                 # table rows are surface points of the trained SDF
    md.train()
    out = md({"step": 600})
    assert {"rgb_pr", "rgb_gt", "loss_rgb", "psnr", "loss_mat_reg", "loss_diffuse_light", "loss_nis"} <= set(out), sorted(out)
    (out["loss_rgb"].mean() + out["loss_nis"].mean() + out["loss_mat_reg"].mean()).backward()
    assert md.shader_network.mat_plane[0].grad is not None and torch.isfinite(out["loss_rgb"]).all()
    md.eval()
    ev = md({"eval": True, "index": 0})
    assert ev["rgb_pr"].shape == (24, 24, 3) and ev["rgb_gt"].shape == (24, 24, 3) and torch.isfinite(ev["rgb_pr"]).all()
    # the TensoIR layout (configs/mat/syn/{lego,armadillo,horse}.yaml) keeps H / W as FLOATS, like the reference's database class:
    # test_step takes the frame size from the image (materialRenderer.py:583), not from them (advisor, round 3)
    import json
    import os
    from PIL import Image
    rng = np.random.default_rng(3)
    for k in range(3):
        dd = tmp_path / "data" / "lego" / f"train_{k:03d}"
        os.makedirs(dd)
        T = np.eye(4)
        T[:3, 3] = [0.0, 0.0, 4.0]
        json.dump({"cam_transform_mat": ",".join(repr(float(x)) for x in T.reshape(-1)), "cam_angle_x": 0.6911, "imh": 12, "imw": 12}, open(dd / "metadata.json", "w"))
        Image.fromarray(rng.integers(0, 256, (12, 12, 4), dtype=np.uint8), "RGBA").save(dd / "rgba_sunset_000.png")
    mi = MaterialRenderer({"mesh": ply, "geo_model_path": ckpt, "shader_cfg": shader_cfg, "nerfDataType": True, "database_name": "tensoIR/lego",
                           "dataset_dir": str(tmp_path / "data"), "train_ray_num": 64, "split_manul": True, "split_borderline": 2}, training=True)
    assert isinstance(mi.database.H, float) and mi.test_num >= 1
    mi.eval()
    ev2 = mi({"eval": True, "index": 0})
    assert ev2["rgb_pr"].shape == (12, 12, 3) and ev2["rgb_gt"].shape == (12, 12, 3) and torch.isfinite(ev2["rgb_pr"]).all()
    assert {"spec_light", "diff_light", "occ_trace", "indirect_light"} <= set(ev2)
    mats = mat.extract_materials(str(tmp_path / "materials"))
    assert np.load(str(tmp_path / "materials" / "albedo.npy")).shape == (v.shape[0], 3) and np.isfinite(mats["roughness"]).all()
