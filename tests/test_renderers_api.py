"""Drop-in renderer modules (network.shapeRenderer / network.materialRenderer): CPU-side surface -- state_dict keys against
the reference checkpoint held by the goldens, parameter groups, checkpoint dictionary layout.  No kernel is launched."""
import numpy as np
import pytest
import torch

from conftest import AABB

SHAPE_CFG = dict(gridSize=[32, 32, 32], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False, device="cpu",
                 nerfDataType=True)


def test_shape_renderer_state_dict_and_groups(golden):
    from tensoflow_amd.network.shapeRenderer import ShapeRenderer
    g = golden("march_r32")
    r = ShapeRenderer(SHAPE_CFG, training=False)
    sd = r.state_dict()
    # the golden holds the reference ShapeRenderer's state_dict minus the keys its generator strips (tools/gen_golden.py:275)
    stripped = lambda k: "FG_LUT" in k or "envlight.base" in k or "outer_light" in k or "gaussian" in k
    assert {k for k in sd if not stripped(k)} == set(g.sd)
    # the late-training golden keeps the Gaussian buffers and has the radiance MLP: the full key set matches
    gl = golden("march_late_r32")
    rl = ShapeRenderer({**SHAPE_CFG, "has_radiance_field": True}, training=False)
    strip2 = lambda k: "FG_LUT" in k or "envlight.base" in k or "outer_light" in k
    assert {k for k in rl.state_dict() if not strip2(k)} == set(gl.sd)
    for k, v in gl.sd.items():
        assert tuple(rl.state_dict()[k].shape) == tuple(v.shape), k
    assert torch.allclose(rl.state_dict()["sdf_network.gaussian2d.kernel"], gl.sd["sdf_network.gaussian2d.kernel"])
    for k, v in g.sd.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert abs(float(r.stepSize) - float(g["step_size"])) < 1e-7 and abs(float(r.base_radii) - float(g["base_radii"])) < 1e-7
    groups = r.get_train_opt_params(0.02, 0.001, 0.0005)
    assert [gr["lr"] for gr in groups] == [0.02, 0.02, 0.001, 0.001, 0.0005, 0.001]      # shapeRenderer.py:372-381
    n_grouped = sum(len(list(gr["params"])) for gr in groups)
    assert n_grouped == len(list(r.parameters()))
    assert r.get_anneal_val(25000) == 0.5 and r.get_anneal_val(10 ** 6) == 1.0


def test_shape_renderer_refuses_unbuilt_modes():
    from tensoflow_amd.network.shapeRenderer import ShapeRenderer
    with pytest.raises(NotImplementedError):
        ShapeRenderer(SHAPE_CFG, training=True)                    # dataset side
    with pytest.raises(NotImplementedError):
        ShapeRenderer({**SHAPE_CFG, "predict_BG": True}, training=False)
    # use_occ_grid builds the on-device occupancy grid (march.OccGrid) and its state rides in the checkpoint (shapeRenderer.py:343-353)
    r = ShapeRenderer({**SHAPE_CFG, "use_occ_grid": True, "occ_grid_reso": 16}, training=False)
    assert r.occ_grid is not None and tuple(r.occ_grid.binaries.shape) == (1, 16, 16, 16)
    ck = r.ckpt_to_save()
    assert "occ_grid_state_dict" in ck
    # nerfacc.OccGridEstimator is an nn.Module inside the reference's ShapeRenderer: its six buffers ride in network_state_dict
    # under `occ_grid.` (and again in occ_grid_state_dict); a checkpoint written there must pass the strict load here and vice versa
    ref_keys = {"occ_grid." + k for k in ("resolution", "aabbs", "occs", "binaries", "grid_coords", "grid_indices")}
    assert ref_keys <= set(ck["network_state_dict"])
    sd = {k: v.clone() for k, v in ck["network_state_dict"].items()}
    assert sd["occ_grid.resolution"].dtype == torch.int32 and sd["occ_grid.binaries"].dtype == torch.bool
    assert sd["occ_grid.grid_coords"].shape == (16 ** 3, 3) and sd["occ_grid.aabbs"].shape == (1, 6) and sd["occ_grid.occs"].shape == (16 ** 3,)
    sd["occ_grid.occs"] = torch.rand(16 ** 3)
    sd["occ_grid.binaries"] = (sd["occ_grid.occs"] > 0.5).reshape(1, 16, 16, 16)
    r2 = ShapeRenderer({**SHAPE_CFG, "use_occ_grid": True, "occ_grid_reso": 16}, training=False)
    r2.load_ckpt({"kwargs": ck["kwargs"], "network_state_dict": sd, "occ_grid_state_dict": {k[9:]: v for k, v in sd.items() if k.startswith("occ_grid.")}})
    assert torch.equal(r2.occ_grid.occs, sd["occ_grid.occs"]) and torch.equal(r2.occ_grid.binaries, sd["occ_grid.binaries"])
    with pytest.raises(RuntimeError):                 # a renderer built WITHOUT the grid refuses those keys, as the reference's strict load does
        ShapeRenderer(SHAPE_CFG, training=False).load_ckpt({"network_state_dict": sd})


def test_shape_renderer_ckpt_layout_and_upsample():
    from tensoflow_amd.network.shapeRenderer import AlphaGridMask, ShapeRenderer
    r = ShapeRenderer({**SHAPE_CFG, "gridSize": [16, 16, 16], "max_levels": 1}, training=False)
    vol = (torch.rand(12, 10, 8) > 0.5).float()
    r.alphaMask = AlphaGridMask("cpu", AABB, vol)
    ck = r.ckpt_to_save()
    assert set(ck) == {"kwargs", "network_state_dict", "alphaMask.shape", "alphaMask.mask", "alphaMask.aabb"}   # shapeRenderer.py:343-353
    assert ck["alphaMask.mask"].dtype == np.uint8 and ck["alphaMask.mask"].size == (12 * 10 * 8 + 7) // 8
    assert set(ck["kwargs"]) == {"aabb", "gridSize", "sdf_n_comp", "appearance_n_comp", "sdf_dim", "app_dim", "sdf_multires",
                                 "alphaMask_thres", "marched_weights_thres", "step_ratio", "max_levels"}
    r2 = ShapeRenderer({**SHAPE_CFG, "gridSize": [16, 16, 16], "max_levels": 1}, training=False)
    r2.load_ckpt(ck)
    assert torch.equal(r2.alphaMask.alpha_volume[0, 0], vol)
    for (k, a), (_, b) in zip(r.state_dict().items(), r2.state_dict().items()):
        assert torch.equal(a, b), k
    # grid upsample: fields.py:169-178 (resolution rounded to a multiple of 2^(levels-1), one more level)
    r2.upsample_sdf_grid([37, 37, 37])
    assert r2.gridSize.tolist() == [36, 36, 36] and r2.max_levels == 2
    assert tuple(r2.sdf_network.sdf_plane[0].shape) == (1, 36, 36, 36) and tuple(r2.sdf_network.sdf_line[0].shape) == (1, 36, 36, 1)
    assert abs(float(r2.stepSize) - 2.0 / 35 * 0.5) < 1e-6


def test_material_renderer_mesh_argument():
    from tensoflow_amd.network.materialRenderer import MaterialRenderer
    with pytest.raises(NotImplementedError):
        MaterialRenderer({"mesh": "x.ply"}, training=True)                 # the dataset side is not mirrored
    with pytest.raises(FileNotFoundError):
        MaterialRenderer({"mesh": "no_such_mesh.ply", "device": "cpu"}, training=False, nvs=True)
    with pytest.raises(NotImplementedError):
        MaterialRenderer({"mesh": "mesh.obj", "device": "cpu"}, training=False, nvs=True)
