"""Training harness (tensoflow_amd/trainer.py): schedule / bookkeeping on CPU, an optimisation run + checkpoint round trip
on the GPU."""
import math

import pytest
import torch


def test_cosine_lr_factor_and_multiplicative_form():
    from tensoflow_amd.trainer import cosine_lr_factor
    n, tgt = 1000, 0.05
    assert cosine_lr_factor(0, n, tgt) == pytest.approx(1.0)
    assert cosine_lr_factor(n, n, tgt) == pytest.approx(tgt)
    assert cosine_lr_factor(n // 2, n, tgt) == pytest.approx(0.5 * (1 - tgt) + tgt)
    # the trainer applies the schedule multiplicatively (lr *= f(s) / f(s-1)), one step late, exactly like the reference
    lr, lr_factor, pre = 1e-3, 1.0, 1.0
    for step in range(n):
        lr *= lr_factor
        cur = cosine_lr_factor(step, n, tgt)
        lr_factor, pre = cur / pre, cur
    assert lr == pytest.approx(1e-3 * cosine_lr_factor(n - 2, n, tgt), rel=1e-9)


def test_n_to_reso():
    from tensoflow_amd.trainer import n_to_reso
    assert n_to_reso(128 ** 3, [[-1.0, -1, -1], [1, 1, 1]]) == [128, 128, 128] or n_to_reso(128 ** 3, [[-1.0, -1, -1], [1, 1, 1]]) == [127, 127, 127]
    r = n_to_reso(2_000_000, [[-1.0, -0.5, -0.25], [1, 0.5, 0.25]])
    assert r[0] > r[1] > r[2] and abs(r[0] * r[1] * r[2] - 2_000_000) / 2_000_000 < 0.05


@pytest.mark.gpu
def test_material_trainer_fits_and_checkpoints(tmp_path):
    """A short optimisation run on a synthetic target lowers the colour loss; the reference's parameter groups and learning
    rates are in place; a checkpoint round trip restores parameters, schedule state and rendered colours."""
    from tensoflow_amd.network.fields import MCShadingNetwork
    from tensoflow_amd.synth import sphere_surface_points, sphere_torus_mesh
    from tensoflow_amd.trainer import MaterialTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(6033)
    verts, faces = sphere_torus_mesh(24, 48, 32, 16)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=64, nis_diffuse_sample_num=32,
               nis_specular_sample_num=32, outer_light_version="envlight")
    net = MCShadingNetwork(cfg, (verts, faces), aabb, 2.0 / 31)
    tr = MaterialTrainer(net, dict(total_step=200, nis_loss_iter=5, nis_start_iter_diffuse=10, nis_update_interval_diffuse=10,
                                   nis_start_iter_specular=10, nis_update_interval_specular=10))
    lrs = [g["lr"] for g in tr.optimizer.param_groups]
    assert lrs[:4] == [1e-2, 1e-2, 1e-2, 1e-3] and len(lrs) == 4 + 4 + 4          # mat lines / planes, env, nets, 2 x flow groups
    # every non-copy parameter is optimised except the coupling blocks' Reshift scale / offset, which the reference registers
    # as nn.Parameter(requires_grad=False) constants (flow.py Reshift)
    in_opt = {id(p) for p in tr.trainable()}
    left_out = [n_ for n_, p in net.named_parameters() if "_copy" not in n_ and id(p) not in in_opt]
    assert all("flow_" in n_ for n_ in left_out) and sum(net.get_parameter(n_).numel() for n_ in left_out) <= 16, left_out
    pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(512, seed=5)]
    target = torch.sigmoid(4.0 * pts)                                               # a smooth colour field on the surface
    first = None
    copy0 = {k: v.clone() for k, v in net.flow_diffuse_copy.state_dict().items()}
    for _ in range(40):
        info = tr.train_step(pts, view, nrm, target)
        assert torch.isfinite(info["loss"])
        first = float(info["loss_rgb"]) if first is None else first
    assert float(info["loss_rgb"]) < 0.7 * first
    assert tr.step_count == 40 and tr.pre_lr_factor < 1.0
    # the frozen copies are refreshed from the trainable flows at the START of steps 9, 19, 29, 39 (update_step precedes the
    # shading in MaterialRenderer.train_step, materialRenderer.py:549): after step 39's optimizer update they trail by one step
    a, b = net.flow_diffuse.state_dict(), net.flow_diffuse_copy.state_dict()
    assert not all(torch.equal(a[k], b[k]) for k in a) and not all(torch.equal(copy0[k], b[k]) for k in b)
    assert tr.refresh_flow_copies(49) == ["diffuse", "specular"]
    b = net.flow_diffuse_copy.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not any(p.requires_grad for p in net.flow_diffuse_copy.parameters())
    # checkpoint round trip
    with torch.no_grad():
        ref_col, _ = net(pts, view, nrm)
    path = str(tmp_path / "model.pth")
    tr.save(path)
    ck = torch.load(path, weights_only=False)
    assert {"step", "best_para", "lr_factor", "pre_lr_factor", "lr_xyz", "lr_net", "optimizer_state_dict", "N_voxel_list",
            "network_state_dict"} <= set(ck)
    # the reference's material checkpoints are MaterialRenderer.state_dict(): every key under 'shader_network.'
    assert all(k.startswith("shader_network.") for k in ck["network_state_dict"]) and "shader_network.mat_plane.0" in ck["network_state_dict"]
    assert {"loss_rgb", "loss_mat_reg", "loss_diffuse_light", "loss_nis"} <= set(info)      # the reference's objective, term by term
    net2 = MCShadingNetwork(cfg, (verts, faces), aabb, 2.0 / 31)
    tr2 = MaterialTrainer(net2, tr.cfg)
    tr2.load(path)
    assert tr2.step_count == 40 and tr2.cur_lr_net == pytest.approx(tr.cur_lr_net) and tr2.lr_factor == pytest.approx(tr.lr_factor)
    with torch.no_grad():
        col2, _ = net2(pts, view, nrm)
    assert torch.equal(col2, ref_col)
    assert tr2.loaded_keys == len(ck["network_state_dict"])
    assert math.isfinite(float(tr2.train_step(pts, view, nrm, target)["loss"]))
    # a round-1 file (bare keys) still loads; a file with nothing in common raises instead of silently loading nothing
    bare = dict(ck, network_state_dict={k[len("shader_network."):]: v for k, v in ck["network_state_dict"].items()})
    tr2.load(bare, load_optimizer=False)
    assert tr2.loaded_keys == len(ck["network_state_dict"])
    with pytest.raises(RuntimeError):
        tr2.load(dict(ck, network_state_dict={"color_network.foo": torch.zeros(1)}), load_optimizer=False)
    # before nis_loss_iter the two trainable flows receive no gradient: they are not exchanged (and Adam skips them on every rank)
    early, late = tr.trainable(step=0), tr.trainable(step=10)
    assert len(early) < len(late) == len(tr.trainable())


def test_shape_schedule_and_loss_terms():
    """Shape-stage bookkeeping on CPU: the voxel schedule of configs/shape/syn/compressor.yaml, the staged weights and the
    reductions of network/loss.py on hand-made render outputs."""
    from tensoflow_amd.trainer import SHAPE_CFG, _staged_ratio, n_to_reso, shape_loss_terms, voxel_schedule
    sched = voxel_schedule(128 ** 3 + 1, 512 ** 3 + 1, [20000, 40000])
    assert sched[0] == 128 ** 3 + 1 and sched[-1] == 512 ** 3 + 1 and len(sched) == 3
    assert abs(sched[1] - 256 ** 3) / 256 ** 3 < 1e-3                                  # log-spaced: the geometric mean
    assert n_to_reso(sched[0], SHAPE_CFG["aabb"]) == [128, 128, 128]
    assert voxel_schedule(1000, 8000, None) == [1000]
    assert _staged_ratio(5, None, [1.0, 1.0]) == 1.0
    assert _staged_ratio(5, [0, 10], [0.3, 0.1]) == 1.0 and _staged_ratio(10, [0, 10], [0.3, 0.1]) == 0.1

    torch.manual_seed(0)
    rn, N = 16, 200
    out = {"ray_rgb": torch.rand(rn, 3), "gradient_error": torch.rand(N), "acc": torch.rand(rn, 1), "std": torch.tensor(0.05),
           "radiance": torch.rand(rn, 3), "roughness_weights": torch.rand(rn), "loss_occ": torch.rand(7), "loss_sparse": torch.tensor(0.4),
           "loss_hessian": torch.tensor(2.0), "loss_tv_sdf": torch.tensor(0.3), "loss_gaussian": torch.tensor(9.0),
           "sdf_pts": torch.randn(N, 3) * 0.7, "sdf_vals": torch.randn(N) * 0.1}
    batch = {"rgbs": torch.rand(rn, 3), "masks": (torch.rand(rn) > 0.5).float()}
    cfg = {**SHAPE_CFG, "loss": SHAPE_CFG["loss"] + ["Hessian"], "hessian_update_list": [0, 100], "hessian_ratio": [0.1, 0.05],
           "eikonal_weight_anneal_end": 200}
    t = shape_loss_terms(cfg, out, batch, step=100)
    w = out["roughness_weights"]
    char = lambda a: torch.sqrt(((a - batch["rgbs"]) ** 2).sum(-1) + 0.001)
    assert torch.allclose(t["loss_rgb"], char(out["ray_rgb"]) * (1 - w)) and torch.allclose(t["loss_radiance"], char(out["radiance"]) * w)
    assert torch.allclose(t["loss_eikonal"], out["gradient_error"] * 0.05)              # annealed: half way to 0.1
    assert float(t["loss_hessian"]) == pytest.approx(2.0 * 5e-4 * 0.05) and float(t["loss_sparse"]) == pytest.approx(0.4 * 0.02)
    assert float(t["loss_tv_sdf"]) == pytest.approx(0.03) and float(t["loss_gaussian"]) == pytest.approx(9.0 * 5e-4)
    assert float(t["loss_occ"]) == pytest.approx(float(out["loss_occ"].mean()))
    bce = torch.nn.functional.binary_cross_entropy(out["acc"].clip(1e-3, 1 - 1e-3)[:, 0], batch["masks"])
    assert float(t["loss_mask"]) == pytest.approx(float(bce) * 0.01, rel=1e-6)
    assert "loss_std" not in t and {"loss_sdf_large", "loss_sdf_small"} <= set(t)
    # the initial-SDF regulariser: points inside r < 0.1 must have sdf < r - 0.1, points beyond 1.05 sdf > r - 1.05; cosine-annealed
    norm = out["sdf_pts"].norm(dim=-1)
    big = norm > 1.05
    ll = torch.clamp((norm[big] - 1.05) - out["sdf_vals"][big], min=0)
    anneal = (math.cos(0.1 * math.pi) + 1) / 2
    assert float(t["loss_sdf_large"]) == pytest.approx(float(ll.sum() / ((ll > 1e-5).sum() + 1e-3)) * anneal, rel=1e-5)
    assert "loss_sdf_large" not in shape_loss_terms(cfg, out, batch, step=1000)
    assert float(shape_loss_terms({**cfg, "eikonal_weight_anneal_begin": 150}, out, batch, 100)["loss_eikonal"].sum()) == 0.0


@pytest.mark.gpu
def test_shape_trainer_runs_schedule_and_resumes(tmp_path):
    """A short shape-stage run on synthetic rays: the loss falls, the alpha mask is refreshed and the grid is upsampled at the
    scheduled steps (new Parameters, new optimizer, one more mip level), and a checkpoint in the reference layout resumes."""
    from tensoflow_amd.network.shapeRenderer import ShapeRenderer
    from tensoflow_amd.synth import pinhole_rays
    from tensoflow_amd.trainer import ShapeTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(6033)

    def make(grid, max_levels=1):
        return ShapeRenderer(dict(gridSize=list(grid), max_levels=max_levels, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False,
                                  device="cuda", nerfDataType=True, clip_sample_variance=False, apply_occ_loss=False,
                                  anneal_end=20), training=False).to(dev)

    tr = ShapeTrainer(make, dict(total_step=100, N_voxel_init=24 ** 3, N_voxel_final=48 ** 3, upsample_list=[5], update_AlphaMask_lst=[3],
                                 loss=["nerf_render", "eikonal", "std", "init_sdf_reg", "Sparse", "TV", "mask", "Hessian"]))
    assert tr.net.gridSize.tolist() == [24, 24, 24] and tr.N_voxel_list == [48 ** 3]
    lrs = [g["lr"] for g in tr.optimizer.param_groups]
    assert lrs == [1e-2, 1e-2, 1e-3, 1e-3, 1e-2, 1e-3]                  # sdf lines / planes, decoder, variance, env map, shading MLPs
    o, d, radii, cos = [torch.from_numpy(a).to(dev) for a in pinhole_rays(512, seed=2)]
    inside = ((torch.cross(o, d, dim=-1).norm(dim=-1)) < 0.45).float()                # rays passing a sphere of radius 0.45
    batch = {"rays_o": o, "rays_d": d, "dirs": d, "radiis": radii, "rays_cos": cos, "masks": inside,
             "rgbs": torch.where(inside[:, None] > 0, torch.tensor([0.2, 0.5, 0.8], device=dev), torch.ones(3, device=dev)).expand(-1, 3).contiguous()}
    hist, events = [], {}
    planes_before = [p for p in tr.net.sdf_network.sdf_plane]
    opt_before = tr.optimizer
    for s in range(12):
        info = tr.train_step(batch)
        assert torch.isfinite(info["loss"]), (s, info)
        hist.append(float(info["loss"]))
        for e in info["events"]:
            events[e] = s
    assert events == {"alpha_mask": 3, "upsample": 5}
    assert tr.net.alphaMask is not None and tr.net.max_levels == 2 and tr.net.gridSize.tolist() == [48, 48, 48] and tr.N_voxel_list == []
    assert tr.optimizer is not opt_before and all(a is not b for a, b in zip(planes_before, tr.net.sdf_network.sdf_plane))
    assert {id(p) for p in tr.net.sdf_network.sdf_plane} <= {id(p) for p in tr.trainable()}
    assert tr.cur_lr_xyz < 1e-2 * 0.5 + 1e-12 and min(hist[-3:]) < hist[0]
    path = str(tmp_path / "shape.pth")
    tr.save(path)
    ck = torch.load(path, weights_only=False)
    assert {"step", "best_para", "lr_factor", "pre_lr_factor", "lr_xyz", "lr_net", "optimizer_state_dict", "N_voxel_list",
            "network_state_dict", "kwargs", "alphaMask.mask", "alphaMask.shape", "alphaMask.aabb"} <= set(ck)
    tr2 = ShapeTrainer.resume(path, make, tr.cfg)
    assert tr2.step_count == 12 and tr2.net.gridSize.tolist() == [48, 48, 48] and tr2.net.max_levels == 2 and tr2.net.alphaMask is not None
    a, b = tr.net.state_dict(), tr2.net.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert torch.isfinite(tr2.train_step(batch)["loss"])



def test_trainer_pinned_to_reference_loss_and_schedule(golden):
    """tests/golden/trainer.npz = outputs of the reference's OWN network/loss.py classes, compute_rgb_loss of both renderers,
    compute_diffuse_light_regularization, TrainerInv.update_learning_rate / N_voxel_list / N_to_reso and
    MaterialRenderer._construct_ray_batch_nerf on seeded inputs (tools/gen_golden.py:gen_trainer)."""
    import types
    from tensoflow_amd import trainer as T
    from tensoflow_amd.dataset import construct_ray_batch_nerf_material
    g = golden("trainer")
    pr = {k[3:]: v for k, v in g.a.items() if k.startswith("pr/")}
    gt = {k[3:]: v for k, v in g.a.items() if k.startswith("gt/")}
    for kind in ("l2", "l1", "smooth_l1", "charbonier"):
        assert torch.allclose(T.rgb_loss(kind, pr["ray_rgb"], gt["rgbs"]), g[f"shape_rgb_loss/{kind}"], atol=1e-7)
    upsample = [2000, 5000, 10000, 20000]
    cfg = {**T.SHAPE_CFG, "loss": ["nerf_render", "eikonal", "std", "init_sdf_reg", "occ", "Sparse", "Hessian", "TV", "mask", "Gaussian"],
           "eikonal_weight": 0.1, "eikonal_weight_anneal_begin": 1000, "eikonal_weight_anneal_end": 4000, "sparse_update_list": upsample,
           "sparse_ratio": [1.0, 0.5, 0.25, 0.1], "hessian_update_list": upsample, "hessian_ratio": [1.0, 0.8, 0.3, 0.0],
           "apply_std_loss": True, "std_loss_weight": 0.05}
    mcfg = {**T.DEFAULT_CFG}
    for st in [int(v) for v in g["steps"]]:
        out = {k: v for k, v in pr.items() if k not in ("radiance", "roughness_weights") or st > 20000}
        terms = T.shape_loss_terms(cfg, out, {"rgbs": gt["rgbs"], "masks": gt["masks"]}, st)
        ref = {k.split("/")[-1]: v for k, v in g.a.items() if k.startswith(f"shape/{st}/")}
        assert set(terms) == set(ref) - {"total"}, (st, sorted(terms), sorted(ref))
        for k, v in terms.items():
            assert float(v.mean()) == pytest.approx(float(ref[k]), rel=1e-6, abs=1e-9), (st, k)
        assert float(sum(v.mean() for v in terms.values())) == pytest.approx(float(ref["total"]), rel=1e-6)
        mterms = T.material_loss_terms(mcfg, pr["ray_rgb"], {"diffuse_light": pr["diffuse_light"], "loss_nis": pr["loss_nis"]}, gt["rgbs"],
                                       pr["loss_mat_reg"], st)
        mref = {k.split("/")[-1]: v for k, v in g.a.items() if k.startswith(f"mat/{st}/")}
        assert set(mterms) == set(mref) - {"total"}
        for k, v in mterms.items():
            assert float(v.mean()) == pytest.approx(float(mref[k]), rel=1e-6, abs=1e-9), (st, k)
        assert float(sum(v.mean() for v in mterms.values())) == pytest.approx(float(mref["total"]), rel=1e-6)
    # learning-rate factor (multiplicative form), voxel schedule, N_to_reso
    pre = 1.0
    for st, fac, cur in g["lr_factor"].tolist():
        c = T.cosine_lr_factor(st, 40000, 5e-2)
        assert c == pytest.approx(cur, rel=1e-12) and c / pre == pytest.approx(fac, rel=1e-12)
        pre = c
    nv = T.voxel_schedule(128 ** 3, 400 ** 3, upsample)
    assert nv == g["N_voxel_list"].tolist()
    assert [T.n_to_reso(n, g["N_to_reso_bbox"]) for n in nv] == g["N_to_reso"].tolist()
    # the material stage's ray table
    info = {"imgs": g["rays/imgs"], "Ks": g["rays/Ks"], "poses": g["rays/poses"]}
    batch, rn, h, w = construct_ray_batch_nerf_material(info)
    assert (rn, h, w) == (60, 5, 6)
    for k in ("rays_o", "rays_d", "human_poses", "rgb"):
        assert torch.allclose(batch[k], g["rays/out_" + k], atol=1e-6), k
    assert torch.allclose(batch["rays_d"].norm(dim=-1), torch.ones(rn), atol=1e-6)      # unit directions: depth is in world units
