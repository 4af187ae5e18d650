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
               nis_specular_sample_num=32)
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
    for _ in range(40):
        info = tr.train_step(pts, view, nrm, target)
        assert torch.isfinite(info["loss"])
        first = float(info["loss_rgb"]) if first is None else first
    assert float(info["loss_rgb"]) < 0.7 * first
    assert tr.step_count == 40 and tr.pre_lr_factor < 1.0
    # the frozen copies were refreshed from the trainable flows at steps 9, 19, 29, 39
    a, b = net.flow_diffuse.state_dict(), net.flow_diffuse_copy.state_dict()
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
    net2 = MCShadingNetwork(cfg, (verts, faces), aabb, 2.0 / 31)
    tr2 = MaterialTrainer(net2, tr.cfg)
    tr2.load(path)
    assert tr2.step_count == 40 and tr2.cur_lr_net == pytest.approx(tr.cur_lr_net) and tr2.lr_factor == pytest.approx(tr.lr_factor)
    with torch.no_grad():
        col2, _ = net2(pts, view, nrm)
    assert torch.equal(col2, ref_col)
    assert math.isfinite(float(tr2.train_step(pts, view, nrm, target)["loss"]))
