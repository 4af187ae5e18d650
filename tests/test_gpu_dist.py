"""The gradient exchange of the N > 1 training path on ONE MI355X over RCCL (backend "nccl", world_size 1): the flat bucket views,
the post-accumulate hooks firing inside a real backward of MaterialTrainer, reduce_scatter_tensor / all_gather_into_tensor on the
backend's stream and finish() -- everything `dist.GradientExchange` does at 8 ranks except the other ranks (the step
being wrapped: /root/reference/train/trainer_inv.py:197-212).  At one rank every collective is the identity, so the exchanged step
must reproduce the un-exchanged one."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def rccl_one_rank():
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        yield dist
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _trainer(force):
    from tensoflow_amd.network.fields import MCShadingNetwork
    from tensoflow_amd.synth import sphere_torus_mesh
    from tensoflow_amd.trainer import MaterialTrainer
    torch.manual_seed(6033)
    verts, faces = sphere_torus_mesh(24, 48, 32, 16)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=64, nis_diffuse_sample_num=32,
               nis_specular_sample_num=32, outer_light_version="envlight")
    net = MCShadingNetwork(cfg, (verts, faces), aabb, 2.0 / 31)
    # flow copies in use and both NIS losses on from step 0: every parameter group receives a gradient
    tr = MaterialTrainer(net, dict(total_step=200, nis_loss_iter=0, nis_start_iter_diffuse=1, nis_update_interval_diffuse=1000,
                                   nis_start_iter_specular=1, nis_update_interval_specular=1000))
    tr.force_exchange = force
    assert tr.refresh_flow_copies(0) == ["diffuse", "specular"]          # the frozen copies take over the sampling (the flow-sampled pass)
    return net, tr


def _one_backward(tr, net, batch, seed):
    """The forward / backward half of MaterialTrainer.train_step (no optimizer step), under a fixed sampler seed -> {name: grad}."""
    from tensoflow_amd.trainer import material_loss_terms
    pts, view, nrm, target = batch
    step = tr.step_count
    net.train()
    ex = tr._exchange()
    if ex is not None:
        ex.zero_grad(expected=tr.trainable(step))
    else:
        tr.optimizer.zero_grad(set_to_none=True)
    tr.refresh_flow_copies(step)
    torch.manual_seed(seed)
    colors, outputs = net(pts, view, nrm, None, step, True)
    mat_reg = net.material_regularization(pts, nrm, outputs["metallic"], outputs["roughness"], outputs["albedo"], step) if tr.cfg["reg_mat"] else None
    terms = material_loss_terms(tr.cfg, colors, outputs, target, mat_reg, step)
    sum(v.mean() for v in terms.values()).backward()
    sent = ex.finish(expected=tr.trainable(step)) if ex is not None else 0
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, sent, ex


def test_gradient_exchange_runs_over_rccl_at_one_rank(rccl_one_rank):
    from tensoflow_amd.synth import sphere_surface_points
    dist = rccl_one_rank
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda:0")
    pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(512, seed=5)]
    batch = (pts, view, nrm, torch.sigmoid(4.0 * pts))
    net_a, tr_a = _trainer(force=False)
    net_b, tr_b = _trainer(force=True)
    assert all(torch.equal(p, q) for p, q in zip(net_a.parameters(), net_b.parameters()))
    for tr in (tr_a, tr_b):
        tr.step_count = 5          # (past nis_start_iter: the frozen copies sample)
    g_plain, _, ex_a = _one_backward(tr_a, net_a, batch, seed=1)
    g_plain2, _, _ = _one_backward(tr_a, net_a, batch, seed=1)           # run-to-run spread of the backward itself (float atomics)
    g_ex, sent, ex = _one_backward(tr_b, net_b, batch, seed=1)
    assert ex_a is None and ex is not None and ex.force and ex.mode == "rs_ag"
    # every bucket went through reduce_scatter_tensor + all_gather_into_tensor, and they were queued from the hooks DURING backward
    assert sent == len(ex.buckets) >= 1 and all(len(b["work"]) == 2 for b in ex.buckets)
    assert ex.launched_in_backward >= 2
    assert set(g_ex) == set(g_plain) and len(g_ex) > 20
    exact = 0
    for n in g_plain:
        spread = float((g_plain2[n] - g_plain[n]).abs().max())
        diff = float((g_ex[n] - g_plain[n]).abs().max())
        # identity collectives: the exchanged gradient differs from the plain one by no more than two plain backward passes differ from
        # each other (weight / grid gradients are summed with float atomics: their low bits depend on the order of arrival; 1e-6 of the
        # tensor's scale is the floor allowed where two plain runs happened to agree).  The bit-for-bit statement is the next block.
        assert diff <= 4.0 * spread + 1e-6 * float(g_plain[n].abs().max()), (n, diff, spread)
        exact += int(torch.equal(g_ex[n], g_plain[n]))
    print(f"exchange at one rank over RCCL: {len(ex.buckets)} buckets, {ex.launched_in_backward} collectives queued inside backward, "
          f"{exact} of {len(g_plain)} gradient tensors bit-identical to the un-exchanged backward (the others within the backward's own run-to-run spread)")
    # the parameters' .grad ARE views of the flat buckets
    for b in ex.buckets:
        for p, v in zip(b["params"], b["views"]):
            assert p.grad is None or p.grad.data_ptr() == v.data_ptr()
    # ---- the collectives alone are the identity on a filled bucket, bit for bit (mul by 1 / 1, reduce-scatter, all-gather)
    for b in ex.buckets:
        before = b["flat"].clone()
        b["sent"], b["work"] = False, []
        ex._send(b)
        for w in b["work"]:
            w.wait()
        torch.cuda.synchronize()
        assert len(b["work"]) == 2 and torch.equal(b["flat"], before)
    # ---- a full exchanged optimizer step leaves finite, changed parameters
    before = [p.detach().clone() for p in tr_b.trainable(5)]
    torch.manual_seed(2)
    info = tr_b.train_step(*batch)
    assert torch.isfinite(info["loss"]) and any(not torch.equal(a, p) for a, p in zip(before, tr_b.trainable(5)))


def test_exchange_guard_fires_on_an_unexpected_gradient_over_rccl(rccl_one_rank):
    """A parameter outside `expected` that receives a gradient is a broken contract: the hook raises out of backward (its bucket may
    already be on the wire)."""
    from tensoflow_amd.dist import GradientExchange
    dev = torch.device("cuda:0")
    a = torch.nn.Parameter(torch.ones(1000, device=dev))
    b = torch.nn.Parameter(torch.ones(3000, device=dev))
    ex = GradientExchange([a, b], 1, force_collectives=True)
    ex.zero_grad(expected=[a])
    with pytest.raises(RuntimeError, match="outside this step's `expected` set"):
        (a.sum() + 2.0 * b.sum()).backward()
    ex.finish(expected=[a])
    torch.cuda.synchronize()
    # the contract kept: only `a` is exchanged, `b` gets its None grad back
    ex.zero_grad(expected=[a])
    (3.0 * a.sum()).backward()
    ex.finish(expected=[a])
    torch.cuda.synchronize()
    assert torch.equal(a.grad, torch.full_like(a, 3.0)) and b.grad is None
    ex.remove()
