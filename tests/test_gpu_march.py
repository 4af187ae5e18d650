"""GPU parity of the ray-march composition (sample_ray, render_core, split-sum shading) against the goldens
generated from the imported reference (ShapeRenderer.sample_ray / render_core / ShapeShadingNetwork.forward)."""
import numpy as np
import pytest
import torch

from conftest import AABB, parity, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test collected but no GPU is visible")
    return torch.device("cuda:0")


AUX_MAPS = ("approximate_light",)
AUX_VARS = ("variance", "variance_diffuse_vis", "variance_specular_vis")


def check_aux(out, ref, sfx="", human_expected=False, tol=TOL):
    """The rest of shade_mixed's output dict (fields.py:1241-1256, :1288-1291) against the reference run: `ref(key)` -> golden tensor.
    The variance figures are second moments of importance-sampling weights (the reference's per-point form is E[g^2] - E[g]^2 in
    fp32): held relative to each array's scale (true_rel_err, floor 1e-3 of its maximum) at 10x the pixel tolerance."""
    from conftest import true_rel_err
    for k in AUX_MAPS:
        parity(out[k + sfx].detach().cpu(), ref(k + sfx), tol=tol, label=f"shade_mixed aux {k + sfx}")
    for k in AUX_VARS:
        got, want = out[k + sfx].detach().cpu(), ref(k + sfx)
        assert got.shape == want.shape, (k + sfx, got.shape, want.shape)
        assert true_rel_err(got, want) < 10 * tol, (k + sfx, true_rel_err(got, want))
    hl, hl_ref = out["human_lights" + sfx].cpu(), ref("human_lights" + sfx)
    assert hl.shape == hl_ref.shape, (hl.shape, hl_ref.shape)                # one row per unmasked specular ray that misses
    parity(hl, hl_ref, tol=tol, label="shade_mixed human_lights")
    assert (float(hl_ref.abs().max()) > 0.1) == human_expected
    inter, inter_ref = out["inter" + sfx].cpu(), ref("inter" + sfx)
    assert inter.shape == inter_ref.shape
    hit = inter_ref.norm(dim=-1) < 5.0         # a missing ray's row is whatever the tracer leaves there (third party, unpinned): o + 10 d here
    assert 0 < int(hit.sum()) < hit.numel() and rel_err(inter[hit], inter_ref[hit]) < tol


def _field(g, dev):
    from tensoflow_amd.march import SdfField
    return SdfField(g.sd, AABB, [32, 32, 32], 3, device=dev)


def test_sample_ray_bit_exact_indices(golden, dev):
    from tensoflow_amd import march
    g = golden("march_r32")
    f = _field(g, dev)
    c = lambda k: g[k].to(dev)
    t0, t1, ridx = march.sample_ray(f, c("rays_o"), c("dirs"), c("near"), c("far"), c("radiis"), c("rays_cos"), float(g["base_radii"]))
    assert ridx.dtype == torch.int64 and torch.equal(ridx.cpu(), g["ray_indices"])       # bit-exact
    parity(t0.cpu(), g["t_starts"], label="sample_ray_bit_exact_indices:57.0")
    parity(t1.cpu(), g["t_ends"], label="sample_ray_bit_exact_indices:57.1")
    near, far = march.near_far_from_sphere(c("rays_o"), c("dirs"))
    parity(near.cpu(), g["near"], abs_tol=1e-6, label="sample_ray_bit_exact_indices:59.0")
    parity(far.cpu(), g["far"], abs_tol=1e-6, label="sample_ray_bit_exact_indices:59.1")


def test_sample_ray_merge_is_a_stable_sort_for_any_order_of_the_new_samples(dev):
    """tf_sample_ray_merge == torch.sort(cat([z, new_t]), stable) (shapeRenderer.py:851-869) also when the new samples do NOT arrive
    sorted among themselves (the inverse-CDF value b0 + frac * fl(b1 - b0) can exceed b1 by an ulp while the next sample equals b1)
    and when values tie (old samples first, then new ones in their own order): every output slot is written exactly once."""
    from tensoflow_amd import ops
    g = torch.Generator().manual_seed(3)
    rn, S, n_imp = 301, 96, 16
    z = torch.sort(torch.rand(rn, S, generator=g), dim=-1)[0]
    new_t = torch.rand(rn, n_imp, generator=g)                       # unsorted
    new_t[:, 3] = z[:, 10]                                           # ties with an old sample ...
    new_t[:, 7] = new_t[:, 2]                                        # ... and among the new ones
    new_t[:, 9] = torch.nextafter(new_t[:, 8], torch.ones(rn))       # one ulp apart, out of order
    sdf, nsdf = torch.randn(rn, S, generator=g), torch.randn(rn, n_imp, generator=g)
    z_out, sdf_out = ops.sample_ray_merge(z.to(dev), sdf.to(dev), new_t.to(dev), nsdf.to(dev))
    zz, idx = torch.sort(torch.cat([z, new_t], -1), dim=-1, stable=True)
    assert torch.equal(z_out.cpu(), zz)
    assert torch.equal(sdf_out.cpu(), torch.gather(torch.cat([sdf, nsdf], -1), 1, idx))


@pytest.mark.parametrize("rn,perturb,cap", [(1023, False, None), (257, True, 100.0), (3, False, 20.0)])
def test_sample_ray_kernels_match_torch_composition(golden, dev, rn, perturb, cap):
    """tf_sample_ray_init / _upsample / _merge (round 4) against the torch composition they replace (march.sample_ray_torch, itself
    pinned to the reference by `march_r32`): ragged ray counts, the stratification offset of perturb > 0, the sharpness cap of
    clip_sample_variance.  ray_indices identical; t within 1e-5 on 99.5 % of the samples and within 1e-3 on the rest: a sample that
    the inverse CDF places in an interval of tiny mass is (u - c0) / (c1 - c0) * (b1 - b0) with c1 - c0 ~ 1e-5 .. 1e-4, so the last
    bit of the cumulative sums (which depends on the order of the scan: torch's CPU and GPU cumsum and the wave scan here all
    differ) moves it by up to 1e-7 / 1e-5 * 0.03 = 3e-4 (measured: 0.12 % of the samples beyond 1e-5, max 1.3e-4)."""
    from tensoflow_amd import march
    from tensoflow_amd.synth import pinhole_rays
    g = golden("march_r32")
    f = _field(g, dev)
    o, d, radii, cos = [torch.from_numpy(a).to(dev) for a in pinhole_rays(rn, seed=5, h=64, w=64, focal=90.0)]
    near, far = march.near_far_from_sphere(o, d)
    t_rand = (torch.rand(rn, 1, generator=torch.Generator().manual_seed(1)) - 0.5).to(dev) if perturb else None
    a = march.sample_ray(f, o, d, near, far, radii, cos, float(g["base_radii"]), t_rand=t_rand, inv_s_cap=cap)
    b = march.sample_ray_torch(f, o, d, near, far, radii, cos, float(g["base_radii"]), t_rand=t_rand, inv_s_cap=cap)
    assert a[2].dtype == torch.int64 and torch.equal(a[2], b[2])
    assert a[0].shape == b[0].shape
    for x, y in ((a[0], b[0]), (a[1], b[1])):
        dlt = (x - y).abs()
        assert float((dlt > 1e-5).float().mean()) < 5e-3 and float(dlt.max()) < 1e-3, (float((dlt > 1e-5).float().mean()), float(dlt.max()))
    assert int(a[2].numel()) > 60 * rn // 2


def test_shape_shading(golden, dev):
    from tensoflow_amd.shape_shading import ShapeShader
    g = golden("march_r32")
    sh = ShapeShader(g.sd, [g["env_spec0"], g["env_spec1"], g["env_spec2"]], g["env_diffuse"], g["fg_lut"], device=dev)
    ridx = g["ray_indices"]
    nrm = torch.nn.functional.normalize(g["sa_grad"], dim=-1)
    col, occ, rough, refl = sh(g["sample_pts"].to(dev), nrm.to(dev), (-g["dirs"][ridx]).to(dev), g["sa_feat"].to(dev))
    parity(col.cpu(), g["shade_color"], label="shape_shading:113")
    parity(occ.cpu(), g["shade_occ_prob"], label="shape_shading:114")
    parity(rough.cpu(), g["shade_roughness"], label="shape_shading:115")
    parity(refl.cpu(), g["shade_reflective"], label="shape_shading:116")


def test_shape_shading_ragged_and_degenerate(golden, dev):
    """Fused shape-shade launch vs the oracle on inputs the golden does not hold: a sample count that is not a multiple of
    the 128-sample workgroup tile, unnormalised normals / view vectors, and the degenerate normal the reference patches
    (n.x + n.y == 0 -> (0, 1e-6, 1), fields.py:455-456)."""
    from oracle import march as om
    from tensoflow_amd.shape_shading import ShapeShader
    g = golden("march_r32")
    env = {"specular": [g["env_spec0"], g["env_spec1"], g["env_spec2"]], "diffuse": g["env_diffuse"]}
    sh = ShapeShader(g.sd, env["specular"], env["diffuse"], g["fg_lut"], device=dev)
    gen = torch.Generator().manual_seed(21)
    n = 1000 + 37
    pts = torch.rand(n, 3, generator=gen) * 2 - 1
    nrm = torch.randn(n, 3, generator=gen) * 3.0
    nrm[5] = torch.tensor([0.0, 0.0, 2.0])
    nrm[77] = torch.tensor([0.5, -0.5, 1.0])
    view = torch.randn(n, 3, generator=gen) * 0.3
    feat = torch.randn(n, 128, generator=gen) * 0.5
    col, occ, rough, refl = sh(pts.to(dev), nrm.to(dev), view.to(dev), feat.to(dev))
    rc, ro, rr, rf = om.shape_shade(g.sd, env, g["fg_lut"], pts, nrm, view, feat)
    parity(col.cpu(), rc, label="shape_shading_ragged_and_degenerate:138.0")
    parity(occ.cpu(), ro, label="shape_shading_ragged_and_degenerate:138.1")
    parity(rough.cpu(), rr, label="shape_shading_ragged_and_degenerate:139.0")
    parity(refl.cpu(), rf, label="shape_shading_ragged_and_degenerate:139.1")


def test_render_core(golden, dev):
    from tensoflow_amd import march
    from tensoflow_amd.shape_shading import ShapeShader
    g = golden("march_r32")
    f = _field(g, dev)
    sh = ShapeShader(g.sd, [g["env_spec0"], g["env_spec1"], g["env_spec2"]], g["env_diffuse"], g["fg_lut"], device=dev)
    c = lambda k: g[k].to(dev)
    inv_s = float(torch.exp(g.sd["deviation_network.variance"] * 10.0))
    out = march.render_core(f, c("rays_o"), c("dirs"), c("radiis"), c("rays_cos"), c("t_starts"), c("t_ends"), c("ray_indices"),
                            float(g["base_radii"]), inv_s, 0.5, shade_fn=lambda p, n, v, ft: sh(p, n, v, ft)[0])
    for k in ("ray_rgb", "acc", "normal", "gradient_error", "loss_sparse"):
        parity(out[k].cpu(), g["rc/" + k], label=f"render_core {k}")
    parity(out["loss_hessian"].cpu(), g["rc/loss_hessian"], tol=2e-3, label="render_core loss_hessian")
    # size-independent properties: weights of a ray sum to acc <= 1, rgb in [0, 1]
    assert float(out["acc"].max()) <= 1 + 1e-5 and float(out["ray_rgb"].min()) >= -1e-6


def test_march_full_size_properties(dev):
    """BASELINE config-1 shape (4096 rays, R=300, C=36, 3 mips): invariants that do not need the oracle."""
    from tensoflow_amd import march, ops
    from tensoflow_amd.synth import pinhole_rays, random_sdf_state
    sd = {"sdf_network." + k: v for k, v in random_sdf_state(seed=1, R=300).items()}
    f = march.SdfField(sd, AABB, [300, 300, 300], 3, device=dev)
    o, d, radii, cos = [torch.from_numpy(a).to(dev) for a in pinhole_rays(4096, seed=2)]
    near, far = march.near_far_from_sphere(o, d)
    base_radii = 2.0 / 2.0 / 300
    t0, t1, ridx = march.sample_ray(f, o, d, near, far, radii, cos, base_radii)
    assert (ridx[1:] >= ridx[:-1]).all()                                   # packed, sorted by ray
    assert (t1 >= t0).all()
    same = ridx[1:] == ridx[:-1]
    assert (t0[1:][same] >= t0[:-1][same] - 1e-6).all()                    # sorted along each ray
    out = march.render_core(f, o, d, radii, cos, t0, t1, ridx, base_radii, 20.0, 1.0)
    w_sum = torch.zeros(4096, device=dev).index_add_(0, ridx, out["weights"])
    parity(w_sum.cpu(), out["acc"][:, 0].cpu(), abs_tol=1e-5, label="march_full_size_properties:175")  # checksum of the scan
    assert float(out["acc"].max()) <= 1 + 1e-5 and float(out["acc"].mean()) > 0.01
    # linearity of compositing in the values
    v = torch.randn(ridx.shape[0], 3, device=dev)
    _, _, a = ops.composite(out["alpha"], ridx, v, 4096)
    _, _, b = ops.composite(out["alpha"], ridx, 2 * v, 4096)
    parity((2 * a).cpu(), b.cpu(), abs_tol=1e-6, label="march_full_size_properties:181")


# ------------------------------------------------------------------------------- drop-in modules
def test_module_tensosdf(golden, dev):
    from tensoflow_amd.network.fields import TensoSDF
    g = golden("tensosdf_r32_l3")
    m = TensoSDF(torch.tensor([32, 32, 32]), AABB, device=dev, init_n_levels=3)
    m.load_state_dict(g.sd)
    with torch.no_grad():
        out = m(g["pts"].to(dev), g["level"].to(dev))
        parity(out.cpu(), g["out_lvl"], label="module_tensosdf:192")
        parity(m.sdf(g["pts"].to(dev)).cpu(), g["out_none"][:, :1], label="module_tensosdf:193")
        grad, nh = m.gradient(g["pts"].to(dev), g["level"].to(dev), training=True, sdf=out[:, :1])
        parity(grad.cpu(), g["grad_lvl"], label="module_tensosdf:195.0")
        parity(nh.cpu(), g["normal_hessian"], tol=2e-3, label="module_tensosdf:195.1")
        grad0, none = m.gradient(g["pts"].to(dev), None, training=False)
        assert none is None and rel_err(grad0.cpu(), g["grad_none"]) < TOL
        # parameters changed in place -> the packed pyramid is rebuilt
        m.sdf_plane[0].mul_(0.5)
        assert not torch.allclose(m(g["pts"].to(dev), g["level"].to(dev)), out)


@pytest.mark.parametrize("tag", ["mr3", "mr2"])
def test_module_tensosdf_multires(golden, dev, tag):
    """TensoSDF(sdf_multires = 3 | 2) (fields.py:66-91, :293-299; the reference class's default is 3, no shipped config sets it): the
    decoder then has 3C + 3 + 6 m inputs and runs as a composition of the gather / encoding / dense-layer kernels (ops.sdf_forward,
    ops.sdf_alpha and autograd.SdfAlphaFn dispatch on the first layer's width).  Forward, FD gradient, hessian term and the parameter
    gradients through compute_sdf_alpha's autograd node against the imported reference."""
    from conftest import in_fp64
    from oracle import vm_field as ovm
    from tensoflow_amd.autograd import SdfAlphaFn
    from tensoflow_amd.network.fields import TensoSDF
    g = golden("tensosdf_" + tag)
    mr = int(g["multires"])
    m = TensoSDF(torch.tensor([32, 32, 32]), AABB, device=dev, init_n_levels=3, sdf_multires=mr)
    assert m.sdf_mat[0].weight.shape == (256, 108 + 3 + 6 * mr)
    m.load_state_dict(g.sd)
    pts, level = g["pts"].to(dev), g["level"].to(dev)
    t64 = in_fp64(ovm.sdf_forward, g.sd, g["pts"], g["level"], AABB, 3)
    with torch.no_grad():
        out = m(pts, level)
        parity(out.cpu(), g["out_lvl"], truth=t64, label=f"TensoSDF multires {mr} forward")
        parity(m(pts, None).cpu(), g["out_none"], truth=in_fp64(ovm.sdf_forward, g.sd, g["pts"], None, AABB, 3), label=f"TensoSDF multires {mr} forward, no levels")
        parity(m.sdf(pts, level).cpu(), g["out_lvl"][:, :1], truth=t64[:, :1], label=f"TensoSDF multires {mr} sdf")
        grad, nh = m.gradient(pts, level, training=True, sdf=out[:, :1])
        g64, nh64 = in_fp64(ovm.sdf_gradient, g.sd, g["pts"], g["level"], AABB, 3, g["grid_size"], sdf=t64[:, :1], training=True)
        parity(grad.cpu(), g["grad_lvl"], truth=g64, label=f"TensoSDF multires {mr} gradient")
        parity(nh.cpu(), g["normal_hessian"], truth=nh64, abs_tol=2e-3, label=f"TensoSDF multires {mr} normal_hessian")
    # the training direction: d sum(out * w) / d parameters through SdfAlphaFn (outputs sdf | features of the centre tap)
    n = pts.shape[0]
    params = list(m.sdf_plane) + list(m.sdf_line) + m._w()
    inv_s = torch.tensor([20.0], device=dev)
    alpha, gr, feat, sdf, nh2 = SdfAlphaFn.apply(pts, level[:, 0].contiguous(), torch.full((n,), 0.01, device=dev), torch.zeros(n, 3, device=dev), inv_s, 0.5,
                                                 AABB, [float(u) for u in m.units], 3, *params)
    w = g["bwd_w"].to(dev)
    ((sdf * w[:, 0]).sum() + (feat * w[:, 1:]).sum()).backward()
    names = [f"sdf_plane.{i}" for i in range(3)] + [f"sdf_line.{i}" for i in range(3)] + ["sdf_mat.0.weight", "sdf_mat.0.bias", "sdf_mat.2.weight", "sdf_mat.2.bias"]
    for k, p in zip(names, params):
        ref = g.grad[k]
        err = float((p.grad.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
        assert err < 2e-4, (k, err)                      # scatter atomics and the dense layers' split reductions reorder fp32 sums


def test_module_tensoflow(golden, dev):
    from tensoflow_amd.network.flow import TensoFlow
    g = golden("tensoflow_r32")
    m = TensoFlow(2, AABB, device=dev, gridSize=[32, 32, 32])
    m.load_state_dict(g.sd)
    m.eval()
    c = lambda k: g[k].to(dev)
    with torch.no_grad():
        parity(m.tenso_feature(c("pts")).cpu(), g["cond_feat"], label="module_tensoflow:211")
        ang, logj = m.sample(c("pts"), c("view_angles"), c("roughness"), 32, return_jacobian=True)
        assert float(torch.quantile((ang.cpu() - g["angles_32"]).abs().flatten(), 0.999)) < TOL
        z, logq = m(c("pts"), c("view_angles"), c("roughness"), c("x_rid"), return_jacobian=True, rays_id=c("rays_id"))
        parity(z.cpu(), g["z_rid"], label="module_tensoflow:215.0")
        parity(logq.cpu(), g["logq_rid"], label="module_tensoflow:215.1")


def test_module_mcshading(golden, dev):
    from tensoflow_amd.network.fields import MCShadingNetwork
    g = golden("shading_small")
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=n_fd, specular_sample_num=n_fs, nis_diffuse_sample_num=sn_d,
               nis_specular_sample_num=sn_s, outer_light_version="envlight")
    m = MCShadingNetwork(cfg, (g["verts"].numpy(), g["faces"].numpy()), AABB, float(g["unit_size"]))
    missing, unexpected = m.load_state_dict(g.sd, strict=False)
    assert not missing, missing                                  # every parameter of the mirror exists in the reference checkpoint
    m.shader()
    colors, outputs = m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), None, None, False)   # autograd on: flow pass
    parity(colors.detach().cpu(), g.out["rgb_pr_nis"], label="module_mcshading:229")
    parity(outputs["albedo"].detach().cpu(), g.out["albedo"], label="module_mcshading:230")
    # eval (no autograd, step=None): the reference runs the fixed-sampler pass (-> colors and the plain outputs) and the
    # flow-sampler pass (-> the *_nis outputs), fields.py:1467-1473
    with torch.no_grad():
        colors, outputs = m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), None, None, False)
    parity(colors.cpu(), g["colors"], label="module_mcshading:235")
    parity(outputs["rgb_pr_nis"].cpu(), g.out["rgb_pr_nis"], label="module_mcshading:236")
    for k in ("albedo", "roughness", "metallic", "diffuse_light", "specular_light", "diffuse_color", "specular_color", "visibility",
              "indirect_light"):
        parity(outputs[k].cpu(), g.out[k], label=f"MCShadingNetwork outputs {k}")
        if k + "_nis" in g.out:
            parity(outputs[k + "_nis"].cpu(), g.out[k + "_nis"], label=f"MCShadingNetwork outputs {k}_nis")
    for sfx in ("", "_nis"):
        check_aux(outputs, lambda k: g.out[k], sfx)
        assert float(outputs["loss_nis" + sfx]) == 0.0
    ref_keys = {k for k in g.out}
    assert ref_keys <= set(outputs.keys()) | {"rgb_pr_nis"}, sorted(ref_keys - set(outputs.keys()))
    img = m.env_light(16, 32)
    assert img.shape == (16, 32, 3) and torch.isfinite(img).all() and not img.requires_grad
    reg = m.material_regularization(g["pts"].to(dev), None, outputs["metallic"], outputs["roughness"], outputs["albedo"], 100)
    assert reg.shape == (1,) and float(reg) >= 0
    assert m.update_step(998) == [] and m.update_step(999) == ["diffuse", "specular"] and m.use_flow_diffuse_copy
    env = m.outer_light.direct_light(g["env_dirs"].to(dev)[None, None])
    assert env.shape == (1, 1, g["env_dirs"].shape[0], 3) and rel_err(env[0, 0].detach().cpu(), g["env_direct"]) < TOL
    env.sum().backward()                                          # cube lookup has a HIP backward
    assert m.outer_light.base.grad is not None and float(m.outer_light.base.grad.abs().sum()) > 0


def test_mcshading_eval_follows_parameter_updates(golden, dev):
    """The fused evaluator is cached on the parameters' version counters: train step -> eval -> train step -> eval must render
    with the weights of THAT moment (a stale pack would make every validation between two flow-copy refreshes reuse the first
    pack), and equal an evaluator built from scratch."""
    from tensoflow_amd.network.fields import MCShadingNetwork
    from tensoflow_amd.shading import MCShader
    g = golden("shading_small")
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=n_fd, specular_sample_num=n_fs, nis_diffuse_sample_num=sn_d,
               nis_specular_sample_num=sn_s, outer_light_version="envlight")
    m = MCShadingNetwork(cfg, (g["verts"].numpy(), g["faces"].numpy()), AABB, float(g["unit_size"]))
    m.load_state_dict(g.sd, strict=False)
    args = (g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev))
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-2)

    def eval_colors():
        with torch.no_grad():
            return m(*args, None, None, False)[0].clone()

    def train_step():
        opt.zero_grad(set_to_none=True)
        colors, out = m(*args, None, 600, True)
        (colors.sum() + out["loss_nis"]).backward()
        opt.step()

    c0 = eval_colors()
    assert m.shader() is m.shader()                       # unchanged parameters: the pack is reused
    train_step()
    c1 = eval_colors()
    train_step()
    c2 = eval_colors()
    assert not torch.equal(c0, c1) and not torch.equal(c1, c2)
    fresh = MCShader({k: t.detach() for k, t in m.state_dict().items()}, g["verts"].numpy(), g["faces"].numpy(), AABB, float(g["unit_size"]),
                     device=dev, n_fixed_diffuse=n_fd, n_fixed_specular=n_fs)
    assert torch.equal(fresh.shade_fixed(*args)["colors"], c2)
    m.load_state_dict(g.sd, strict=False)                 # load_state_dict copies in place: versions move, the pack follows
    assert torch.equal(eval_colors(), c0)


def test_update_alpha_mask_golden(golden, dev):
    """ShapeRenderer.updateAlphaMask / compute_gridAlpha (shapeRenderer.py:257-325) run by the reference on the march_r32 network
    (tests/golden/alpha_mask_r32.npz): the binary volumes bit-exact, the raw alpha lattice to 1e-4, the shrunk aabb equal --
    first without a previous mask, then on a finer lattice sampled through the first mask."""
    from tensoflow_amd import march
    g, base = golden("alpha_mask_r32"), golden("march_r32")
    assert bool(g["sd_is_march_r32"])
    f = march.SdfField(base.sd, AABB, [32, 32, 32], 3, device=dev)
    inv_s, thres, mul = float(g["inv_s"]), float(g["thres"]), float(g["mul_length"])
    m1, aabb1, raw1 = march.update_alpha_mask(f, inv_s, grid=(24, 20, 28), thres=thres, mul_length=mul, return_alpha=True)
    parity(raw1.cpu(), g["alpha_24x20x28"], label="update_alpha_mask_golden:307")
    assert torch.equal(m1.volume.cpu().bool(), g["mask1"].bool())
    assert torch.allclose(aabb1.cpu(), g["aabb1"], atol=1e-6)
    m2, aabb2, raw2 = march.update_alpha_mask(f, inv_s, grid=(40, 40, 40), thres=thres, mul_length=mul, prev=m1, return_alpha=True)
    parity(raw2.cpu(), g["alpha_40_masked"], label="update_alpha_mask_golden:311")
    assert torch.equal(m2.volume.cpu().bool(), g["mask2"].bool())
    assert torch.allclose(aabb2.cpu(), g["aabb2"], atol=1e-6)
    assert 0.1 < float(m2.volume.float().mean()) < 0.4


def test_trace_sdf_with_mesh(golden, dev):
    from tensoflow_amd import ops, surface
    from tensoflow_amd.march import SdfField
    g = golden("refine_r32")
    f = SdfField(g.sd, AABB, [32, 32, 32], 3, device=dev)
    bvh = ops.Bvh(g["verts"].numpy(), g["faces"].numpy(), dev)
    inters, normals, depth, hit = surface.trace_sdf_with_mesh(bvh, f, g["rays_o"].to(dev), g["rays_d"].to(dev), float(g["inv_s"]),
                                                             float(g["unit_size"]))
    assert torch.equal(hit.cpu(), g["hit"].bool())
    parity(depth.cpu(), g["depth"], label="trace_sdf_with_mesh:326.0")
    parity(inters.cpu(), g["inters"], label="trace_sdf_with_mesh:326.1")
    parity(normals.cpu(), g["normals"], tol=5e-4, label="trace_sdf_with_mesh:327")  # FD normal of a normalised difference of ~1e-3-sized sdf values


def test_render_frame_small(golden, dev):
    """Full-frame pipeline (BVH -> SDF refinement -> flow-sampled shading) on a 48x48 crop: finite, white background,
    and identical to shading the refined surface points directly."""
    from tensoflow_amd import surface
    from tensoflow_amd.march import SdfField
    from tensoflow_amd.shading import MCShader
    from tensoflow_amd.synth import pinhole_rays
    gs, gr = golden("shading_small"), golden("refine_r32")
    f = SdfField(gr.sd, AABB, [32, 32, 32], 3, device=dev)
    sh = MCShader(gs.sd, gr["verts"].numpy(), gr["faces"].numpy(), AABB, float(gr["unit_size"]), device=dev, n_fixed_diffuse=32)
    o, d, _, _ = [torch.from_numpy(a).to(dev) for a in pinhole_rays(48 * 48, seed=5, focal=2400.0)]
    out = surface.render_frame(sh, f, o, d, float(gr["inv_s"]), float(gr["unit_size"]), 16, 8, chunk=1000)
    assert torch.isfinite(out["color"]).all()
    assert 0.05 < float(out["hit"].float().mean()) < 0.95
    assert torch.equal(out["color"][~out["hit"]], torch.ones_like(out["color"][~out["hit"]]))
    inters, nrm, depth, hit = surface.trace_sdf_with_mesh(sh.bvh, f, o, d, float(gr["inv_s"]), float(gr["unit_size"]))
    idx = torch.nonzero(hit[:, 0])[:, 0]
    direct = sh.shade(inters[idx], -d[idx], nrm[idx], 16, 8)["colors"]
    parity(out["color"][idx].cpu(), direct.cpu(), abs_tol=1e-6, label="render_frame_small:348")


def test_tensoflow_backward_golden(golden, dev):
    """NIS-style loss -(w*logq).mean(): gradients of EVERY TensoFlow parameter vs the reference's autograd (golden)."""
    from tensoflow_amd.network.flow import TensoFlow
    g = golden("tensoflow_r32")
    m = TensoFlow(2, AABB, device=dev, gridSize=[32, 32, 32])
    m.load_state_dict(g.sd)
    c = lambda k: g[k].to(dev)
    z, logq = m(c("pts"), c("view_angles"), c("roughness"), c("x_rand"), return_jacobian=True)
    parity(logq.detach().cpu(), g["logq_rand"], label="tensoflow_backward_golden:359")
    (-(c("bwd_w") * logq).mean()).backward()
    checked = 0
    for name, p in m.named_parameters():
        if name in g.grad:
            assert p.grad is not None, name
            scale = float(g.grad[name].abs().max()) + 1e-12
            err = float((p.grad.cpu() - g.grad[name]).abs().max()) / scale
            assert err < 1e-4, (name, err)             # measured worst 4.2e-5 (float-atomic sums over 768 rows)
            checked += 1
    assert checked >= 20
    # rays_id form (rows of different points inside one tile: per-lane atomics path)
    m.zero_grad()
    z, logq = m(c("pts"), c("view_angles"), c("roughness"), c("x_rid"), return_jacobian=True, rays_id=c("rays_id"))
    logq.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_mcshading_training_step_golden(golden, dev):
    """Material-stage training direction: loss = sum(colors*w) + loss_nis; gradients of every trainable tensor
    (material planes/lines, predictors, inner-light net, environment map, both trainable flows) vs the reference autograd."""
    from tensoflow_amd.network.fields import MCShadingNetwork
    g = golden("shading_grad")
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=n_fd, specular_sample_num=n_fs, nis_diffuse_sample_num=sn_d,
               nis_specular_sample_num=sn_s, outer_light_version="envlight")
    m = MCShadingNetwork(cfg, (g["verts"].numpy(), g["faces"].numpy()), AABB, float(g["unit_size"]))
    missing, _ = m.load_state_dict(g.sd, strict=False)
    assert not missing
    for fl in (m.flow_diffuse_copy, m.flow_specular_copy):
        for p in fl.parameters():
            p.requires_grad = False
    m.eval()
    m.use_flow_diffuse_copy = m.use_flow_specular_copy = True          # as after update_step(999): the flow copies sample
    colors, out = m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), None, 600, False)
    parity(colors.detach().cpu(), g["colors"], label="mcshading_training_step_golden:394")
    assert abs(float(out["loss_nis_diffuse"]) - float(g["loss_nis_diffuse"])) < 1e-4 * max(1, abs(float(g["loss_nis_diffuse"])))
    assert abs(float(out["loss_nis_specular"]) - float(g["loss_nis_specular"])) < 1e-4 * max(1, abs(float(g["loss_nis_specular"])))
    for k in ("diffuse_light", "specular_light", "diffuse_color", "specular_color", "visibility", "indirect_light"):
        parity(out[k].detach().cpu(), g["out600/" + k], label=f"MCShadingNetwork training outputs {k}")
    check_aux(out, lambda k: g["out600/" + k])             # `variance` is what the reference trainer's progress line reads (trainer_inv.py:299)
    ((colors * g["bwd_w"].to(dev)).sum() + out["loss_nis"]).backward()
    worst, checked = 0.0, 0
    for name, p in m.named_parameters():
        if name in g.grad and p.requires_grad:
            assert p.grad is not None, name
            scale = float(g.grad[name].abs().max()) + 1e-12
            err = float((p.grad.cpu() - g.grad[name]).abs().max()) / scale
            l2 = float((p.grad.cpu() - g.grad[name]).norm() / (g.grad[name].norm() + 1e-20))
            worst = max(worst, l2)
            if "inner_light" in name:
                # ReLU net on ~900 hit rays: a single activation whose pre-activation is within 1e-6 of zero flips between the
                # CPU (MKL) and GPU arithmetic and moves that unit's gradient by ~1/#rays -> bound the L2 error, not the max
                assert l2 < 5e-3 and err < 3e-2, (name, err, l2)
            else:
                assert err < 5e-4 and l2 < 5e-4, (name, err, l2)
            checked += 1
    print("checked", checked, "worst", worst)
    assert checked >= 80


@pytest.mark.parametrize("step", [100, 600])
def test_mcshading_training_step_before_flow_copies_golden(golden, dev, step):
    """The material stage's first 1000 steps (use_flow_*_copy False, fields.py:1050-1065,1082,1160): both lobes on the fixed
    samplers, the specular directions warped by the predicted roughness.  Colours, NIS losses (fitted on the fixed samples from
    nis_loss_iter on) and the gradient of every trainable tensor against the reference's autograd (golden shading_grad_fixed,
    state and mesh of shading_grad)."""
    from tensoflow_amd.network.fields import MCShadingNetwork
    g, base = golden("shading_grad_fixed"), golden("shading_grad")
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=n_fd, specular_sample_num=n_fs, nis_diffuse_sample_num=sn_d,
               nis_specular_sample_num=sn_s, outer_light_version="envlight")
    m = MCShadingNetwork(cfg, (base["verts"].numpy(), base["faces"].numpy()), AABB, float(g["unit_size"]))
    m.load_state_dict(base.sd, strict=False)
    for fl in (m.flow_diffuse_copy, m.flow_specular_copy):
        for p in fl.parameters():
            p.requires_grad = False
    m.eval()
    assert not m.use_flow_diffuse_copy and not m.use_flow_specular_copy
    colors, out = m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), None, step, False)
    parity(colors.detach().cpu(), g[f"colors_{step}"], label="mcshading_training_step_before_flow_copies_golden:439")
    for k in ("loss_nis_diffuse", "loss_nis_specular"):
        ref = float(g[f"{k}_{step}"])
        assert abs(float(out[k]) - ref) < 1e-4 * max(1, abs(ref)), (k, float(out[k]), ref)
    for k in ("diffuse_light", "specular_light", "diffuse_color", "specular_color", "visibility", "indirect_light"):
        parity(out[k].detach().cpu(), g[f"out{step}/" + k], label=f"MCShadingNetwork training outputs (step {step}) {k}")
    check_aux(out, lambda k: g[f"out{step}/" + k])
    ((colors * g["bwd_w"].to(dev)).sum() + out["loss_nis"]).backward()
    grads = {k[len(f"grad{step}/"):]: v for k, v in g.a.items() if k.startswith(f"grad{step}/")}
    checked, bad = 0, []
    for name, p in m.named_parameters():
        if name in grads and p.requires_grad:
            assert p.grad is not None, name
            ref = grads[name]
            scale = float(ref.abs().max()) + 1e-12
            err = float((p.grad.cpu() - ref).abs().max()) / scale
            l2 = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-20))
            if "inner_light" in name:
                ok = l2 < 2e-2 and err < 3e-2       # ReLU flips on ~300 hit rays (see the flow-pass test; fewer rays here)
            elif name.startswith("flow_"):
                # 640 + 320 fixed samples, several of them ON a knot of the piecewise-quadratic spline (the cosine set's half angles
                # are a lattice): which side a sample falls on differs between CPU and GPU arithmetic and moves 1 / 640 of the gradient
                ok = l2 < 1e-2 and err < 3e-2
            else:
                ok = err < 1e-3 and l2 < 1e-3
            if not ok:
                bad.append((name, round(err, 5), round(l2, 5)))
            checked += 1
    assert not bad, bad
    assert checked >= (80 if step >= 500 else 30), checked
    with_grad = {n for n, p in m.named_parameters() if p.grad is not None}
    assert with_grad >= set(grads), sorted(set(grads) - with_grad)[:5]


def _direction_net(golden, dev):
    from tensoflow_amd.network.fields import MCShadingNetwork
    g, base = golden("shading_direction"), golden("shading_grad")
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=n_fd, specular_sample_num=n_fs, nis_diffuse_sample_num=sn_d,
               nis_specular_sample_num=sn_s, outer_light_version="direction")
    m = MCShadingNetwork(cfg, (base["verts"].numpy(), base["faces"].numpy()), AABB, float(g["unit_size"]))
    sd = {k: v for k, v in base.sd.items() if not k.startswith("outer_light.")}
    sd.update(g.sd)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not [k for k in unexpected if k.startswith("outer_light.")]     # the reference's keys of the 'direction' net, all of them
    for fl in (m.flow_diffuse_copy, m.flow_specular_copy):
        for p in fl.parameters():
            p.requires_grad = False
    m.eval()
    return g, base, sd, m


def test_direction_outer_light_eval_golden(golden, dev):
    """outer_light_version='direction' (configs/mat/syn/{lego,armadillo,horse}.yaml; fields.py:716-718, 913-916): get_lights,
    predict_outer_lights_pts and the eval forward (fixed pass + flow pass) against the reference run on the same state."""
    from tensoflow_amd.shading import MCShader
    g, base, sd, m = _direction_net(golden, dev)
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    sh = MCShader(sd, base["verts"].numpy(), base["faces"].numpy(), AABB, float(g["unit_size"]), device=dev, n_fixed_diffuse=n_fd,
                  n_fixed_specular=n_fs)
    assert sh.env is None and sh.outer is not None
    pts16 = g["pts"].repeat_interleave(16, 0).to(dev)
    lights, hit, _ = sh.lights(pts16, g["gl_dirs"].to(dev).contiguous())
    assert torch.equal(hit.cpu(), g["gl_hit"].bool())
    # Per RAY the reference's own fp32 answer is only good to ~6e-4 on this (trained) net: the IDE's degree-16 polynomials cancel
    # catastrophically in fp32 (oracle in fp32 vs the same oracle in fp64).  The HIP path is held to the fp64 answer within twice
    # the reference's own distance from it -- and to the reference within 1e-4 per PIXEL below, where the integral averages the noise.
    from oracle import shading as osh
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = osh.outer_light_direction(sd64, g["gl_dirs"].double())
    rel = lambda a: float(((a.double() - ref64).abs() / ref64.abs().clamp_min(1e-2)).max())
    with torch.no_grad():
        env = m.predict_outer_lights_pts(g["gl_dirs"].to(dev))                    # the differentiable composition (exact-fp32 products)
    miss = ~hit.cpu()
    e_ref, e_kernel, e_comp = rel(g["outer_pts"]), rel(torch.where(miss[:, None], lights.cpu(), ref64.float())), rel(env.cpu())
    print(f"direction outer light, per-ray error against the fp64 oracle: reference {e_ref:.2e}, staggered kernel {e_kernel:.2e}, "
          f"composed {e_comp:.2e} ({int(miss.sum())} missing + {int(hit.sum())} hit rays)")
    assert e_kernel < max(TOL, 2 * e_ref) and e_comp < max(TOL, 2 * e_ref)
    # both forms of the staggered kernel (MCShader's default rounds the activations to f16 once per layer: 128-ray form; every operand
    # split: 64-ray form) are held to that bar
    from tensoflow_amd import ops
    keep_ip = sh.inner_precision
    for ip in (ops.PREC_F16X3, ops.PREC_F16X2):
        sh.inner_precision = ip
        l2, h2, _ = sh.lights(pts16, g["gl_dirs"].to(dev).contiguous())
        e2 = rel(torch.where(miss[:, None], l2.cpu(), ref64.float()))
        print(f"  inner_precision {ip}: per-ray error of the outer net against fp64 {e2:.2e}")
        assert torch.equal(h2.cpu(), g["gl_hit"].bool()) and e2 < max(TOL, 2 * e_ref), (ip, e2, e_ref)
    sh.inner_precision = keep_ip
    hl = lights.cpu()[~miss]
    assert float(((hl - g["gl_lights"][~miss]).abs() / g["gl_lights"][~miss].abs().clamp_min(1e-2)).max()) < 1e-3     # inner light on the hits
    with torch.no_grad():
        colors, out = m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), None, None, False)
    parity(colors.cpu(), g["colors"], label="direction_outer_light_eval_golden:532")
    for k in ("diffuse_light", "specular_light", "diffuse_color", "specular_color", "visibility", "indirect_light", "rgb_pr_nis",
              "diffuse_color_nis", "specular_color_nis", "visibility_nis", "indirect_light_nis", "diffuse_light_nis", "specular_light_nis"):
        parity(out[k].cpu(), g.out[k], label=f"direction outer light outputs {k}")
    # the throughput path (zero-weight rays culled: their rows are never evaluated) gives the same colours
    sh.cull_dead_rays = True
    o2 = sh.shade(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), sn_d, sn_s)
    parity(o2["colors"].cpu(), g.out["rgb_pr_nis"], label="direction_outer_light_eval_golden:539")
    # lat-long image of the learned light (fields.py:1475-1510) goes through the same net
    img = m.env_light(8, 16)
    assert img.shape == (8, 16, 3) and torch.isfinite(img).all()


@pytest.mark.parametrize("step", [100, 1200])
def test_direction_outer_light_training_golden(golden, dev, step):
    """Training direction of the 'direction' outer light: the fixed-sampler pass (step 100: the colour gradient reaches the
    material grids through the roughness-warped directions AND the IDE of the outer net's input) and the flow pass (step 1200)
    against the reference's autograd."""
    g, base, sd, m = _direction_net(golden, dev)
    if step >= 1000:
        m.use_flow_diffuse_copy = m.use_flow_specular_copy = True
    colors, out = m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), None, step, False)
    parity(colors.detach().cpu(), g[f"colors_{step}"], label="direction_outer_light_training_golden:554")
    check_aux(out, lambda k: g[f"out{step}/" + k])
    if step >= 1000:
        assert abs(float(out["loss_nis"]) - float(g["loss_nis_1200"])) < 1e-4 * max(1, abs(float(g["loss_nis_1200"])))
    ((colors * g["bwd_w"].to(dev)).sum() + out["loss_nis"]).backward()
    grads = {k[len(f"grad{step}/"):]: v for k, v in g.a.items() if k.startswith(f"grad{step}/")}
    checked, bad, vs64 = 0, [], []
    for name, p in m.named_parameters():
        if name in grads and p.requires_grad:
            assert p.grad is not None, name
            ref = grads[name]
            scale = float(ref.abs().max()) + 1e-12
            err = float((p.grad.cpu() - ref).abs().max()) / scale
            l2 = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-20))
            key64 = f"grad{step}_f64/{name}"
            if key64 in g.a:
                # the reference's own fp32 gradient is 1e-3 away from the same pass run in fp64 (the colour gradient reaches the grids
                # through d IDE / d direction: degree-16 polynomials that cancel badly in fp32; tools/gen_golden.py stores both): the
                # HIP path is held to the fp64 answer within 2.5x the reference's own distance from it
                r64 = g[key64]
                d_ref = float((ref.double() - r64).abs().max() / r64.abs().max()), float((ref.double() - r64).norm() / r64.norm())
                d_hip = float((p.grad.cpu().double() - r64).abs().max() / r64.abs().max()), float((p.grad.cpu().double() - r64).norm() / r64.norm())
                ok = d_hip[0] < max(1e-3, 2.5 * d_ref[0]) and d_hip[1] < max(1e-3, 2.5 * d_ref[1])
                vs64.append((name.split(".parametrizations")[0], "ref %.1e/%.1e" % d_ref, "hip %.1e/%.1e" % d_hip))
            elif "inner_light" in name or "outer_light" in name:
                ok = l2 < 2e-2 and err < 6e-2       # ReLU flips on a few hundred rays (see test_mcshading_training_step_golden)
            else:
                # step 100: through the outer net's input gradient (see above; the reference's fp32 noise there is 1.6e-3); step 1200: no
                # gradient flows through the directions
                ok = (err < 5e-3 and l2 < 5e-3) if step < 1000 else (err < 1e-3 and l2 < 1e-3)
            if not ok:
                bad.append((name, round(err, 5), round(l2, 5)))
            checked += 1
    print("checked", checked, "against the fp64 reference (max / l2):", vs64)
    assert not bad, bad
    assert checked >= 40 and sum(n.startswith("outer_light.") for n in grads) == 12 and (step >= 1000 or len(vs64) >= 6)
    assert all(m.get_parameter(n).grad is not None for n in grads if n.startswith("outer_light."))


def _custom_net(golden, dev):
    from tensoflow_amd.network.fields import MCShadingNetwork
    g, base = golden("shading_custom"), golden("shading_grad")
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=n_fd, specular_sample_num=n_fs, nis_diffuse_sample_num=sn_d,
               nis_specular_sample_num=sn_s, outer_light_version="sphere_direction", human_lights=True)
    m = MCShadingNetwork(cfg, (base["verts"].numpy(), base["faces"].numpy()), AABB, float(g["unit_size"]))
    sd = {k: v for k, v in base.sd.items() if not k.startswith("outer_light.")}
    sd.update(g.sd)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not [k for k in unexpected if k.startswith(("outer_light.", "human_light."))]
    for fl in (m.flow_diffuse_copy, m.flow_specular_copy):
        for p in fl.parameters():
            p.requires_grad = False
    m.eval()
    return g, base, sd, m


def test_sphere_direction_and_human_lights_eval_golden(golden, dev):
    """configs/mat/custom/*.yaml: outer_light_version='sphere_direction' + human_lights=True.  The miss branch is a composition
    (encodings in torch, dense layers on tf_linear_fwd) inside the otherwise fused eval path; per pixel against the reference run."""
    g, base, sd, m = _custom_net(golden, dev)
    poses = g["human_poses"].to(dev)
    with torch.no_grad():
        colors, out = m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), poses, None, False)
    parity(colors.cpu(), g["colors"], label="sphere_direction_and_human_lights_eval_golden:618")
    for k in ("diffuse_light", "specular_light", "diffuse_color", "specular_color", "visibility", "indirect_light", "rgb_pr_nis",
              "diffuse_color_nis", "specular_color_nis", "visibility_nis", "indirect_light_nis", "diffuse_light_nis", "specular_light_nis"):
        parity(out[k].cpu(), g.out[k], label=f"sphere_direction outputs {k}")
    for sfx in ("", "_nis"):
        check_aux(out, lambda k: g.out[k], sfx, human_expected=True)       # human_lights * human_weights of the missing specular rays
    img = m.env_light(8, 16)                          # predict_outer_lights_pts('sphere_direction') feeds the direction's IDE twice (:1515-1516)
    assert img.shape == (8, 16, 3) and torch.isfinite(img).all() and float(img.std()) > 0
    sh = m.shader()
    lights, hit, _ = sh.lights(g["pts"].repeat_interleave(16, 0).to(dev), g["gl_dirs"].to(dev).contiguous())       # no poses: outer net alone
    assert torch.equal(hit.cpu(), g["gl_hit"].bool())
    with pytest.raises(ValueError, match="human_poses"):
        m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), None, None, False)


@pytest.mark.parametrize("step", [100, 1200])
def test_sphere_direction_and_human_lights_training_golden(golden, dev, step):
    g, base, sd, m = _custom_net(golden, dev)
    if step >= 1000:
        m.use_flow_diffuse_copy = m.use_flow_specular_copy = True
    colors, out = m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), g["human_poses"].to(dev), step, False)
    parity(colors.detach().cpu(), g[f"colors_{step}"], label="sphere_direction_and_human_lights_training_golden:639")
    check_aux(out, lambda k: g[f"out{step}/" + k], human_expected=True)
    ((colors * g["bwd_w"].to(dev)).sum() + out["loss_nis"]).backward()
    grads = {k[len(f"grad{step}/"):]: v for k, v in g.a.items() if k.startswith(f"grad{step}/")}
    checked, bad = 0, []
    for name, p in m.named_parameters():
        if name in grads and p.requires_grad:
            assert p.grad is not None, name
            ref = grads[name]
            err = float((p.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
            l2 = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-20))
            # nets: ReLU flips on a few hundred rays; grids (step 100): the reference's own fp32 noise through d IDE / d direction
            # (see test_direction_outer_light_training_golden: 1-2e-3 between its fp32 and fp64 runs)
            ok = (l2 < 2e-2 and err < 6e-2) if ("outer_light" in name or "human_light" in name) else (err < 5e-3 and l2 < 5e-3)
            if not ok:
                bad.append((name, round(err, 5), round(l2, 5)))
            checked += 1
    assert not bad, bad
    assert checked >= 24 and sum(n.startswith("human_light.") for n in grads) == 12


def test_sdf_alpha_training_golden(golden, dev):
    """Shape-stage training direction (geometry): loss over compute_sdf_alpha + compositing; gradients of the SDF planes, lines,
    decoder and variance vs the reference autograd."""
    from tensoflow_amd.autograd import CompositeFn, SdfAlphaFn
    g = golden("march_grad")
    P = lambda k: g.sd["sdf_network." + k].to(dev).requires_grad_(True)
    params = [P(f"sdf_plane.{i}") for i in range(3)] + [P(f"sdf_line.{i}") for i in range(3)] + \
             [P("sdf_mat.0.weight"), P("sdf_mat.0.bias"), P("sdf_mat.2.weight"), P("sdf_mat.2.bias")]
    var = g.sd["deviation_network.variance"].to(dev).requires_grad_(True)
    inv_s = torch.exp(var * 10.0)
    c = lambda k: g[k].to(dev)
    units = [2.0 / 31] * 3
    rn = int(g["n_rays"])
    alpha, grad, feat, sdf, nh = SdfAlphaFn.apply(c("pts"), c("level")[:, 0].contiguous(), c("dists"), c("dirs"), inv_s, 0.5, AABB, units,
                                                  3, *params)
    parity(alpha.detach().cpu(), g["alpha"], label="sdf_alpha_training_golden:675")
    vals = torch.cat([grad, feat[:, :8]], -1).contiguous()
    w, acc, out = CompositeFn.apply(alpha, vals, c("ray_indices"), rn)
    inv_vec = inv_s.expand(alpha.shape[0]).clip(1e-6, 1e6)
    loss = (acc[:, None] * c("wa")).sum() + (out[:, :3] * c("wn")).sum() + (out[:, 3:] * c("wf")).sum() \
        + 0.1 * ((grad.norm(dim=-1) - 1.0) ** 2).mean() + 0.01 * nh.abs().mean() + torch.exp(-20.0 * sdf.abs()).mean() \
        + torch.mean(1 / inv_vec)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    loss.backward()
    names = [f"sdf_network.sdf_plane.{i}" for i in range(3)] + [f"sdf_network.sdf_line.{i}" for i in range(3)] + \
            ["sdf_network.sdf_mat.0.weight", "sdf_network.sdf_mat.0.bias", "sdf_network.sdf_mat.2.weight", "sdf_network.sdf_mat.2.bias"]
    for n, p in zip(names + ["deviation_network.variance"], params + [var]):
        ref = g.grad[n]
        l2 = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-20))
        assert l2 < 1e-3, (n, l2)


def _occupancy(seed=3, n=64):
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij")
    vol = ((x ** 2 + y ** 2 + z ** 2 < 0.55 ** 2) | (rng.random((n, n, n)) < 0.002)).astype(np.uint8)
    return torch.from_numpy(vol)


def test_alpha_mask_sample_bit_exact(dev):
    """A8: AlphaGridMask.sample_alpha > 0, incl. points outside the aabb and exactly on voxel planes."""
    from oracle import march as om
    from tensoflow_amd import ops
    vol = _occupancy()
    aabb = AABB.clone()
    rng = np.random.default_rng(4)
    pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (200000, 3)).astype(np.float32))
    lattice = torch.from_numpy(np.linspace(-1, 1, 64).astype(np.float32))
    pts[:4096, 0] = lattice[torch.from_numpy(rng.integers(0, 64, 4096))]            # exact grid planes: zero weights
    pts[2048:6144, 1] = lattice[torch.from_numpy(rng.integers(0, 64, 4096))]
    ref = om.alpha_mask_sample(vol, aabb, pts) > 0
    got = ops.alpha_mask_sample(vol.to(dev), aabb, pts.to(dev)).cpu()
    assert torch.equal(got, ref)
    assert 0.02 < ref.float().mean() < 0.5


@pytest.mark.parametrize("mode", ["uniform", "uniform_mask", "fixed_step_mask"])
def test_march_uniform_matches_oracle(dev, mode):
    """A7': fixed-step sampler with per-wavefront compaction: packed (ray, t) order and integer ray indices bit-exact."""
    from oracle import march as om
    from tensoflow_amd import ops
    from tensoflow_amd.synth import pinhole_rays
    o, d, _, _ = [torch.from_numpy(a) for a in pinhole_rays(3000, seed=9)]
    d[5] = torch.tensor([0.0, 0.0, -1.0]); o[5] = torch.tensor([0.1, 0.2, 2.0])      # zero direction components
    o[6] = torch.tensor([5.0, 5.0, 5.0])                                              # misses the box
    near, far = om.near_far_from_sphere(o, d)
    aabb = AABB.clone()
    vol = None if mode == "uniform" else _occupancy()
    n_steps, step = (256, 0.0) if mode != "fixed_step_mask" else (400, 0.011)
    rt0, rt1, rr = om.march_uniform(o, d, near, far, aabb, n_steps, step, vol)
    t0, t1, ridx = ops.march_uniform(o.to(dev), d.to(dev), near.to(dev), far.to(dev), aabb, n_steps, step,
                                     None if vol is None else vol.to(dev))
    assert ridx.dtype == torch.int64 and torch.equal(ridx.cpu(), rr)
    assert torch.equal(t0.cpu(), rt0) and torch.equal(t1.cpu(), rt1)
    assert rr.numel() > 10000


def test_sampling_algebra_kernels_equal_the_composition(dev):
    """tf_sample_ray_intervals / tf_sample_points (round 5: sample_ray's tail and render_core's prelude, one launch each) against the
    element-wise composition they replace (shapeRenderer.py:921-932, :1118-1131): intervals and the inside-the-box mask bit for bit,
    points / levels to fp32 rounding."""
    from tensoflow_amd import march, ops
    gen = torch.Generator().manual_seed(31)
    rn, S = 777, 41
    o = (torch.randn(rn, 3, generator=gen) * 0.4).to(dev)
    d = torch.nn.functional.normalize(torch.randn(rn, 3, generator=gen), dim=-1).to(dev)
    t = torch.sort(torch.rand(rn, S, generator=gen) * 3.0, dim=-1).values.to(dev)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    t0, t1, inner = ops.sample_ray_intervals(o, d, t, aabb)
    dist = t[:, 1:] - t[:, :-1]
    dist = torch.cat([dist, dist[:, -1:]], -1)
    p = o[:, None] + d[:, None] * (t + dist * 0.5)[..., None]
    ref_in = ~((aabb[0].to(dev) > p) | (p > aabb[1].to(dev))).any(-1)
    assert torch.equal(t0, t.reshape(-1)) and torch.equal(t1, (t + dist).reshape(-1)) and torch.equal(inner.bool(), ref_in.reshape(-1))
    assert 0.1 < float(inner.float().mean()) < 0.9
    keep = torch.nonzero(inner)[:, 0]
    ridx = torch.div(keep, S, rounding_mode="floor")
    radii = (torch.rand(rn, 1, generator=gen) * 2e-3 + 1e-4).to(dev)
    cos = (torch.rand(rn, 1, generator=gen) * 0.3 + 0.7).to(dev)
    a, b = t0[keep], t1[keep]
    mid, dists, vd, pts, lv = ops.sample_points(o, d, radii, cos, ridx, a, b, 2.0 / 300 / 2)
    m_ref = (a + b) * 0.5
    assert torch.equal(mid, m_ref) and torch.equal(dists, b - a) and torch.equal(vd, d[ridx])
    assert torch.equal(pts, o[ridx] + d[ridx] * m_ref[:, None])
    lv_ref = torch.log2(march.ball_radii(m_ref[:, None], radii[ridx], cos[ridx]) / (2.0 / 300 / 2))
    assert float((lv - lv_ref).abs().max()) < 2e-6 * float(lv_ref.abs().max())
