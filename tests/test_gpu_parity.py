"""GPU parity: HIP kernels (through the C ABI, tensoflow_amd.ops) vs the CPU oracle and the golden
vectors generated from the imported reference.  Tolerance: 1e-4 relative fp32 (north_star);
integer / boolean outputs bit-exact.  Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

from conftest import AABB, in_fp64, parity, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test collected but no GPU is visible")
    return torch.device("cuda:0")


def _planes(sd, pfx, dev):
    return [sd[f"{pfx}plane.{i}"].to(dev) for i in range(3)], [sd[f"{pfx}line.{i}"].to(dev) for i in range(3)]


# ------------------------------------------------------------------------------- VM field
@pytest.mark.parametrize("tag", ["r32_l1", "r32_l3", "r24x32x40_l3"])
def test_vm_gather_matches_oracle(golden, dev, tag):
    from oracle import vm_field as ovm
    from tensoflow_amd import ops
    g = golden("tensosdf_" + tag)
    nl = int(g["n_levels"])
    planes, lines = _planes(g.sd, "sdf_", dev)
    packed = ops.VmPacked(planes, lines, nl)
    for level in (None, g["level"]):
        feat = ops.vm_gather(packed, g["pts"].to(dev), None if level is None else level.to(dev), AABB)
        pf, lf = ovm.vm_gather([p.cpu() for p in planes], [l.cpu() for l in lines],
                               ovm.contraction(g["pts"], AABB), None if level is None else level[:, 0], nl)
        t64 = in_fp64(lambda P, L, x, lv: torch.mul(*ovm.vm_gather(P, L, ovm.contraction(x, AABB.double()), lv, nl)),
                      planes, lines, g["pts"], None if level is None else level[:, 0])
        parity(feat.cpu(), pf * lf, truth=t64, label="vm_gather")


def test_vm_pack_roundtrip_and_backward(golden, dev):
    """pack -> gather -> backward through (gather_bwd, pack_bwd) == autograd of the oracle."""
    from oracle import vm_field as ovm
    from tensoflow_amd import ops
    g = golden("tensosdf_r32_l3")
    planes, lines = _planes(g.sd, "sdf_", dev)
    packed = ops.VmPacked(planes, lines, 3)
    pts, level = g["pts"], g["level"]
    gen = torch.Generator().manual_seed(0)
    gfeat = torch.randn(pts.shape[0], 108, generator=gen)
    gp = ops.vm_gather_bwd(packed, pts.to(dev), level.to(dev), AABB, gfeat.to(dev))
    gplanes, glines = packed.unpack_grad(gp, planes, lines)
    pl = [p.cpu().clone().requires_grad_(True) for p in planes]
    ln = [l.cpu().clone().requires_grad_(True) for l in lines]
    pf, lf = ovm.vm_gather(pl, ln, ovm.contraction(pts, AABB), level[:, 0], 3)
    ((pf * lf) * gfeat).sum().backward()
    for i in range(3):
        parity(gplanes[i].cpu(), pl[i].grad, label="vm_pack_roundtrip_and_backward:58")
        parity(glines[i].cpu(), ln[i].grad, label="vm_pack_roundtrip_and_backward:59")


# ------------------------------------------------------------------------------- SDF
@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f16x3"])
@pytest.mark.parametrize("tag", ["r32_l1", "r32_l3", "r24x32x40_l3"])
def test_sdf_forward_golden(golden, dev, tag, prec):
    from oracle import vm_field as ovm
    from tensoflow_amd import ops
    g = golden("tensosdf_" + tag)
    nl = int(g["n_levels"])
    planes, lines = _planes(g.sd, "sdf_", dev)
    packed = ops.VmPacked(planes, lines, nl)
    W = [g.sd[k].to(dev) for k in ("sdf_mat.0.weight", "sdf_mat.0.bias", "sdf_mat.2.weight", "sdf_mat.2.bias")]
    for level, ref in ((None, g["out_none"]), (g["level"], g["out_lvl"])):
        sdf, feat = ops.sdf_forward(packed, *W, g["pts"].to(dev), None if level is None else level.to(dev), AABB, precision=prec)
        t64 = in_fp64(ovm.sdf_forward, g.sd, g["pts"], level, AABB, nl)
        parity(sdf.cpu(), ref[:, 0], truth=t64[:, 0], label="sdf_forward sdf")
        parity(feat.cpu(), ref[:, 1:], truth=t64[:, 1:], label="sdf_forward features")
        sdf_only, none = ops.sdf_forward(packed, *W, g["pts"].to(dev), None if level is None else level.to(dev), AABB,
                                         want_feat=False, precision=prec)
        assert none is None and torch.equal(sdf_only, sdf)


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f16x3"])
def test_sdf_alpha_golden(golden, dev, prec):
    """alpha, the finite-difference gradient and the hessian term are DIFFERENCES of O(1) decoder outputs: a sample of alpha 1e-4 carries
    the ~1e-7 rounding of the sigmoids it is the difference of.  The goldens are the reference's fp32 values; where one of them is
    beyond 1e-4 RELATIVE, the oracle evaluated in fp64 says how far the reference itself is from the function (conftest.parity)."""
    from oracle import march as omarch
    from tensoflow_amd import ops
    g = golden("march_r32")
    planes, lines = _planes(g.sd, "sdf_network.sdf_", dev)
    packed = ops.VmPacked(planes, lines, 3)
    W = [g.sd["sdf_network." + k].to(dev) for k in ("sdf_mat.0.weight", "sdf_mat.0.bias", "sdf_mat.2.weight", "sdf_mat.2.bias")]
    ridx = g["ray_indices"]
    dists = g["t_ends"] - g["t_starts"]
    units = (AABB[1] - AABB[0]) / (torch.tensor([32.0, 32.0, 32.0]) - 1)
    inv_s = float(torch.exp(g.sd["deviation_network.variance"] * 10.0))
    for ca in (0.0, 0.5, 1.0):
        alpha, grad, feat, sdf, nh = ops.sdf_alpha(packed, *W, g["sample_pts"].to(dev), g["sample_levels"].to(dev),
                                                   dists.to(dev), g["dirs"][ridx].to(dev), AABB, units, inv_s, ca, precision=prec)
        t64 = in_fp64(omarch.sdf_alpha, g.sd, g["sample_pts"], g["sample_levels"], dists, g["dirs"][ridx], ca, AABB, [32, 32, 32], 3)
        parity(alpha.cpu(), g[f"alpha_{ca}"], truth=t64[0], label=f"sdf_alpha alpha (cos anneal {ca})")
    parity(grad.cpu(), g["sa_grad"], truth=t64[1], label="sdf_alpha gradient")
    parity(feat.cpu(), g["sa_feat"], truth=t64[2], label="sdf_alpha features")
    parity(sdf.cpu(), g["sa_sdf"], truth=t64[4], label="sdf_alpha sdf")
    # second differences divide rounding noise by eps^2 = 4e-3: same looser bound as the oracle-vs-reference test
    # (measured 1.1e-3 with the exact-fp32 and with the f16x3 decoder: the reference's own rounding dominates)
    parity(nh.cpu(), g["sa_hess"], truth=t64[5], abs_tol=2e-3, label="sdf_alpha normal_hessian")


# ------------------------------------------------------------------------------- compositing
def test_composite_matches_oracle(dev):
    from oracle import segments as oseg
    from tensoflow_amd import ops
    gen = torch.Generator().manual_seed(3)
    n_rays = 257
    counts = torch.randint(0, 200, (n_rays,), generator=gen)
    counts[5] = 0
    counts[17] = 300                       # > 4 wave chunks
    ridx = torch.repeat_interleave(torch.arange(n_rays), counts)
    n = ridx.shape[0]
    alpha = torch.rand(n, generator=gen) ** 3
    alpha[::97] = 1.0
    alpha[1::89] = 0.0
    vals = torch.randn(n, 3, generator=gen)
    w, acc, out = ops.composite(alpha.to(dev), ridx.to(dev), vals.to(dev), n_rays)
    w_ref, _ = oseg.render_weight_from_alpha_seq(alpha, ridx)
    parity(w.cpu(), w_ref, abs_tol=1e-5, label="composite_matches_oracle:121")
    parity(acc.cpu(), oseg.accumulate_along_rays(w_ref, None, ridx, n_rays)[:, 0], abs_tol=1e-5, label="composite_matches_oracle:122")
    parity(out.cpu(), oseg.accumulate_along_rays(w_ref, vals, ridx, n_rays), abs_tol=1e-5, label="composite_matches_oracle:123")
    # empty input
    w0, acc0, out0 = ops.composite(alpha[:0].to(dev), ridx[:0].to(dev), vals[:0].to(dev), 4)
    assert acc0.abs().sum() == 0 and out0.abs().sum() == 0
    # backward vs autograd through the oracle scan
    a = alpha.clone().clamp(max=0.98).requires_grad_(True)
    v = vals.clone().requires_grad_(True)
    wr, _ = oseg.render_weight_from_alpha(a, ray_indices=ridx, n_rays=n_rays)
    g_acc = torch.randn(n_rays, generator=gen)
    g_out = torch.randn(n_rays, 3, generator=gen)
    (oseg.accumulate_along_rays(wr, None, ridx, n_rays)[:, 0] * g_acc).sum().backward(retain_graph=True)
    (oseg.accumulate_along_rays(wr, v, ridx, n_rays) * g_out).sum().backward()
    ad = a.detach().to(dev)
    w2, _, _ = ops.composite(ad, ridx.to(dev), vals.to(dev), n_rays)
    ga, gv = ops.composite_bwd(ad, ridx.to(dev), vals.to(dev), w2, g_acc.to(dev), g_out.to(dev), n_rays)
    parity(ga.cpu(), a.grad, tol=1e-4, label="composite_matches_oracle:138")
    parity(gv.cpu(), v.grad, abs_tol=1e-5, label="composite_matches_oracle:139")


# ------------------------------------------------------------------------------- flow
def _flow_weights(sd, pfx, dev):
    return [[(sd[f"{pfx}flows.{b}.nn.{l}.weight"].to(dev), sd[f"{pfx}flows.{b}.nn.{l}.bias"].to(dev)) for l in (1, 3, 5, 7)]
            for b in range(2)]


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f16x3"])
def test_flow_sample_and_logq_golden(golden, dev, prec):
    from oracle import flow as oflow
    from tensoflow_amd import ops
    from tensoflow_amd.shading import sphere_latent
    g = golden("tensoflow_r32")
    cond = oflow.flow_condition(g.sd, g["pts"], g["view_angles"], g["roughness"], AABB).to(dev)
    Wt = _flow_weights(g.sd, "", dev)
    for sn in (8, 32, 128):
        parity(sphere_latent(sn).clamp(1e-6, 1 - 1e-6), g[f"latent_{sn}"], abs_tol=1e-7, label="flow_sample_and_logq_golden:157")
        ang, logj, bins = ops.flow_sample(Wt, cond, sphere_latent(sn).to(dev), want_bins=True, precision=prec)
        # The reference's closed-form spline root (flow.py:479-493) loses digits when the quadratic
        # coefficient a = (v[e+1]-v[e])*w[e] is tiny: |d sol| ~ eps*b/|a|.  A ~1e-6 difference in the MLP
        # output (MKL vs MFMA summation order) can therefore move an isolated sample by > 1e-4 in the
        # reference's own arithmetic; the bound is 1e-4 for 99.9 % of the samples and 2e-3 for the rest.
        ea = (ang.cpu() - g[f"angles_{sn}"]).abs().flatten()
        el = ((logj.cpu() - g[f"logj_{sn}"]).abs() / g[f"logj_{sn}"].abs().clamp_min(1.0)).flatten()
        assert float(torch.quantile(ea, 0.999)) < TOL and float(ea.max()) < 2e-3
        assert float(torch.quantile(el, 0.999)) < TOL and float(el.max()) < 2e-3
        _, _, b0, b1 = oflow.flow_sample(g.sd, g["pts"], g["view_angles"], g["roughness"], sn, AABB, return_bins=True)
        assert torch.equal(bins[..., 0].cpu().long(), b0) and torch.equal(bins[..., 1].cpu().long(), b1)   # bit-exact
        z, logq, zb = ops.flow_logq(Wt, cond, g[f"angles_{sn}"].to(dev), want_bins=True, precision=prec)
        z64, lq64 = in_fp64(oflow.flow_logq, g.sd, g["pts"], g["view_angles"], g["roughness"], g[f"angles_{sn}"], AABB)[:2]
        parity(z.cpu(), g[f"z_{sn}"], truth=z64, label=f"flow_logq z ({sn} samples)")
        parity(logq.cpu(), g[f"logq_{sn}"], truth=lq64, label=f"flow_logq logq ({sn} samples)")
        _, _, b0, b1 = oflow.flow_logq(g.sd, g["pts"], g["view_angles"], g["roughness"], g[f"angles_{sn}"], AABB, return_bins=True)
        assert torch.equal(zb[..., 0].cpu().long(), b0) and torch.equal(zb[..., 1].cpu().long(), b1)
    z, logq = ops.flow_logq(Wt, cond, g["x_rand"].to(dev), precision=prec)
    z64, lq64 = in_fp64(oflow.flow_logq, g.sd, g["pts"], g["view_angles"], g["roughness"], g["x_rand"], AABB)[:2]
    parity(z.cpu(), g["z_rand"], truth=z64, label="flow_logq z (random x)")
    parity(logq.cpu(), g["logq_rand"], truth=lq64, label="flow_logq logq (random x)")
    z, logq = ops.flow_logq(Wt, cond, g["x_rid"].to(dev), rays_id=g["rays_id"].to(dev), precision=prec)
    z64, lq64 = in_fp64(oflow.flow_logq, g.sd, g["pts"], g["view_angles"], g["roughness"], g["x_rid"], AABB, rays_id=g["rays_id"])[:2]
    parity(z.cpu(), g["z_rid"], truth=z64, label="flow_logq z (rays_id)")
    parity(logq.cpu(), g["logq_rid"], truth=lq64, label="flow_logq logq (rays_id)")


def test_pwquad_spline_on_reference_vectors(golden, dev):
    """The device spline (the pw_inverse / pw_forward functions the fused flow kernels call, through tf_pwquad_eval) on the
    reference's own spline vectors, edge rows included (y -> 0 / 1, equal knots, w_tilde = -12 / +6, all-zero rows;
    tools/gen_golden.py:gen_pwquad <- network/flow.py:332-525).  Bins bit-exact against the oracle; density direction strict;
    sampling direction strict except rows the reference's closed-form root itself cannot hold in fp32 -- each such row is shown
    to be ill conditioned by evaluating the ORACLE in fp32 and fp64 on it."""
    from oracle import flow as oflow
    from tensoflow_amd import ops
    g = golden("pwquad")
    wv, y = g["wv"], g["y"]
    # density direction (flow_inv): strict
    out, lj, bins = ops.pwquad(wv.to(dev), y.to(dev), inverse=False)
    o_ref, lj_ref, b_ref = oflow.pwquad_forward(y, wv)
    assert torch.equal(bins.cpu().long(), b_ref.long())
    o64, lj64, _ = oflow.pwquad_forward(y.double(), wv.double())
    parity(out.cpu(), g["fwd_out"], truth=o64, label="pwquad forward value")
    parity(lj.cpu(), g["fwd_logj"], truth=lj64, label="pwquad forward log-Jacobian")
    # sampling direction (flow)
    x, ljx, binx = ops.pwquad(wv.to(dev), y.to(dev), inverse=True)
    x_ref, _, bx_ref = oflow.pwquad_inverse(y, wv)
    assert torch.equal(binx.cpu().long(), bx_ref.long())
    err = (x.cpu() - g["inv_x"]).abs()
    bad = (err > TOL).nonzero()[:, 0]
    x64, ljx64, _ = oflow.pwquad_inverse(y.double(), wv.double())
    spread = (x64 - g["inv_x"].double()).abs()                 # the reference's fp32 answer against the exact root
    print(f"pwquad inverse: {len(bad)} of {len(y)} rows beyond 1e-4 (max {float(err.max()):.2e}); their fp32-vs-fp64 spread in the "
          f"reference formula: {[f'{float(spread[i]):.1e}' for i in bad[:8]]}")
    assert len(bad) <= 8
    for i in bad.tolist():
        assert float(spread[i]) > 0.1 * float(err[i]), (i, float(err[i]), float(spread[i]))   # same order: ill conditioned in the reference itself
    ok = err <= TOL
    parity(ljx.cpu()[ok], g["inv_logj"][ok], truth=ljx64[ok], label="pwquad inverse log-Jacobian (rows whose root holds 1e-4)")
    assert float(err.max()) < 2e-2


@pytest.mark.parametrize("masked", [False, True])
def test_flow_logq_gradient_wrt_samples_matches_oracle_autograd(golden, dev, masked):
    """d logq / d x (asked for between nis_loss_iter and nis_start_iter: the fixed GGX half angles depend on the predicted roughness)
    from the fused backward kernel -- closed form through both splines and the kept coordinate's embedding -- against autograd of the
    oracle flow in fp64, on samples that include knots of narrow spline bins (slopes of 2e3)."""
    from oracle import flow as ofl
    from tensoflow_amd.network.flow import TensoFlow
    g = golden("tensoflow_r32")
    m = TensoFlow(2, AABB, device=dev, gridSize=[32, 32, 32])
    m.load_state_dict(g.sd)
    pts, va, rough = g["pts"], g["view_angles"], g["roughness"]
    pn, sn = pts.shape[0], 64
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(pn, sn, 2, generator=gen).clamp(1e-3, 1 - 1e-3)
    w = torch.randn(pn, sn, 1, generator=gen)
    rid = None
    if masked:                                   # the specular lobe's form: a ragged subset of the samples with their point ids
        keep = torch.rand(pn, sn, generator=gen) < 0.6
        rid = torch.arange(pn)[:, None].expand(pn, sn)[keep]
        x, w = x[keep], w[keep]
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in g.sd.items()}
    x64 = x.double().requires_grad_(True)
    _, lq = ofl.flow_logq(sd64, pts.double(), va.double(), rough.double(), x64, AABB.double(), rays_id=rid)
    (lq * w.double()).sum().backward()
    ref = x64.grad
    xd = x.to(dev).requires_grad_(True)
    _, lq2 = m(pts.to(dev), va.to(dev), rough.to(dev), xd, return_jacobian=True, rays_id=None if rid is None else rid.to(dev))
    (lq2 * w.to(dev)).sum().backward()
    err = (xd.grad.cpu().double() - ref).abs()
    scale = float(ref.abs().max())
    print(f"d logq / d x: max |g| {scale:.1f}, max err {float(err.max()):.2e}, median {float(err.median()):.2e}")
    assert scale > 100 and float(err.max()) < 1e-3 * scale and float(err.median()) < 1e-5 * scale


def test_flow_roundtrip_full_size(dev, golden):
    """size-independent property at BASELINE size (128 samples, 4096 points): logq(sample) == -logj."""
    from oracle import flow as oflow
    from tensoflow_amd import ops
    from tensoflow_amd.shading import sphere_latent
    g = golden("tensoflow_r32")
    gen = torch.Generator().manual_seed(5)
    pn = 4096
    pts = torch.rand(pn, 3, generator=gen) * 1.6 - 0.8
    va, rough = torch.rand(pn, 2, generator=gen), torch.rand(pn, 1, generator=gen)
    cond = oflow.flow_condition(g.sd, pts, va, rough, AABB).to(dev)
    Wt = _flow_weights(g.sd, "", dev)
    jit = torch.rand(pn, 128, generator=gen).to(dev)
    ang, logj = ops.flow_sample(Wt, cond, sphere_latent(128).to(dev), jitter=jit)
    assert torch.isfinite(ang).all() and ang.min() > 0 and ang.max() < 1
    z, logq = ops.flow_logq(Wt, cond, ang)
    assert float(torch.quantile((logq + logj).abs().flatten()[:1_000_000], 0.999)) < 1e-3


# ------------------------------------------------------------------------------- light, mesh, shading
def test_cube_lookup_golden(golden, dev):
    from oracle import texture as otex
    from tensoflow_amd import ops
    g = golden("shading_small")
    base = g.sd["outer_light.base"]
    got = ops.cube_lookup(base.to(dev), g["env_dirs"].to(dev))
    parity(got.cpu(), g["env_direct"], label="cube_lookup_golden:273")
    gen = torch.Generator().manual_seed(2)
    # directions hugging edges and corners of the cube
    d = torch.randn(20000, 3, generator=gen)
    d[:5000] = torch.sign(d[:5000]) * (1 + 0.02 * torch.randn(5000, 3, generator=gen))
    got = ops.cube_lookup(base.to(dev), d.to(dev), apply_exp=False)
    parity(got.cpu(), otex.cube_bilinear(base, d), label="cube_lookup_golden:279")
    gout = torch.randn(20000, 3, generator=gen)
    b = base.clone().requires_grad_(True)
    (torch.exp(otex.cube_bilinear(b, d)) * gout).sum().backward()
    gb = ops.cube_lookup_bwd(base.to(dev), d.to(dev), gout.to(dev), apply_exp=True)
    parity(gb.cpu(), b.grad, label="cube_lookup_golden:284")


def test_bvh_trace_matches_brute_force(golden, dev):
    from oracle import shading as osh
    from tensoflow_amd import ops
    from tensoflow_amd.synth import sphere_torus_mesh
    verts, faces = sphere_torus_mesh(n_lat=24, n_lon=48, n_major=64, n_minor=24)
    bvh = ops.Bvh(verts, faces, dev)
    tr = osh.MeshTracer(torch.from_numpy(verts)[torch.from_numpy(faces).long()])
    gen = torch.Generator().manual_seed(9)
    o = torch.randn(6000, 3, generator=gen) * 0.6
    d = torch.nn.functional.normalize(torch.randn(6000, 3, generator=gen), dim=-1)
    pos, nrm, depth, hit = bvh.trace(o.to(dev), d.to(dev), dynamic=True)
    pos2, nrm2, depth2, hit2 = bvh.trace(o.to(dev), d.to(dev), dynamic=False)
    # same algorithm, different instruction selection (FMA contraction): flags identical, values to the last bits
    assert torch.equal(hit, hit2) and rel_err(depth, depth2) < 1e-6 and rel_err(pos, pos2) < 1e-6
    # hit_rows_only: identical depth / hit, identical pos / nrm rows where the ray hits (the other rows are not written)
    pos3, nrm3, depth3, hit3 = bvh.trace(o.to(dev), d.to(dev), hit_rows_only=True)
    assert torch.equal(hit3, hit) and torch.equal(depth3, depth)
    assert torch.equal(pos3[hit], pos[hit]) and torch.equal(nrm3[hit], nrm[hit])
    rpos, rnrm, rdepth, rhit = tr(o, d)
    assert torch.equal(hit.cpu(), rhit)                     # boolean, bit-exact
    assert 0.2 < rhit.float().mean() < 0.98
    parity(depth.cpu(), rdepth[:, 0], abs_tol=1e-5, label="bvh_trace_matches_brute_force:308")
    parity(pos.cpu(), rpos, abs_tol=1e-5, label="bvh_trace_matches_brute_force:309")
    # normal of either adjacent face is acceptable on shared edges: compare where depths are not tied
    assert float((nrm.cpu() - rnrm).norm(dim=-1).gt(1e-4).float().mean()) < 2e-3


def test_inner_light_matches_oracle(golden, dev):
    from oracle import shading as osh
    from tensoflow_amd import ops
    from tensoflow_amd.shading import wn_weight
    g = golden("shading_small")
    gen = torch.Generator().manual_seed(4)
    m = 1000                                    # not a multiple of 32: exercises the ragged last tile
    pts = torch.rand(m, 3, generator=gen) * 2 - 1
    view = torch.randn(m, 3, generator=gen)
    nrm = torch.randn(m, 3, generator=gen)
    ref = osh.inner_light(g.sd, pts, view, nrm)
    W = [(wn_weight(g.sd, f"inner_light.{i}").to(dev), g.sd[f"inner_light.{i}.bias"].to(dev)) for i in (0, 2, 4, 6)]
    for prec in (ops.PREC_F32, ops.PREC_F16X3, ops.PREC_F16X2):
        got = ops.inner_light(W, pts.to(dev), view.to(dev), nrm.to(dev), precision=prec)
        err = rel_err(got.cpu(), ref)
        print(f"inner_light precision={prec}: max rel err {err:.2e}")
        assert err < TOL


@pytest.mark.parametrize("prec", ["f16x3", "f16x2"])
def test_inner_light_indexed_ragged_counts(golden, dev, prec):
    """The staggered kernel's two forms (64-ray passes, f16x3; 128-ray passes, f16x2: MCShader's default) through the hit list:
    device-side counts around the pass sizes (1 ray ... two workgroups' worth), only the listed rows written, the near mask applied,
    against the oracle's inner_light on the same rays."""
    from oracle import shading as osh
    from tensoflow_amd import ops
    from tensoflow_amd.shading import wn_weight
    g = golden("shading_small")
    p = {"f16x3": ops.PREC_F16X3, "f16x2": ops.PREC_F16X2}[prec]
    gen = torch.Generator().manual_seed(7)
    n = 3000
    pos = torch.rand(n, 3, generator=gen) * 1.6 - 0.8
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    depth = torch.rand(n, generator=gen) + 0.1
    depth[::17] = 0.0                                   # below near_eps: the light is masked to 0
    idx = torch.randperm(n, generator=gen)
    ref = osh.inner_light(g.sd, pos, -dirs, nrm) * (depth > 1e-5).float()[:, None]
    W = [(wn_weight(g.sd, f"inner_light.{i}").to(dev), g.sd[f"inner_light.{i}.bias"].to(dev)) for i in (0, 2, 4, 6)]
    args = [t.to(dev) for t in (pos, dirs, nrm)]
    for m in (1, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1000, 3000):
        lights = torch.full((n, 3), -7.0, device=dev)
        count = torch.tensor([m], dtype=torch.int64, device=dev)
        ops.inner_light_indexed(W, *args, idx.to(dev), count, depth.to(dev), lights, precision=p)
        got = lights.cpu()
        sel = idx[:m]
        rest = torch.ones(n, dtype=torch.bool)
        rest[sel] = False
        assert bool((got[rest] == -7.0).all()), m        # rows outside the list are not touched
        parity(got[sel], ref[sel], label="inner_light_indexed_ragged_counts:363")
        assert bool((got[sel][depth[sel] <= 1e-5] == 0).all())


def test_point_prep_matches_oracle(golden, dev):
    """Fused per-point launch (tf_point_fwd): materials and both flow condition rows vs the oracle, ragged last tile."""
    from oracle import flow as ofl
    from oracle import shading as osh
    from tensoflow_amd import ops
    from tensoflow_amd.shading import MCShader
    g = golden("shading_small")
    sh = MCShader(g.sd, g["verts"].numpy(), g["faces"].numpy(), AABB, float(g["unit_size"]), device=dev,
                  n_fixed_diffuse=int(g["sn"][0]))
    gen = torch.Generator().manual_seed(11)
    pn = 1000 + 13                                          # not a multiple of 32
    pts = torch.rand(pn, 3, generator=gen) * 2.2 - 1.1      # some points outside the aabb (clamped taps)
    nrm = torch.nn.functional.normalize(torch.randn(pn, 3, generator=gen), dim=-1)
    view = torch.nn.functional.normalize(torch.randn(pn, 3, generator=gen), dim=-1)
    va = ops.view_angles(nrm.to(dev), view.to(dev))
    met, rough, alb, cd, cs = sh.point_prep(pts.to(dev), va)
    rm, rr, ra = osh.predict_materials(g.sd, pts, AABB)
    parity(met.cpu(), rm, label="point_prep_matches_oracle:384.0")
    parity(rough.cpu(), rr, label="point_prep_matches_oracle:384.1")
    parity(alb.cpu(), ra, label="point_prep_matches_oracle:384.2")
    for got, pfx in ((cd, "flow_diffuse_copy."), (cs, "flow_specular_copy.")):
        ref = ofl.flow_condition(g.sd, pts, va.cpu(), rr, AABB, pfx=pfx)
        assert got.shape == ref.shape == (pn, 37)
        parity(got.cpu(), ref, label="point_prep_matches_oracle:388")
        assert torch.equal(got[:, 30:].cpu(), torch.zeros(pn, 7))


@pytest.mark.parametrize("tag", ["small", "default"])
def test_shade_golden(golden, dev, tag):
    """Whole integral (flow samplers active) vs the reference's `*_nis` outputs."""
    from tensoflow_amd.shading import MCShader
    g = golden("shading_" + tag)
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    sh = MCShader(g.sd, g["verts"].numpy(), g["faces"].numpy(), AABB, float(g["unit_size"]), device=dev, n_fixed_diffuse=n_fd)
    out = sh.shade(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), sn_d, sn_s)
    parity(out["metallic"].cpu(), g.out["metallic"], label="shade_golden:400")
    parity(out["roughness"].cpu(), g.out["roughness"], label="shade_golden:401")
    parity(out["albedo"].cpu(), g.out["albedo"], label="shade_golden:402")
    parity(out["colors"].cpu(), g.out["rgb_pr_nis"], label="shade_golden:403")  # per-pixel, 1e-4
    # integer / boolean outputs against the oracle, bit-exact
    from oracle import shading as osh
    tr = osh.MeshTracer(g["verts"][g["faces"].long()])
    ref = osh.shade(g.sd, tr, float(g["unit_size"]), AABB, g["pts"], g["view_in"], g["normals_in"], sn_d, sn_s,
                    n_fixed_diffuse=n_fd, n_fixed_specular=n_fs, use_flow=True)
    assert torch.equal(out["specular_mask"].cpu(), ref["specular_mask"])
    assert torch.equal(out["specular_rays_id"].cpu(), ref["specular_rays_id"])
    nd = sn_d + n_fd
    live = out["live"].bool().cpu()
    assert torch.equal(out["hit"][:, :nd].cpu(), ref["diffuse_hit"] & live[:, :nd])     # dead rays are never traced
    assert 0.5 < float(live.float().mean()) < 1.0
    # the fused reduction (environment light of the missing rays evaluated in place) equals the two-step form: the full
    # [pn,T,3] light array of get_lights (materialised on demand) reduced by tf_shade_reduce
    from tensoflow_amd import ops as _ops
    col2, dl2, sl2 = _ops.shade_reduce(out["wgt"], out["lights"], nd, sn_s)
    parity(col2.cpu(), out["colors"].cpu(), abs_tol=1e-6, label="shade_golden:419.0")
    parity(dl2.cpu(), out["diffuse_lin"].cpu(), abs_tol=1e-6, label="shade_golden:419.1")
    sh.cull_dead_rays = False                                                         # reference-faithful: trace everything
    out2 = sh.shade(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), sn_d, sn_s)
    assert torch.equal(out2["hit"][:, :nd].cpu(), ref["diffuse_hit"])
    assert torch.equal(out2["colors"], out["colors"])                                 # culling changes nothing, bit for bit
    sh.cull_dead_rays, sh.sort_rays = True, False                                     # rays stored and traced in slot order instead of direction-sorted
    out3 = sh.shade(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), sn_d, sn_s)
    # every per-ray quantity is the same bit for bit; the pixel is the same sum taken over the rows in another order
    for k in ("hit", "dirs", "wgt", "depth"):
        assert torch.equal(out3[k], out[k]), k
    assert torch.equal(out3["hit_lights"][out["hit"]], out["hit_lights"][out["hit"]])
    parity(out3["colors"].cpu(), out["colors"].cpu(), abs_tol=2e-6, label="shade_golden:430")
    order = sh.slot_order(sn_d, sn_s).cpu().long()
    assert torch.equal(order.sort().values, torch.arange(sn_d + n_fd + sn_s))         # a permutation of the slots


def _wide_golden(golden, tag):
    """shading_{s256,s512,stress}.npz hold only what differs from shading_default.npz (tools/gen_golden.py:gen_shading_wide)."""
    base, g = golden("shading_default"), golden("shading_" + tag)
    sd = dict(base.sd)
    sd.update(g.sd)
    return base, g, sd


@pytest.mark.parametrize("tag", ["s256", "s512"])
def test_shade_golden_256_512_flow_samples(golden, dev, tag):
    """BASELINE configs[3] / [4] sample counts: 256 and 512 flow samples per lobe (+ the 512 fixed cosine directions) against the
    reference's MCShadingNetwork.forward, per pixel, in the fp32-grade mode and in the default (f16 inner-light operands) mode."""
    from tensoflow_amd import ops
    from tensoflow_amd.shading import MCShader
    base, g, sd = _wide_golden(golden, tag)
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    assert sn_d == sn_s == int(tag[1:])
    sh = MCShader(sd, base["verts"].numpy(), base["faces"].numpy(), AABB, float(g["unit_size"]), device=dev, n_fixed_diffuse=n_fd)
    args = (g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), sn_d, sn_s)
    for ip in (ops.PREC_F16X3, ops.PREC_F16X2, ops.PREC_F16):
        sh.inner_precision = ip
        out = sh.shade(*args)
        err = rel_err(out["colors"].cpu(), g.out["rgb_pr_nis"])
        print(f"shading_{tag} inner precision {ip}: per-pixel max err {err:.2e}")
        assert err < TOL
        parity(out["roughness"].cpu(), g.out["roughness"], label="shade_golden_256_512_flow_samples:460.0")
        parity(out["albedo"].cpu(), g.out["albedo"], label="shade_golden_256_512_flow_samples:460.1")


def test_inner_light_operand_modes_on_trained_like_net(golden, dev):
    """shading_stress: the reference's MCShadingNetwork with the inner-light net's gains raised until its log-radiance spans
    [-1.8, 0.5] over the hit rays (a freshly initialised net answers ~ -0.69 everywhere, which would hide operand rounding in
    its 256-wide layers).  Per ray (get_lights) and per pixel against the reference, for every operand mode of the decoder."""
    from tensoflow_amd import ops
    from tensoflow_amd.shading import MCShader
    base, g, sd = _wide_golden(golden, "stress")
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    sh = MCShader(sd, base["verts"].numpy(), base["faces"].numpy(), AABB, float(g["unit_size"]), device=dev, n_fixed_diffuse=n_fd)
    args = (g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), sn_d, sn_s)
    hit_ref = g["gl_hit"].bool()
    pts_rep = g["pts"].repeat_interleave(16, 0).to(dev)
    base_l, worst = None, {}
    # Per ray, every mode (the exact-fp32 MFMA one included) differs from the reference's own rays by ~5e-4..2e-3 on this net: the
    # degree-16 IDE polynomials cancel catastrophically in fp32 (any two summation orders differ by ~1e-4 in those features) and
    # the 25x gains amplify it.  The modes run on three kernels with three summation orders of those polynomials (f16x3: the staggered
    # two-team kernel, IDE coefficients four at a time from LDS, 256 -> 3 layer in exact fp32 on the vector unit; f16x2 / f16: the
    # column-owned kernel; fp32: the slab-ring kernel), so per ray they are held to the same 3e-3 band among themselves as against
    # the reference's rays, and the reference comparison is made where the path's bar applies: per pixel.
    for ip, ray_tol in ((ops.PREC_F16X3, 0.0), (ops.PREC_F16X2, 3e-3), (ops.PREC_F16, 3e-3), (ops.PREC_F32, 3e-3)):
        sh.inner_precision = ip
        lights, hit, _ = sh.lights(pts_rep, g["gl_dirs"].to(dev))
        assert torch.equal(hit.cpu(), hit_ref)
        lights = lights.cpu()
        if base_l is None:
            base_l = lights
        ray_ref = float(((lights - g["gl_lights"]).abs() / g["gl_lights"].abs())[hit_ref].max())
        ray = float(((lights - base_l).abs() / base_l.abs())[hit_ref].max())
        pix = rel_err(sh.shade(*args)["colors"].cpu(), g.out["rgb_pr_nis"])
        worst[ip] = (ray, pix)
        print(f"shading_stress inner precision {ip}: per-ray max rel err vs f16x3 mode {ray:.2e}, vs reference {ray_ref:.2e}; per-pixel max err {pix:.2e}")
        assert pix < TOL, (ip, pix)                    # the bar of the path: 1e-4 per pixel, every mode
        assert ray <= ray_tol and ray_ref < 3e-3, (ip, ray, ray_ref)
    # per pixel the four modes sit within 1e-5 of each other: what they share is one flow sample on an ill-conditioned spline root
    # (tests/test_oracle_flow.py), which sets the level (1.6e-5 with round 3's Softplus in the flows' feature net, 5.5e-5 with round
    # 4's -- for every mode, the exact-fp32 one included)
    assert max(v[1] for v in worst.values()) - min(v[1] for v in worst.values()) < 0.1 * TOL


# ------------------------------------------------------------------------------ env-light prefilter (A12)
def test_cubemap_prefilter_vs_oracle(dev):
    from oracle import cubemap as oc
    from tensoflow_amd import ops
    from tensoflow_amd.network.light import ndf_cutoff
    rng = np.random.default_rng(5)
    base = (np.log(0.5) + 0.5 * rng.standard_normal((6, 32, 32, 3))).astype(np.float32)
    tb = torch.from_numpy(base).to(dev)
    m = ops.cubemap_mip(tb)
    assert torch.equal(m.cpu(), torch.from_numpy(oc.mip(base)))                   # box mip: bit-exact
    m16 = m.cpu().numpy()
    d = ops.cubemap_diffuse(m)
    parity(d.cpu(), torch.from_numpy(oc.diffuse(m16)), abs_tol=1e-5, label="cubemap_prefilter_vs_oracle:514")
    for tex, r in ((base, 0.08), (base, 0.29), (m16, 0.5), (m16, 1.0)):
        assert abs(ndf_cutoff(r) - oc.ndf_cutoff(r)) == 0.0
        out, ws = ops.cubemap_specular(torch.from_numpy(tex).to(dev), r, ndf_cutoff(r))
        ref, wref = oc.specular(tex, r, return_wsum=True)
        parity(ws.cpu(), torch.from_numpy(wref), abs_tol=1e-5, label="cubemap_prefilter_vs_oracle:519")
        parity(out.cpu(), torch.from_numpy(ref), label="cubemap_prefilter_vs_oracle:520")
    # adjoints (gather formulation) against the oracle's transposed weights
    g = rng.standard_normal((6, 16, 16, 3)).astype(np.float32)
    tg = torch.from_numpy(g).to(dev)
    parity(ops.cubemap_diffuse(tg, adjoint=True).cpu(), torch.from_numpy(oc.diffuse_bwd(g)), abs_tol=1e-5, rel_tol=1e-3,
           why="measured 3.4e-4 relative at 7.5e-8 absolute: the adjoint sums the 1 536 texel weights of an output in fp32, in another order than numpy", label="cubemap diffuse adjoint")
    for r in (0.29, 1.0):
        _, ws = ops.cubemap_specular(m, r, ndf_cutoff(r))
        gb = ops.cubemap_specular_bwd(tg, ws, r, ndf_cutoff(r))
        parity(gb.cpu(), torch.from_numpy(oc.specular_bwd(g, 16, r)), rel_tol=5e-4,
               why="measured 1.1e-4 relative at 6e-7 absolute: fp32 sums over the GGX window in another order than numpy", label=f"cubemap specular adjoint (roughness {r})")


def test_cubemap_texel_table_changes_no_bit(dev):
    """The GGX prefilter and its adjoint with the texel table (tf_cubemap_texel_table, round 5: direction and area of the running texel
    looked up instead of derived per pair) against the in-kernel derivation: the same expressions, so every output bit is the same."""
    from tensoflow_amd import ops
    from tensoflow_amd.network.light import ndf_cutoff
    gen = torch.Generator().manual_seed(9)
    for res, r in ((128, 0.08), (64, 0.29), (32, 0.5), (16, 1.0)):
        cube = (torch.randn(6, res, res, 3, generator=gen) * 0.5).to(dev)
        a, wa = ops.cubemap_specular(cube, r, ndf_cutoff(r), use_table=True)
        b, wb = ops.cubemap_specular(cube, r, ndf_cutoff(r), use_table=False)
        assert torch.equal(a, b) and torch.equal(wa, wb), (res, r)
        g = torch.randn(6, res, res, 3, generator=gen).to(dev)
        assert torch.equal(ops.cubemap_specular_bwd(g, wa, r, ndf_cutoff(r), use_table=True),
                           ops.cubemap_specular_bwd(g, wa, r, ndf_cutoff(r), use_table=False)), (res, r)
    tab = ops.cubemap_texel_table(16, dev)
    assert tab.shape == (6, 16, 16, 4) and float((tab[..., :3].norm(dim=-1) - 1).abs().max()) < 1e-6 and float(tab[..., 3].min()) > 0


def test_cubemap_prefilter_full_res_rows(dev):
    """128^2 base at roughness 0.08 (the sharp, ill-conditioned lobe): sampled output texels incl. face corners and edges."""
    from oracle import cubemap as oc
    from tensoflow_amd import ops
    from tensoflow_amd.network.light import ndf_cutoff
    rng = np.random.default_rng(6)
    base = (np.log(0.5) + 0.5 * rng.standard_normal((6, 128, 128, 3))).astype(np.float32)
    out, _ = ops.cubemap_specular(torch.from_numpy(base).to(dev), 0.08, ndf_cutoff(0.08))
    rows = np.concatenate([rng.integers(0, 6 * 128 * 128, 500), [0, 127, 128 * 127, 128 * 128 - 1, 5 * 128 * 128 + 64]])
    ref = oc.specular_rows(base, 0.08, rows)
    got = out.reshape(-1, 3)[torch.from_numpy(rows).to(dev)].cpu().numpy()
    err = np.abs(got - ref) / (np.abs(ref) + 1e-3)
    assert np.quantile(err, 0.99) < TOL and err.max() < 2e-3, (np.quantile(err, 0.99), err.max())


def test_envlight_build_mips_autograd(dev):
    """EnvLight.build_mips + diffuse / specular fetch: values vs the oracle stack, d loss / d base vs the oracle adjoint chain."""
    from oracle import cubemap as oc
    from oracle import texture as ot
    from tensoflow_amd.network.light import EnvLight
    rng = np.random.default_rng(7)
    env = EnvLight(trainable=True, max_res=32, min_res=8, device=dev)
    base = (np.log(0.5) + 0.5 * rng.standard_normal((6, 32, 32, 3))).astype(np.float32)
    env.base.data = torch.from_numpy(base).to(dev)
    env.build_mips()
    spec, diff = oc.build_mips(base, min_res=8)
    for a, b in zip(env.specular, spec):
        parity(a.detach().cpu(), torch.from_numpy(b), label="envlight_build_mips_autograd:558")
    parity(env.diffuse.detach().cpu(), torch.from_numpy(diff), abs_tol=1e-5, label="envlight_build_mips_autograd:559")
    dirs = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((4096, 3)).astype(np.float32)), dim=-1)
    rough = torch.from_numpy(rng.uniform(0.02, 1.0, (4096, 1)).astype(np.float32))
    wd = torch.from_numpy(rng.standard_normal((4096, 3)).astype(np.float32))
    ws = torch.from_numpy(rng.standard_normal((4096, 3)).astype(np.float32))
    ld = env(dirs.to(dev))
    ls = env(dirs.to(dev), rough.to(dev))
    loss = (ld * wd.to(dev)).sum() + (ls * ws.to(dev)).sum()
    loss.backward()
    # oracle: values through oracle/texture.py; gradient by hand through the oracle adjoints
    tspec = [torch.from_numpy(s).requires_grad_(True) for s in spec]
    tdiff = torch.from_numpy(diff).requires_grad_(True)
    od = torch.exp(ot.cube_bilinear(tdiff, dirs))
    n = len(spec)
    mip = torch.where(rough < 0.5, (rough.clamp(0.08, 0.5) - 0.08) / 0.42 * (n - 2), (rough.clamp(0.5, 1.0) - 0.5) / 0.5 + n - 2)[:, 0]
    os_ = torch.exp(ot.texture(tspec[0][None], dirs[None, None], mip=[t[None] for t in tspec[1:]], mip_level_bias=mip[None, None],
                               filter_mode="linear-mipmap-linear", boundary_mode="cube")[0, 0])
    parity(ld.detach().cpu(), od.detach(), label="envlight_build_mips_autograd:576.0")
    parity(ls.detach().cpu(), os_.detach(), label="envlight_build_mips_autograd:576.1")
    ((od * wd).sum() + (os_ * ws).sum()).backward()
    rs = [0.08, 0.5, 1.0]
    g_levels = [oc.specular_bwd(tspec[i].grad.numpy(), spec[i].shape[1], rs[i]) for i in range(3)]
    g2 = g_levels[2] + oc.diffuse_bwd(tdiff.grad.numpy())
    g1 = g_levels[1] + oc.mip_bwd(g2)
    g0 = g_levels[0] + oc.mip_bwd(g1)
    parity(env.base.grad.cpu(), torch.from_numpy(g0), tol=5e-4, label="envlight_build_mips_autograd:583")


def test_f16_mode_is_close_but_not_parity_grade(golden, dev):
    """TF_PREC_F16 (plain f16 decoder operands; BASELINE configs[4]) on the default shading golden: the per-pixel colour stays
    within ~1e-3 of the reference (PSNR > 60 dB) -- reported as a separate mode, never used for parity claims."""
    import math
    from tensoflow_amd import ops
    from tensoflow_amd.shading import MCShader
    g = golden("shading_default")
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
    sh = MCShader(g.sd, g["verts"].numpy(), g["faces"].numpy(), AABB, float(g["unit_size"]), device=dev, n_fixed_diffuse=n_fd)
    args = (g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), sn_d, sn_s)
    ref = sh.shade(*args)["colors"].cpu()
    sh.precision = ops.PREC_F16
    got = sh.shade(*args)["colors"].cpu()
    parity(ref, g.out["rgb_pr_nis"], label="f16_mode_is_close_but_not_parity_grade:599")
    mse = float(((got - g.out["rgb_pr_nis"]) ** 2).mean())
    assert 20 * math.log10(1 / math.sqrt(mse)) > 60.0 and rel_err(got, g.out["rgb_pr_nis"]) < 5e-3
    assert not torch.equal(got, ref)                      # the mode really is a different arithmetic
    with pytest.raises(RuntimeError):                     # entry points that do not implement the mode refuse it
        from tensoflow_amd.march import SdfField
        f = SdfField(golden("march_r32").sd, AABB, [32, 32, 32], 3, device=dev)
        f.sdf_alpha(torch.zeros(4, 3, device=dev), None, torch.zeros(4, device=dev), torch.zeros(4, 3, device=dev), 1.0, 0.0,
                    precision=ops.PREC_F16)


@pytest.mark.gpu
def test_half_texel_pyramid_is_the_fp32_path_on_the_rounded_field(golden, dev):
    """TfVmDesc.texel_f16 (BASELINE configs[4] 'fp16 field'): tf_vm_pack_to_f16 rounds the fp32 pyramid to nearest-even halves, and
    the consumers return what their fp32-texel path returns on a pyramid that holds those rounded values -- the standalone gather bit
    for bit, the fused sdf / 7-tap kernel to rounding (its two instantiations contract the blend's multiply-adds differently: ulp-level,
    three orders of magnitude below the effect of the rounded field itself): the format changes the bytes fetched, not the arithmetic."""
    from tensoflow_amd import ops
    g = golden("tensosdf_r32_l3")
    sd = {k: v.to(dev) for k, v in g.sd.items()}
    planes = [sd[f"sdf_plane.{i}"] for i in range(3)]
    lines = [sd[f"sdf_line.{i}"] for i in range(3)]
    p32 = ops.VmPacked(planes, lines, 3)
    p16 = ops.VmPacked(planes, lines, 3, texel_f16=True)
    assert p16.data is None and p16.data16.dtype == torch.float16 and torch.equal(p16.data16, p32.data.half())
    prnd = ops.VmPacked(planes, lines, 3)
    prnd.data.copy_(p32.data.half().float())                         # fp32 pyramid holding the rounded texels
    gen = torch.Generator().manual_seed(5)
    xyz = (torch.rand(5000, 3, generator=gen) * 2.2 - 1.1).to(dev)    # incl. out-of-aabb points (clamped taps)
    lv = (torch.rand(5000, generator=gen) * 4 - 1).to(dev)
    for level in (None, lv):
        assert torch.equal(ops.vm_gather(p16, xyz, level, AABB), ops.vm_gather(prnd, xyz, level, AABB))
    W = [sd["sdf_mat.0.weight"], sd["sdf_mat.0.bias"], sd["sdf_mat.2.weight"], sd["sdf_mat.2.bias"]]
    a = ops.sdf_forward(p16, *W, xyz, lv, AABB, want_feat=True)
    b = ops.sdf_forward(prnd, *W, xyz, lv, AABB, want_feat=True)
    e_same = max(rel_err(a[0], b[0]), rel_err(a[1], b[1]))
    full = ops.sdf_forward(p32, *W, xyz, lv, AABB, want_feat=True)
    e_field = float((a[0] - full[0]).abs().max())
    print(f"half pyramid vs fp32 path on the rounded field: {e_same:.2e}; vs the unrounded field: {e_field:.2e}")
    assert e_same < 2e-6 and 50 * e_same < e_field < 5e-3              # the half field IS a different (rounded) field
    dists = torch.full((5000,), 0.01, device=dev)
    dirs = torch.nn.functional.normalize(torch.randn(5000, 3, generator=gen), dim=-1).to(dev)
    units = [2.0 / 31] * 3
    oa = ops.sdf_alpha(p16, *W, xyz, lv, dists, dirs, AABB, units, 20.0, 0.5, want_hess=False)
    ob = ops.sdf_alpha(prnd, *W, xyz, lv, dists, dirs, AABB, units, 20.0, 0.5, want_hess=False)
    for x, y in zip(oa, ob):
        assert (x is None and y is None) or rel_err(x, y) < 2e-5      # alpha / gradient: differences of decoder outputs (FD over 2 * 0.0645)
    with pytest.raises(RuntimeError):                                  # the exact-fp32 decoder does not take a half pyramid
        ops.sdf_forward(p16, *W, xyz, lv, AABB, want_feat=False, precision=ops.PREC_F32)
    with pytest.raises(RuntimeError):                                  # inference-only format
        ops.vm_gather_bwd(p16, xyz, lv, AABB, torch.ones(5000, 108, device=dev))


def test_inner_light_train_forward_saves_hidden_activations(golden, dev):
    """tf_inner_light_indexed_train_fwd: the lights of tf_inner_light_indexed_fwd (same bits) plus the three hidden layers' post-ReLU
    activations by hit-list row, against the dense-layer kernels on the encoded rows (tf_inner_light_encode + tf_linear_fwd: what
    LightsFn.backward recomputed before round 4); rows beyond the device-side count are not written; the weight gradients through
    LightsFn agree with the recompute path."""
    from tensoflow_amd import ops
    from tensoflow_amd.shading import wn_weight
    g = golden("shading_small")
    gen = torch.Generator().manual_seed(12)
    n, m = 5000, 3217                                   # capacity (all rays) and hits: neither a multiple of the 64-ray pass
    pos = (torch.rand(n, 3, generator=gen) * 1.6 - 0.8).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1).to(dev)
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1).to(dev)
    depth = (torch.rand(n, generator=gen) + 0.1).to(dev)
    idx = torch.randperm(n, generator=gen).to(dev)
    count = torch.tensor([m], dtype=torch.int64, device=dev)
    W = [(wn_weight(g.sd, f"inner_light.{i}").to(dev).contiguous(), g.sd[f"inner_light.{i}.bias"].to(dev).contiguous()) for i in (0, 2, 4, 6)]
    a = torch.full((n, 3), -7.0, device=dev)
    b = torch.full((n, 3), -7.0, device=dev)
    acts = torch.full((3, n, 256), -7.0, device=dev)
    ops.inner_light_indexed(W, pos, dirs, nrm, idx, count, depth, a, precision=ops.PREC_F16X3)
    ops.inner_light_indexed(W, pos, dirs, nrm, idx, count, depth, b, precision=ops.PREC_F16X3, acts=acts)
    assert torch.equal(a, b)
    assert bool((acts[:, m:] == -7.0).all())            # rows beyond the count are not touched
    # layer by layer with the dense-layer kernels (exact fp32): each saved layer from the saved layer below it, the lights from the
    # last one -- the row / unit mapping and the values to fp32 rounding.  The first layer against the encoding kernel's rows: the two
    # kernels sum the degree-16 IDE polynomials in different orders (they cancel in fp32: ~1e-4 in those features), so that step is
    # held to the level the operand-mode tests hold the rays to.
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
    X = ops.inner_light_encode(pos, dirs, nrm, idx, count, ld=128)
    w0 = torch.nn.functional.pad(W[0][0], (0, 5))
    e1 = rel(acts[0, :m], ops.linear_fwd(X, w0, W[0][1], ops.ACT_RELU, 0.0, n_dev=count)[:m])
    e2 = rel(acts[1, :m], ops.linear_fwd(acts[0].contiguous(), W[1][0], W[1][1], ops.ACT_RELU, 0.0, n_dev=count)[:m])
    e3 = rel(acts[2, :m], ops.linear_fwd(acts[1].contiguous(), W[2][0], W[2][1], ops.ACT_RELU, 0.0, n_dev=count)[:m])
    out = ops.linear_fwd(acts[2].contiguous(), W[3][0], W[3][1], ops.ACT_EXP_CLAMP, 5.0, n_dev=count)[:m]
    e4 = rel(out, a[idx[:m]])
    print(f"saved activations: layer 1 vs encode + dense {e1:.2e}; layer 2 from saved 1 {e2:.2e}; layer 3 from saved 2 {e3:.2e}; lights from saved 3 {e4:.2e}")
    assert e1 < 3e-3 and e2 < 1e-5 and e3 < 1e-5 and e4 < 1e-5, (e1, e2, e3, e4)


@pytest.mark.parametrize("n,cnt", [(5000, 3217), (300, 300)])
def test_linear_bwd_chain_equals_layer_by_layer(dev, n, cnt):
    """tf_linear_bwd_fused down a 123(128)-256-256-256-3 stack (ReLU, ReLU, ReLU, exp-clamp: make_predictor_4layer) against tf_linear_bwd
    layer by layer: the activation backward of the layer below folded into the data-gradient product (matrix-core epilogue for the
    256-wide layers, the streaming kernel under the 3-wide one) and its bias gradient as column sums; device-side row count."""
    from tensoflow_amd import ops
    gen = torch.Generator().manual_seed(5)
    dims = [128, 256, 256, 256, 3]
    acts = [ops.ACT_RELU, ops.ACT_RELU, ops.ACT_RELU, ops.ACT_EXP_CLAMP]
    Ws = [(torch.randn(dims[l + 1], dims[l], generator=gen) * (1.5 / dims[l] ** 0.5)).to(dev) for l in range(4)]
    bs = [(0.1 * torch.randn(dims[l + 1], generator=gen)).to(dev) for l in range(4)]
    count = torch.tensor([cnt], dtype=torch.int64, device=dev)
    hs = [torch.randn(n, 128, generator=gen).to(dev)]
    for l in range(4):
        hs.append(ops.linear_fwd(hs[-1], Ws[l], bs[l], acts[l], 5.0, n_dev=count))
    g = torch.randn(n, 3, generator=gen).to(dev)
    ref, gy = {}, g
    for l in (3, 2, 1, 0):
        gx, gw, gb = ops.linear_bwd(hs[l], Ws[l], hs[l + 1], gy, acts[l], 5.0, need_gx=l > 0, n_dev=count)
        ref[l] = (gw, gb)
        gy = gx
    got, gy, is_gz = {}, g, False
    for l in (3, 2, 1, 0):
        gx, gw, gb, gbx = ops.linear_bwd_fused(hs[l], Ws[l], hs[l + 1], gy, acts[l], 5.0, gy_is_gz=is_gz,
                                               x_act=acts[l - 1] if l > 0 else ops.ACT_NONE, need_gx=l > 0, need_gbx=l > 0, n_dev=count)
        got.setdefault(l, [None, None])[0] = gw
        if not is_gz:
            got[l][1] = gb
        if l > 0:
            got.setdefault(l - 1, [None, None])[1] = gbx
        gy, is_gz = gx, l > 0
    for l in range(4):
        for a, b, name in ((got[l][0], ref[l][0], "weight"), (got[l][1], ref[l][1], "bias")):
            err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            assert err < 2e-5, (l, name, err)          # the same products; sums accumulate in another order (atomics)


def test_flow_logq_bwd_any_row_order_and_saved_z(dev):
    """tf_flow_logq_bwd's per-point sums are run-length sums over a tile's rows (round 5): rows of a point are contiguous in every
    caller's order, but any order must give the same gradients -- a random permutation of the (rays_id, x, g) rows against the sorted
    order -- and the forward's own z handed in (`z=`) must give what the kernel's re-evaluation gives."""
    from tensoflow_amd import ops
    from tensoflow_amd.shading import FlowParams
    from tensoflow_amd.synth import random_mc_state
    sd = random_mc_state(seed=4, R=32, flow_R=32, env_res=8)
    fp = FlowParams(sd, "flow_diffuse_copy.", dev)
    gen = torch.Generator().manual_seed(11)
    pn, sn = 300, 37                                              # tiles straddle points; the last tile is ragged
    cond = torch.rand(pn, 37, generator=gen).to(dev)
    keep = torch.rand(pn, sn, generator=gen) < 0.6
    rid = torch.arange(pn)[:, None].expand(pn, sn)[keep]
    m = rid.numel()
    x = torch.rand(m, 2, generator=gen).clamp(0.05, 0.95)
    g = torch.randn(m, 1, generator=gen) / m
    perm = torch.randperm(m, generator=gen)

    def run(order, with_z):
        xx, rr, gg = x[order].contiguous().to(dev), rid[order].contiguous().to(dev), g[order].contiguous().to(dev)
        z = ops.flow_logq(fp.nets, cond, xx, rays_id=rr, precision=ops.PREC_F16X3)[0] if with_z else None
        grads, g_cond, g_x = ops.flow_logq_bwd(fp.nets, cond, xx, gg, rays_id=rr, want_gx=True, z=z)
        flat = [t.clone() for k in range(2) for pair in grads[k] for t in pair] + [g_cond.clone()]
        return flat, g_x.clone()

    ident = torch.arange(m)
    ref, gx_ref = run(ident, False)
    for order, with_z, what in ((perm, False, "permuted rows"), (ident, True, "forward's z handed in"), (perm, True, "both")):
        got, gx = run(order, with_z)
        for a, b in zip(got, ref):
            scale = float(b.abs().max())
            assert scale > 0 and float((a - b).abs().max()) < 2e-5 * scale, what
        inv = torch.empty_like(order)
        inv[order] = torch.arange(m)
        assert float((gx[inv.to(dev)] - gx_ref).abs().max()) < 2e-5 * float(gx_ref.abs().max()), what
