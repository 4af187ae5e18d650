"""Generate golden fixtures under tests/golden/ by RUNNING THE REFERENCE on CPU.

Build-container only (needs /root/reference).  It imports the reference's own classes
through tools/ref_shim.py (stub modules for absent third-party packages, cuda->cpu
rewriting) and records seeded inputs, the reference `state_dict`s and the reference outputs
as small .npz files.  Nothing of the reference's source is stored -- only numbers.

The third-party ops the reference calls (dr.texture, segment_coo, nerfacc scan, BVH) are
served by the oracle's restatements (see oracle/__init__.py: parity unpinned there).

    python tools/gen_golden.py            # regenerate everything
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_shim  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
AABB = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])


def _np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, sd=None, **arrays):
    flat = {k: _np(v) for k, v in arrays.items() if v is not None}
    if sd is not None:
        for k, v in sd.items():
            flat["sd/" + k] = _np(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(flat)} arrays")


def perturb_(params, scale, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in params:
            p.add_(scale * torch.randn(p.shape, generator=g))


def gen_tensosdf():
    from network.fields import TensoSDF
    for tag, R, nlev in (("r32_l1", 32, 1), ("r32_l3", 32, 3), ("r24x32x40_l3", (24, 32, 40), 3)):
        torch.manual_seed(6033)
        gs = torch.tensor([R] * 3 if isinstance(R, int) else list(R))
        net = TensoSDF(gs, AABB, device="cpu", sdf_n_comp=36, sdf_dim=256, app_dim=128,
                       init_n_levels=nlev, sdf_multires=0)
        perturb_(list(net.sdf_plane) + list(net.sdf_line), 0.05, 1)
        net.eval()
        g = torch.Generator().manual_seed(11)
        pts = torch.rand(384, 3, generator=g) * 2.4 - 1.2          # incl. out-of-aabb
        pts[:8] = torch.tensor([[-1., -1, -1], [1, 1, 1], [0, 0, 0], [1, -1, 0.5], [-1.0001, 0.3, 1.0001],
                                [0.999, 0.999, -0.999], [0.5, 0.5, 0.5], [-0.25, 0.75, 0]])
        level = torch.rand(384, 1, generator=g) * 4 - 1            # U(-1,3): exercises both clamps
        level[8:16, 0] = torch.tensor([0.0, 1.0, 2.0, 0.5, 1.5, 2.5, -0.5, 3.5])
        with torch.no_grad():
            out_none = net(pts, None)
            out_lvl = net(pts, level)
            sdf = out_lvl[..., :1]
            grad, nh = net.gradient(pts, level, training=True, sdf=sdf)
            grad0, _ = net.gradient(pts, None, training=False)
        # backward golden: d(sum(out*w))/d(params)
        w = torch.randn(out_lvl.shape, generator=g)
        net.zero_grad()
        (net(pts, level) * w).sum().backward()
        grads = {"grad/" + k: p.grad for k, p in net.named_parameters()}
        save(f"tensosdf_{tag}", sd=net.state_dict(), pts=pts, level=level, out_none=out_none, out_lvl=out_lvl,
             grad_lvl=grad, normal_hessian=nh, grad_none=grad0, grid_size=gs, n_levels=np.int32(nlev),
             units=net.units, bwd_w=w, **grads)


def gen_tensosdf_multires():
    """TensoSDF with sdf_multires > 0 (fields.py:66-91, :293-299; the class default is 3: embed the CONTRACTED point; any other m
    embeds the raw point): forward with / without levels, FD gradient + hessian term, parameter gradients."""
    from network.fields import TensoSDF
    for tag, m in (("mr3", 3), ("mr2", 2)):
        torch.manual_seed(6033)
        gs = torch.tensor([32, 32, 32])
        net = TensoSDF(gs, AABB, device="cpu", sdf_n_comp=36, sdf_dim=256, app_dim=128, init_n_levels=3, sdf_multires=m)
        perturb_(list(net.sdf_plane) + list(net.sdf_line), 0.05, 1)
        perturb_([net.sdf_mat[0].weight], 0.02, 2)          # the embedding's columns start at zero (:84-86): give them something to do
        net.eval()
        g = torch.Generator().manual_seed(12)
        pts = torch.rand(256, 3, generator=g) * 2.4 - 1.2
        level = torch.rand(256, 1, generator=g) * 4 - 1
        with torch.no_grad():
            out_none = net(pts, None)
            out_lvl = net(pts, level)
            grad, nh = net.gradient(pts, level, training=True, sdf=out_lvl[..., :1])
        w = torch.randn(out_lvl.shape, generator=g)
        net.zero_grad()
        (net(pts, level) * w).sum().backward()
        grads = {"grad/" + k: p.grad for k, p in net.named_parameters()}
        save(f"tensosdf_{tag}", sd=net.state_dict(), pts=pts, level=level, out_none=out_none, out_lvl=out_lvl, grad_lvl=grad,
             normal_hessian=nh, grid_size=gs, n_levels=np.int32(3), units=net.units, bwd_w=w, multires=np.int32(m), **grads)


def gen_pwquad():
    from network.flow import ElementWisePWQuadraticTransform
    t = ElementWisePWQuadraticTransform()
    g = torch.Generator().manual_seed(21)
    M = 2048
    wv = torch.randn(M, 1, 21, generator=g) * 1.5
    y = torch.rand(M, 1, generator=g)
    # edge cases: y -> 0/1, equal v, tiny / huge w
    y[:8, 0] = torch.tensor([1e-6, 1 - 1e-6, 0.5, 1e-3, 0.999, 0.25, 0.75, 0.1])
    wv[8:16, 0, :11] = 0.3
    wv[16:24, 0, 11:] = torch.tensor([-12.0, 0, 0, 0, 0, 0, 0, 0, 0, 6.0])
    wv[24:32] = 0.0
    x, lj = t.flow(y, wv, True)
    # bins: recompute with the same public function outputs by calling on detached clones is not
    # possible (bins are internal) -> bins are checked through the oracle (tests) only.
    out, lji = t.flow_inv(y, wv, True)
    save("pwquad", wv=wv[:, 0], y=y[:, 0], inv_x=x[:, 0], inv_logj=lj[:, 0], fwd_out=out[:, 0], fwd_logj=lji[:, 0])


def make_flow(R=32, seed=4):
    from network.flow import TensoFlow
    torch.manual_seed(seed)
    net = TensoFlow(2, AABB, device="cpu", gridSize=[R, R, R])
    perturb_(list(net.nis_plane) + list(net.nis_line), 0.1, 3)
    # make the coupling nets non-trivial but tame
    perturb_([p for n, p in net.flows.named_parameters() if "weight" in n], 0.05, 5)
    net.eval()
    return net


def gen_flow():
    net = make_flow()
    g = torch.Generator().manual_seed(31)
    pn = 48
    pts = torch.rand(pn, 3, generator=g) * 1.6 - 0.8
    va = torch.rand(pn, 2, generator=g)
    rough = torch.rand(pn, 1, generator=g)
    arrays = dict(pts=pts, view_angles=va, roughness=rough)
    with torch.no_grad():
        arrays["cond_feat"] = net.tenso_feature(pts)
        for sn in (8, 32, 128):
            lat, lat_logj = net.latent_prior((pn, sn))
            ang, logj = net.sample(pts, va, rough, sn, return_jacobian=True)
            z, logq = net(pts, va, rough, ang, return_jacobian=True)
            arrays.update({f"latent_{sn}": lat[0], f"latent_logj_{sn}": lat_logj[0], f"angles_{sn}": ang,
                           f"logj_{sn}": logj, f"z_{sn}": z, f"logq_{sn}": logq})
        # arbitrary x (not produced by the flow) + rays_id gather form
        x = torch.rand(pn, 16, 2, generator=g)
        z, logq = net(pts, va, rough, x, return_jacobian=True)
        rid = torch.sort(torch.randint(0, pn, (300,), generator=g)).values
        xr = torch.rand(300, 2, generator=g)
        zr, logqr = net(pts, va, rough, xr, return_jacobian=True, rays_id=rid)
        arrays.update(x_rand=x, z_rand=z, logq_rand=logq, rays_id=rid, x_rid=xr, z_rid=zr, logq_rid=logqr)
    # backward golden of the NIS-style loss  -(w * logq).mean()
    w = torch.rand(pn, 16, 1, generator=g)
    net.zero_grad()
    z, logq = net(pts, va, rough, x, return_jacobian=True)
    (-(w * logq).mean()).backward()
    grads = {"grad/" + k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    save("tensoflow_r32", sd=net.state_dict(), bwd_w=w, **arrays, **grads)


def gen_flow_variants():
    """TensoFlow outside the reference default (flow.py:644-648: flow='pwlinear'; n_bins != 10), the configurations the build evaluates
    by composition: the element-wise transforms on random net outputs (with edge rows) and the whole module -- sample, density of the
    samples and of arbitrary points (with and without rays_id), parameter gradients of an NIS-style loss."""
    from network.flow import ElementWisePWLinearTransform, ElementWisePWQuadraticTransform, TensoFlow
    g = torch.Generator().manual_seed(77)
    arrays = {}
    M = 1024
    for name, T, width in (("pwlinear_b10", ElementWisePWLinearTransform(), 10), ("pwlinear_b7", ElementWisePWLinearTransform(), 7),
                           ("pwquad_b6", ElementWisePWQuadraticTransform(), 13), ("pwquad_b16", ElementWisePWQuadraticTransform(), 33)):
        st = torch.randn(M, 1, width, generator=g) * 1.5
        y = torch.rand(M, 1, generator=g)
        y[:8, 0] = torch.tensor([1e-6, 1 - 1e-6, 0.5, 1e-3, 0.999, 0.25, 0.75, 0.1])
        st[8:16] = 0.0
        st[16:24, 0, -1] = 8.0
        st[24:32, 0, 0] = -12.0
        x, lj = T.flow(y, st, True)
        o, lji = T.flow_inv(y, st, True)
        arrays.update({f"t/{name}/st": st[:, 0], f"t/{name}/y": y[:, 0], f"t/{name}/sample_x": x[:, 0], f"t/{name}/sample_logj": lj[:, 0],
                       f"t/{name}/density_out": o[:, 0], f"t/{name}/density_logj": lji[:, 0]})
    pn = 40
    pts = torch.rand(pn, 3, generator=g) * 1.6 - 0.8
    va = torch.rand(pn, 2, generator=g)
    rough = torch.rand(pn, 1, generator=g)
    arrays.update(pts=pts, view_angles=va, roughness=rough)
    for tag, kw in (("pwlinear", dict(flow="pwlinear", n_bins=10)), ("pwquad6", dict(flow="pwquad", n_bins=6))):
        torch.manual_seed(11)
        net = TensoFlow(2, AABB, device="cpu", gridSize=[32, 32, 32], **kw)
        perturb_(list(net.nis_plane) + list(net.nis_line), 0.1, 3)
        perturb_([p for n, p in net.flows.named_parameters() if "weight" in n], 0.05, 5)
        net.eval()
        with torch.no_grad():
            ang, logj = net.sample(pts, va, rough, 32, return_jacobian=True)
            z, logq = net(pts, va, rough, ang, return_jacobian=True)
            x = torch.rand(pn, 16, 2, generator=g)
            zx, lqx = net(pts, va, rough, x, return_jacobian=True)
            rid = torch.sort(torch.randint(0, pn, (200,), generator=g)).values
            xr = torch.rand(200, 2, generator=g)
            zr, lqr = net(pts, va, rough, xr, return_jacobian=True, rays_id=rid)
        w = torch.rand(pn, 16, 1, generator=g)
        net.zero_grad()
        _, lq = net(pts, va, rough, x, return_jacobian=True)
        (-(w * lq).mean()).backward()
        arrays.update({f"{tag}/angles": ang, f"{tag}/logj": logj, f"{tag}/z": z, f"{tag}/logq": logq, f"{tag}/x_rand": x, f"{tag}/z_rand": zx,
                       f"{tag}/logq_rand": lqx, f"{tag}/rays_id": rid, f"{tag}/x_rid": xr, f"{tag}/z_rid": zr, f"{tag}/logq_rid": lqr,
                       f"{tag}/bwd_w": w})
        arrays.update({f"{tag}/sd/" + k: v for k, v in net.state_dict().items()})
        arrays.update({f"{tag}/grad/" + k: p.grad for k, p in net.named_parameters() if p.grad is not None})
    save("tensoflow_variants", **arrays)


def gen_flow_realnvp():
    """TensoFlow(flow='realnvp') (flow.py:645: Gaussian latent prior :9-24, affine couplings :527-547 without the Reshift input
    activation, the analytic-sigmoid output cell :123-144): the element-wise transform and the cell on random inputs with edge rows; the
    whole module -- sample on RECORDED latent draws (the prior draws fresh normals: `realnvp/latent`), density of the samples and of
    arbitrary points (with and without rays_id), parameter gradients of an NIS-style loss."""
    from network.flow import ElementWiseAffineTransform, InvertibleAnalyticSigmoid, TensoFlow
    g = torch.Generator().manual_seed(78)
    arrays = {}
    M = 512
    T = ElementWiseAffineTransform()
    st = torch.randn(M, 1, 2, generator=g) * 1.5
    y = torch.randn(M, 1, generator=g)
    st[:4, 0, 0] = torch.tensor([-20.0, 20.0, 0.0, -14.0])          # exp(+-s) at the 1e-6 clamp of the log-Jacobian
    x, lj = T.flow(y, st, True)
    o, lji = T.flow_inv(y, st, True)
    arrays.update({"t/affine/st": st[:, 0], "t/affine/y": y[:, 0], "t/affine/sample_x": x[:, 0], "t/affine/sample_logj": lj[:, 0],
                   "t/affine/density_out": o[:, 0], "t/affine/density_logj": lji[:, 0]})
    C = InvertibleAnalyticSigmoid()
    v = torch.randn(M, 2, generator=g) * 4
    v[:4] = torch.tensor([[-30.0, 30.0], [0.0, 0.0], [-14.0, 14.0], [20.0, -20.0]])
    u = torch.rand(M, 2, generator=g)
    u[:4] = torch.tensor([[1e-6, 1 - 1e-6], [0.5, 0.5], [1e-3, 0.999], [1e-7, 0.25]])
    zero = torch.zeros(M, 1)
    cy, clj = C.flow(v, zero, None, True)
    cz, clji = C.flow_inv(u, zero, None, True)
    arrays.update({"t/cell/v": v, "t/cell/sample_y": cy, "t/cell/sample_logj": clj, "t/cell/u": u, "t/cell/density_z": cz, "t/cell/density_logj": clji})
    pn = 40
    pts = torch.rand(pn, 3, generator=g) * 1.6 - 0.8
    va = torch.rand(pn, 2, generator=g)
    rough = torch.rand(pn, 1, generator=g)
    arrays.update(pts=pts, view_angles=va, roughness=rough)
    torch.manual_seed(11)
    net = TensoFlow(2, AABB, device="cpu", gridSize=[32, 32, 32], flow="realnvp")
    perturb_(list(net.nis_plane) + list(net.nis_line), 0.1, 3)
    perturb_([p for n, p in net.flows.named_parameters() if "weight" in n], 0.05, 5)
    net.eval()
    latent = torch.randn(pn, 32, 2, generator=g)
    prior = net.latent_prior
    prior.forward = lambda shape: (latent, -prior.log_prob(latent))       # the recorded draws instead of fresh ones (shape == (pn, 32))
    with torch.no_grad():
        ang, logj = net.sample(pts, va, rough, 32, return_jacobian=True)
        z, logq = net(pts, va, rough, ang, return_jacobian=True)
        x = torch.rand(pn, 16, 2, generator=g)
        zx, lqx = net(pts, va, rough, x, return_jacobian=True)
        rid = torch.sort(torch.randint(0, pn, (200,), generator=g)).values
        xr = torch.rand(200, 2, generator=g)
        zr, lqr = net(pts, va, rough, xr, return_jacobian=True, rays_id=rid)
    w = torch.rand(pn, 16, 1, generator=g)
    net.zero_grad()
    _, lq = net(pts, va, rough, x, return_jacobian=True)
    (-(w * lq).mean()).backward()
    tag = "realnvp"
    arrays.update({f"{tag}/latent": latent, f"{tag}/angles": ang, f"{tag}/logj": logj, f"{tag}/z": z, f"{tag}/logq": logq, f"{tag}/x_rand": x,
                   f"{tag}/z_rand": zx, f"{tag}/logq_rand": lqx, f"{tag}/rays_id": rid, f"{tag}/x_rid": xr, f"{tag}/z_rid": zr, f"{tag}/logq_rid": lqr,
                   f"{tag}/bwd_w": w})
    arrays.update({f"{tag}/sd/" + k: v for k, v in net.state_dict().items()})
    arrays.update({f"{tag}/grad/" + k: p.grad for k, p in net.named_parameters() if p.grad is not None})
    save("tensoflow_realnvp", **arrays)


def gen_encodings():
    from utils.network_utils import get_embedder
    from utils.ref_utils import generate_ide_fn
    from utils.raw_utils import linear_to_srgb
    g = torch.Generator().manual_seed(41)
    x3 = torch.randn(64, 3, generator=g)
    d = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    lin = torch.cat([torch.rand(60, generator=g) * 2, torch.tensor([0.0, 0.0031308, 1e-9, 5.0])])
    arr = dict(x3=x3, dirs=d, lin=lin, srgb=linear_to_srgb(lin))
    for nf in (3, 6, 8):
        e, _ = get_embedder(nf, 3)
        arr[f"posenc{nf}"] = e(x3)
    ide = generate_ide_fn(5)
    arr["ide5"] = ide(d, 0)
    arr["ide5_rough"] = ide(d, torch.rand(64, 1, generator=g))
    save("encodings", **arr)


# the part of shade_mixed's output dict (fields.py:1232-1256, :1288-1291) that rounds 1-3 did not store: `variance` is read by the
# trainer's progress line (train/trainer_inv.py:299), the rest only inside the reference
AUX_KEYS = ("human_lights", "approximate_light", "inter", "variance", "variance_diffuse_vis", "variance_specular_vis")


def aux_arrays(outputs, prefix="out/"):
    arr = {}
    for k in AUX_KEYS:
        for sfx in ("", "_nis"):
            if k + sfx in outputs:
                arr[prefix + k + sfx] = outputs[k + sfx]
    return arr


def small_mesh():
    from tensoflow_amd.synth import sphere_torus_mesh
    return sphere_torus_mesh(n_lat=8, n_lon=12, n_major=16, n_minor=8)


def gen_shading():
    """MCShadingNetwork.forward (eval, step=None => fixed-sampler pass + flow-sampler pass)."""
    from network.fields import MCShadingNetwork
    from network.materialRenderer import MaterialRenderer
    from oracle.mesh import BruteForceRayTracer
    from tensoflow_amd.synth import sphere_surface_points
    verts, faces = small_mesh()
    host = types.SimpleNamespace(ray_tracer=BruteForceRayTracer(verts, faces), warned_normal=True)
    R = 32
    unit = float((2.0 / (R - 1)))
    trace = lambda o, d: MaterialRenderer.trace(host, o + 2 * unit * d, d)
    for tag, cfg, pn in (
        ("small", dict(diffuse_sample_num=32, specular_sample_num=16, nis_diffuse_sample_num=16,
                       nis_specular_sample_num=8), 48),
        ("default", dict(), 6),
    ):
        torch.manual_seed(4)
        cfg = dict(outer_light_version="envlight", light_exp_max=5.0, inner_light_exp_max=5.0, human_lights=False,
                   gridSize=[R, R, R], light_reso=16, **cfg)
        net = MCShadingNetwork(cfg, trace, AABB)
        # small material planes (the 512^2 default would make a 113 MB fixture)
        g = torch.Generator().manual_seed(3)
        net.mat_plane = torch.nn.ParameterList(
            [torch.nn.Parameter(0.3 * torch.randn(1, 36, R, R, generator=g)) for _ in range(3)])
        net.mat_line = torch.nn.ParameterList(
            [torch.nn.Parameter(0.5 + 0.3 * torch.randn(1, 36, R, 1, generator=g)) for _ in range(3)])
        for fl in (net.flow_diffuse, net.flow_specular, net.flow_diffuse_copy, net.flow_specular_copy):
            perturb_(list(fl.nis_plane) + list(fl.nis_line), 0.1, 3)
            perturb_([p for n, p in fl.flows.named_parameters() if "weight" in n], 0.05, 5)
        with torch.no_grad():
            net.outer_light.base.add_(0.5 * torch.randn(net.outer_light.base.shape, generator=g))
        net.eval()
        pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(pn, seed=6)]
        # un-normalised inputs: forward() normalises them itself
        view_in, nrm_in = view * 1.7, nrm * 0.6
        with torch.no_grad():
            colors, outputs = net(pts, view_in, nrm_in, None, None, False)
            dirs = torch.nn.functional.normalize(torch.randn(pn * 4, 3, generator=g), dim=-1)
            direct = net.outer_light.direct_light(dirs)
            lights, _, inters, lnrm, hit = net.get_lights(pts.repeat_interleave(4, 0)[:, None], dirs[:, None], None)
        keep = ("albedo", "roughness", "metallic", "diffuse_light", "specular_light", "diffuse_color",
                "specular_color", "visibility", "indirect_light", "rgb_pr_nis", "diffuse_color_nis",
                "specular_color_nis", "visibility_nis", "indirect_light_nis", "diffuse_light_nis",
                "specular_light_nis")
        arr = {"out/" + k: outputs[k] for k in keep}
        arr.update(aux_arrays(outputs))
        save(f"shading_{tag}", sd=net.state_dict(), pts=pts, view_in=view_in, normals_in=nrm_in, colors=colors,
             verts=verts, faces=faces, unit_size=np.float32(unit), env_dirs=dirs, env_direct=direct,
             gl_lights=lights[:, 0], gl_hit=hit[:, 0], gl_inters=inters[:, 0],
             sn=np.array([cfg["diffuse_sample_num"] if "diffuse_sample_num" in cfg else 512,
                          cfg.get("specular_sample_num", 256), cfg.get("nis_diffuse_sample_num", 64),
                          cfg.get("nis_specular_sample_num", 32)], np.int32), **arr)


def gen_shading_wide():
    """BASELINE configs[3] / [4] sample counts (256 / 512 flow samples per lobe) and a trained-like inner-light net, on the
    `shading_default` network (same seed, same state_dict: only the arrays that differ are stored, the tests merge them over
    shading_default.npz).  `stress`: the inner-light net's weight-norm gains are raised until its output log-radiance spans about
    +-2 (a freshly initialised net answers ~0 everywhere, which would hide operand-rounding error of its 256-wide layers)."""
    from network.fields import MCShadingNetwork
    from network.materialRenderer import MaterialRenderer
    from oracle.mesh import BruteForceRayTracer
    from tensoflow_amd.synth import sphere_surface_points
    verts, faces = small_mesh()
    host = types.SimpleNamespace(ray_tracer=BruteForceRayTracer(verts, faces), warned_normal=True)
    R = 32
    unit = float((2.0 / (R - 1)))
    trace = lambda o, d: MaterialRenderer.trace(host, o + 2 * unit * d, d)
    base_sd = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "shading_default.npz")).items() if k.startswith("sd/")}
    for tag, extra, pn, seed in (("s256", dict(nis_diffuse_sample_num=256, nis_specular_sample_num=256), 5, 16),
                                 ("s512", dict(nis_diffuse_sample_num=512, nis_specular_sample_num=512), 4, 26),
                                 ("stress", dict(), 24, 36)):
        torch.manual_seed(4)
        cfg = dict(outer_light_version="envlight", light_exp_max=5.0, inner_light_exp_max=5.0, human_lights=False,
                   gridSize=[R, R, R], light_reso=16, **extra)
        net = MCShadingNetwork(cfg, trace, AABB)
        g = torch.Generator().manual_seed(3)
        net.mat_plane = torch.nn.ParameterList(
            [torch.nn.Parameter(0.3 * torch.randn(1, 36, R, R, generator=g)) for _ in range(3)])
        net.mat_line = torch.nn.ParameterList(
            [torch.nn.Parameter(0.5 + 0.3 * torch.randn(1, 36, R, 1, generator=g)) for _ in range(3)])
        for fl in (net.flow_diffuse, net.flow_specular, net.flow_diffuse_copy, net.flow_specular_copy):
            perturb_(list(fl.nis_plane) + list(fl.nis_line), 0.1, 3)
            perturb_([p for n, p in fl.flows.named_parameters() if "weight" in n], 0.05, 5)
        with torch.no_grad():
            net.outer_light.base.add_(0.5 * torch.randn(net.outer_light.base.shape, generator=g))
        changed = {}
        if tag == "stress":
            with torch.no_grad():
                for n_, p_ in net.inner_light.named_parameters():
                    if n_.endswith("original0"):                      # weight-norm gain g
                        p_.mul_(6.0 if n_.startswith("6.") else 1.6)
                    if n_.endswith("bias") and not n_.startswith("6."):
                        p_.add_(0.05 * torch.randn(p_.shape, generator=g))
        net.eval()
        sd = net.state_dict()
        for k, v in sd.items():
            if k not in base_sd or not torch.equal(v, base_sd[k]):
                changed[k] = v
        assert all(k.startswith("inner_light.") for k in changed), list(changed)[:5]
        pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(pn, seed=seed)]
        with torch.no_grad():
            colors, outputs = net(pts, view, nrm, None, None, False)
            gd = torch.Generator().manual_seed(seed + 1)
            dirs = torch.nn.functional.normalize(torch.randn(pn * 16, 3, generator=gd), dim=-1)
            lights, _, inters, lnrm, hit = net.get_lights(pts.repeat_interleave(16, 0)[:, None], dirs[:, None], None)
        hl = lights[:, 0][hit[:, 0]]
        print(tag, "hit rays", int(hit.sum()), "log-radiance of hit rays: mean %.3f std %.3f min %.3f max %.3f" % (
            float(hl.log().mean()), float(hl.log().std()), float(hl.log().min()), float(hl.log().max())))
        keep = ("albedo", "roughness", "metallic", "rgb_pr_nis", "diffuse_color_nis", "specular_color_nis", "visibility_nis",
                "indirect_light_nis", "diffuse_light_nis", "specular_light_nis")
        arr = {"out/" + k: outputs[k] for k in keep}
        save(f"shading_{tag}", sd=changed, pts=pts, view_in=view, normals_in=nrm, colors=colors, unit_size=np.float32(unit),
             gl_dirs=dirs, gl_lights=lights[:, 0], gl_hit=hit[:, 0], gl_inters=inters[:, 0], gl_normals=lnrm[:, 0],
             sn=np.array([512, 256, cfg.get("nis_diffuse_sample_num", 64), cfg.get("nis_specular_sample_num", 32)], np.int32), **arr)


def _march_renderer(**over):
    """The R = 32 ShapeRenderer of the `march_r32` fixture (same seeds -> same state_dict), with injected pre-filtered environment
    maps and FG LUT."""
    from network.shapeRenderer import ShapeRenderer
    R = 32
    cfg = dict(gridSize=[R, R, R], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False,
               isBGWhite=True, has_radiance_field=False, clip_sample_variance=False, apply_occ_loss=True,
               occ_loss_step=10000, device="cpu", database_name="tensoSDF/compressor", nerfDataType=True,
               apply_gaussian_loss=False, inv_s_init=0.3)
    cfg.update(over)
    torch.manual_seed(6033)
    r = ShapeRenderer(cfg, training=False)
    net = r.sdf_network
    perturb_(list(net.sdf_plane) + list(net.sdf_line), 0.02, 1)
    g = torch.Generator().manual_seed(7)
    cn = r.color_network
    perturb_([p for n, p in cn.named_parameters() if "original1" in n], 0.05, 9)
    # injected pre-filtered environment (EnvLight.build_mips needs the CUDA renderutils plugin)
    spec = [0.5 * torch.randn(6, s, s, 3, generator=g) - 0.7 for s in (16, 8, 4)]
    diff = 0.5 * torch.randn(6, 4, 4, 3, generator=g) - 0.7
    cn.envlight.specular, cn.envlight.diffuse = spec, diff
    u = torch.linspace(0, 1, 32)
    lut = torch.stack(torch.meshgrid(u, u, indexing="ij"), -1)
    cn.FG_LUT = torch.stack([0.9 * (1 - lut[..., 1]) * lut[..., 0] + 0.05, 0.1 * (1 - lut[..., 0]) ** 2], -1)[None].contiguous()
    return r, g, spec, diff


def gen_march():
    """ShapeRenderer.sample_ray / compute_sdf_alpha / render_core (shapeRenderer.py:871,995,1105)."""
    from tensoflow_amd.synth import pinhole_rays
    r, g, spec, diff = _march_renderer()
    net, cn = r.sdf_network, r.color_network
    r.eval()
    rn = 96
    o, d, radii, cos = [torch.from_numpy(a) for a in pinhole_rays(rn, seed=2)]
    near, far = r.near_far_from_sphere(o, d)
    arr = dict(rays_o=o, dirs=d, radiis=radii, rays_cos=cos, near=near, far=far)
    with torch.no_grad():
        t0, t1, ridx = r.sample_ray(o, d, near, far, 0, radiis=radii, rays_cos=cos)
    arr.update(t_starts=t0, t_ends=t1, ray_indices=ridx)
    mid = (t0 + t1) * 0.5
    pts = o[ridx] + d[ridx] * mid[:, None]
    lv = torch.log2(r.compute_ball_radii(mid[:, None], radii[ridx], cos[ridx]) / r.base_radii)
    arr.update(sample_pts=pts, sample_levels=lv)
    with torch.no_grad():
        for ca in (0.0, 0.5, 1.0):
            alpha, grad, feat, inv_s, sdf, hess = r.compute_sdf_alpha(pts, lv, t1 - t0, d[ridx], ca, 100, True)
            arr.update({f"alpha_{ca}": alpha})
        arr.update(sa_grad=grad, sa_feat=feat, sa_inv_s=inv_s, sa_sdf=sdf, sa_hess=hess)
        hp = torch.zeros(rn, 3, 4)
        out = r.render_core(o, d, d, radii, cos, t0, t1, ridx, hp, cos_anneal_ratio=0.5, step=100, is_train=True)
        for k in ("ray_rgb", "acc", "normal", "gradient_error", "std", "loss_sparse", "loss_hessian", "loss_tv_sdf"):
            arr["rc/" + k] = out[k]
        col, _, occ = cn(pts, torch.nn.functional.normalize(grad, dim=-1), -d[ridx], feat, hp[ridx], step=100)
        arr.update(shade_color=col, shade_occ_prob=occ["occ_prob"], shade_roughness=occ["roughness"],
                   shade_reflective=occ["reflective"])
    # backward of the rendered colour / acc wrt every parameter that receives a gradient
    w = torch.rand(rn, 3, generator=g)
    r.zero_grad()
    out = r.render_core(o, d, d, radii, cos, t0, t1, ridx, hp, cos_anneal_ratio=0.5, step=100, is_train=True)
    ((out["ray_rgb"] * w).sum() + out["acc"].sum() + 0.1 * out["gradient_error"].mean()).backward()
    grads = {"grad/" + k: p.grad for k, p in r.named_parameters() if p.grad is not None and "envlight" not in k}
    sd = {k: v for k, v in r.state_dict().items() if "FG_LUT" not in k and "envlight.base" not in k and "gaussian" not in k
          and "outer_light" not in k}
    save("march_r32", sd=sd, bwd_w=w, fg_lut=cn.FG_LUT, env_diffuse=diff, env_spec0=spec[0], env_spec1=spec[1],
         env_spec2=spec[2], step_size=r.stepSize, base_radii=r.base_radii, **arr, **grads)


def gen_march_eval():
    """The validation branch of ShapeRenderer.render_core (is_train=False, shapeRenderer.py:1246-1275: expected-depth point,
    re-evaluated normal / materials / lights, get_intersection(sn0=128, sn1=9) of utils/network_utils.py:172-202) on the rays of
    `march_r32`, and a 24 x 24 ShapeRenderer.nvs frame (:569-668).  The renderer is the one of `march_r32` (same seeds); only the
    outputs, the camera and a checksum of the state are stored.  nvs is run with cfg perturb = 0 (the reference leaves the training
    jitter on in nvs -- torch.rand on its device, not reproducible across devices) and torch.set_default_tensor_type made a no-op
    (the reference switches the default tensor type to CUDA there)."""
    from tensoflow_amd.synth import pinhole_rays
    r, g, spec, diff = _march_renderer(perturb=0.0, test_ray_num=200)
    # SMOOTH pre-filtered environment stacks (bilinear blow-up of 2 x 2 control values per face), as GGX / cosine pre-filtering
    # leaves them: the white-noise texels of `march_r32` (log-radiance jumping by 0.7 between neighbours) multiply the ~1e-5 fp32
    # noise of a finite-difference normal by ~10 per radian in the looked-up light -- a fixture that no two fp32 implementations
    # pass at 1e-4 (measured: the exact-fp32 and the f16x3 decoders deviate from the reference by the same 1.4e-4 there)
    ge = torch.Generator().manual_seed(21)
    smooth = lambda s_: (torch.nn.functional.interpolate(0.5 * torch.randn(6, 3, 2, 2, generator=ge), size=(s_, s_), mode="bilinear",
                                                         align_corners=True) - 0.7).permute(0, 2, 3, 1).contiguous()
    spec = [smooth(s_) for s_ in (16, 8, 4)]
    diff = smooth(4)
    r.color_network.envlight.specular, r.color_network.envlight.diffuse = spec, diff
    net = r.sdf_network
    bump = float(os.environ.get("TF_GOLDEN_BUMP", "0.2"))
    perturb_(list(net.sdf_plane) + list(net.sdf_line), bump, 11)       # bumpy geometry: reflected rays must hit something
    with torch.no_grad():
        # the reference's initial field is sdf ~ |x| (a point at the origin): lower the decoder's sdf bias so that there IS a surface
        last = [m for m in net.sdf_mat if isinstance(m, torch.nn.Linear)][-1]
        last.bias[0] -= float(os.environ.get("TF_GOLDEN_RADIUS", "0.35"))
        r.deviation_network.variance.fill_(float(os.environ.get("TF_GOLDEN_VAR", "0.45")))   # sharp surface: acc -> 1, the depth point sits on it
    r.eval()
    rn = 96
    o, d, radii, cos = [torch.from_numpy(a) for a in pinhole_rays(rn, seed=12, h=64, w=64, focal=213.0)]    # every ray near the object
    near, far = r.near_far_from_sphere(o, d)
    arr = {}
    with torch.no_grad():
        t0, t1, ridx = r.sample_ray(o, d, near, far, 0, radiis=radii, rays_cos=cos)
        hp = torch.zeros(rn, 3, 4)
        out = r.render_core(o, d, d, radii, cos, t0, t1, ridx, hp, cos_anneal_ratio=1.0, step=300000, is_train=False)
    for k, v in out.items():
        if isinstance(v, torch.Tensor):
            arr["val/" + k] = v
    # 24 x 24 frame: camera on the unit-2 sphere looking at the origin (blender convention: -z forward, y up)
    h = w = 24
    eye = torch.tensor([1.1, -1.3, 0.9])
    eye = eye / eye.norm() * 2.0
    fwd = -eye / eye.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0]))
    right = right / right.norm()
    up = torch.linalg.cross(right, fwd)
    pose = torch.stack([right, up, -fwd, eye], 1).numpy().astype(np.float32)             # [3,4] camera-to-world
    f = 0.5 * w / np.tan(0.5 * 0.35)
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
    keep = torch.set_default_tensor_type
    torch.set_default_tensor_type = lambda *a, **k: None
    try:
        frame = r.nvs(pose, K, h, w)
    finally:
        torch.set_default_tensor_type = keep
    for k, v in frame.items():
        arr["nvs/" + k] = v
    sd = r.state_dict()
    chk = np.array([float(sd[k].double().abs().sum()) for k in sorted(sd) if sd[k].is_floating_point() and "FG_LUT" not in k
                    and "envlight.base" not in k and "gaussian" not in k and "outer_light" not in k], np.float64)
    over = {k: v for k, v in sd.items() if "sdf_plane" in k or "sdf_line" in k or "deviation_network" in k or "sdf_mat" in k}       # the only tensors that differ from march_r32
    arr.update(env_diffuse=diff, env_spec0=spec[0], env_spec1=spec[1], env_spec2=spec[2])
    save("march_eval_r32", sd=over, nvs_pose=pose, nvs_K=K, nvs_hw=np.array([h, w]), state_checksum=chk, rays_o=o, dirs=d, radiis=radii,
         rays_cos=cos, t_starts=t0, t_ends=t1, ray_indices=ridx, **arr)


def gen_march_late():
    """ShapeRenderer.render_core late in training (step 30000, configs/shape/syn/compressor.yaml switches): radiance field on
    (rad_mlp, 'radiance' / 'roughness_weights'), occlusion loss (compute_occ_loss -> get_intersection 64 + 16), TV and Gaussian
    regularisers; plus the gradients of every parameter under a loss touching all of them."""
    from network.shapeRenderer import ShapeRenderer
    from tensoflow_amd.synth import pinhole_rays
    R = 32
    cfg = dict(gridSize=[R, R, R], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False,
               isBGWhite=True, has_radiance_field=True, radiance_field_step=20000, clip_sample_variance=False, apply_occ_loss=True,
               occ_loss_step=10000, occ_sdf_thresh=0.05, device="cpu", database_name="tensoSDF/compressor", nerfDataType=True,
               apply_gaussian_loss=True, gaussianLoss_step=20000, inv_s_init=0.3, freeze_inv_s_step=8000)
    torch.manual_seed(6033)
    r = ShapeRenderer(cfg, training=False)
    net = r.sdf_network
    perturb_(list(net.sdf_plane) + list(net.sdf_line), 0.02, 1)
    g = torch.Generator().manual_seed(17)
    cn = r.color_network
    perturb_([p for n, p in cn.named_parameters() if "original1" in n], 0.05, 9)
    spec = [0.5 * torch.randn(6, s, s, 3, generator=g) - 0.7 for s in (16, 8, 4)]
    diff = 0.5 * torch.randn(6, 4, 4, 3, generator=g) - 0.7
    cn.envlight.specular, cn.envlight.diffuse = spec, diff
    u = torch.linspace(0, 1, 32)
    lut = torch.stack(torch.meshgrid(u, u, indexing="ij"), -1)
    cn.FG_LUT = torch.stack([0.9 * (1 - lut[..., 1]) * lut[..., 0] + 0.05, 0.1 * (1 - lut[..., 0]) ** 2], -1)[None].contiguous()
    r.eval()
    rn = 64
    o, d, radii, cos = [torch.from_numpy(a) for a in pinhole_rays(rn, seed=12)]
    near, far = r.near_far_from_sphere(o, d)
    with torch.no_grad():
        t0, t1, ridx = r.sample_ray(o, d, near, far, 0, radiis=radii, rays_cos=cos)
    hp = torch.zeros(rn, 3, 4)
    w = torch.rand(rn, 3, generator=g)
    r.zero_grad()
    out = r.render_core(o, d, d, radii, cos, t0, t1, ridx, hp, cos_anneal_ratio=0.6, step=30000, is_train=True)
    arr = dict(rays_o=o, dirs=d, radiis=radii, rays_cos=cos, near=near, far=far, t_starts=t0, t_ends=t1, ray_indices=ridx, bwd_w=w)
    for k in ("ray_rgb", "acc", "normal", "radiance", "roughness_weights", "std", "loss_occ", "loss_gaussian", "loss_tv_sdf",
              "loss_sparse", "loss_hessian"):
        arr["rc/" + k] = out[k]
    loss = ((out["ray_rgb"] * w).sum() + (out["radiance"] * w.flip(0)).sum() + out["acc"].sum() + 0.1 * out["gradient_error"].mean()
            + out["loss_occ"].sum() + 1e-3 * out["loss_gaussian"] + out["loss_tv_sdf"] + 0.1 * out["loss_sparse"])
    loss.backward()
    grads = {"grad/" + k: p.grad for k, p in r.named_parameters() if p.grad is not None and "envlight" not in k}
    sd = {k: v for k, v in r.state_dict().items() if "FG_LUT" not in k and "envlight.base" not in k and "outer_light" not in k}
    save("march_late_r32", sd=sd, fg_lut=cn.FG_LUT, env_diffuse=diff, env_spec0=spec[0], env_spec1=spec[1], env_spec2=spec[2],
         loss=loss.detach(), **arr, **grads)


def gen_refine():
    """MaterialRenderer.trace_sdf_with_mesh (materialRenderer.py:316-343) on a sphere mesh + a TensoSDF of the same sphere."""
    from network.fields import TensoSDF
    from network.materialRenderer import MaterialRenderer
    from network.other_field import SingleVarianceNetwork
    from oracle.mesh import BruteForceRayTracer
    from tensoflow_amd.synth import pinhole_rays, uv_sphere
    R = 32
    torch.manual_seed(6033)
    gs = torch.tensor([R, R, R])
    net = TensoSDF(gs, AABB, device="cpu", sdf_n_comp=36, sdf_dim=256, app_dim=128, init_n_levels=3, sdf_multires=0)
    perturb_(list(net.sdf_plane) + list(net.sdf_line), 0.01, 1)
    net.eval()
    dev_net = SingleVarianceNetwork(init_val=0.3, activation="exp")
    verts, faces = uv_sphere(0.2, 12, 24)                    # the circle-initialised SDF has its zero set near r = 0.2
    host = types.SimpleNamespace(ray_tracer=BruteForceRayTracer(verts, faces), warned_normal=True, radius=torch.tensor(1.0),
                                 unit_size=torch.tensor(2.0 / (R - 1)), sdf_network=net, deviation_net=dev_net,
                                 sdf_inter_fun=lambda x: net.sdf(x, None))
    for name in ("trace", "near_far_from_sphere", "get_intersection_around_mesh"):
        setattr(host, name, types.MethodType(getattr(MaterialRenderer, name), host))
    o, d, _, _ = [torch.from_numpy(a) for a in pinhole_rays(512, seed=3, focal=2400.0)]
    with torch.no_grad():
        inters, normals, depth, hit = MaterialRenderer.trace_sdf_with_mesh(host, o, d, 32, 9)
    save("refine_r32", sd={"sdf_network." + k: v for k, v in net.state_dict().items() if "gaussian" not in k},
         rays_o=o, rays_d=d, verts=verts, faces=faces, inters=inters, normals=normals, depth=depth, hit=hit,
         inv_s=dev_net(torch.zeros(1, 3))[0, 0], unit_size=np.float32(2.0 / (R - 1)))


def gen_material_nvs():
    """A 24 x 24 MaterialRenderer.nvs frame (materialRenderer.py:641-752: NeRF-type ray constructor :647-672, 512-ray chunk loop
    :705-745, _get_trace_ray_batch_info -> trace_sdf_with_mesh(32, 9), shade with step=None, sqrt of the squared roughness :739,
    white background :743, normal (0,0,1) on pixels that miss :725) -- BASELINE configs[4]'s path, run on the imported reference.
    Geometry: the sphere of `refine_r32` (mesh r = 0.2 + the circle-initialised TensoSDF of the same seeds) inside a torus ring that
    the camera does not see (it sits on the ring's axis, the frame covers +-0.15 rad) but the secondary rays do: visibility,
    indirect light and the inner-light net are exercised.  Shader network: the one of `shading_small` (same seeds; asserted).
    Stored: mesh, camera, the 15 output maps, the refined surface points of the frame's rays.  set_default_tensor_type is made a
    no-op (the reference switches the default tensor type to CUDA in nvs)."""
    from network.fields import MCShadingNetwork, TensoSDF
    from network.materialRenderer import MaterialRenderer
    from network.other_field import SingleVarianceNetwork
    from oracle.mesh import BruteForceRayTracer
    from tensoflow_amd.synth import torus, uv_sphere
    R = 32
    unit = float(2.0 / (R - 1))
    # ---- geometry: gen_refine's SDF (same seeds) + sphere mesh, plus the ring
    torch.manual_seed(6033)
    gs = torch.tensor([R, R, R])
    sdf = TensoSDF(gs, AABB, device="cpu", sdf_n_comp=36, sdf_dim=256, app_dim=128, init_n_levels=3, sdf_multires=0)
    perturb_(list(sdf.sdf_plane) + list(sdf.sdf_line), 0.01, 1)
    sdf.eval()
    ref_sd = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "refine_r32.npz")).items() if k.startswith("sd/")}
    assert all(torch.equal(v, ref_sd["sdf_network." + k]) for k, v in sdf.state_dict().items() if "gaussian" not in k), "SDF state differs from refine_r32.npz"
    dev_net = SingleVarianceNetwork(init_val=0.3, activation="exp")
    v0, f0 = uv_sphere(0.2, 12, 24)
    v1, f1 = torus(0.75, 0.12, 16, 8)
    verts, faces = np.concatenate([v0, v1], 0), np.concatenate([f0, f1 + len(v0)], 0)
    mr = MaterialRenderer.__new__(MaterialRenderer)
    torch.nn.Module.__init__(mr)
    shader_cfg = dict(outer_light_version="envlight", light_exp_max=5.0, inner_light_exp_max=5.0, human_lights=False, gridSize=[R, R, R],
                      light_reso=16, diffuse_sample_num=32, specular_sample_num=16, nis_diffuse_sample_num=16, nis_specular_sample_num=8)
    mr.cfg = {**MaterialRenderer.default_cfg, "nerfDataType": True, "device": "cpu", "database_name": "syn/golden", "shader_cfg": shader_cfg}
    mr.warned_normal, mr.device = True, "cpu"
    mr.ray_tracer = BruteForceRayTracer(verts, faces)
    mr.aabb, mr.gridSize = AABB, gs
    mr.center = torch.mean(AABB, 0).float().view(1, 1, 3)
    mr.radius = (AABB[1] - mr.center).mean().float()
    mr.unit_size = torch.mean((AABB[1] - AABB[0]) / (gs - 1), dim=-1)
    mr.sdf_network, mr.deviation_net = sdf, dev_net
    mr.sdf_inter_fun = lambda x: sdf.sdf(x, None)
    # ---- shader: gen_shading('small') (same seeds, same perturbations)
    torch.manual_seed(4)
    mr._init_shader()
    net = mr.shader_network
    g = torch.Generator().manual_seed(3)
    net.mat_plane = torch.nn.ParameterList([torch.nn.Parameter(0.3 * torch.randn(1, 36, R, R, generator=g)) for _ in range(3)])
    net.mat_line = torch.nn.ParameterList([torch.nn.Parameter(0.5 + 0.3 * torch.randn(1, 36, R, 1, generator=g)) for _ in range(3)])
    for fl in (net.flow_diffuse, net.flow_specular, net.flow_diffuse_copy, net.flow_specular_copy):
        perturb_(list(fl.nis_plane) + list(fl.nis_line), 0.1, 3)
        perturb_([p for n, p in fl.flows.named_parameters() if "weight" in n], 0.05, 5)
    with torch.no_grad():
        net.outer_light.base.add_(0.5 * torch.randn(net.outer_light.base.shape, generator=g))
    small = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "shading_small.npz")).items() if k.startswith("sd/")}
    assert all(torch.equal(v, small[k]) for k, v in net.state_dict().items()), "shader state differs from shading_small.npz"
    mr.eval()
    # ---- camera on the ring's axis (blender convention: -z forward, y up), 24 x 24, the sphere fills about two thirds of the frame
    h = w = 24
    pose = np.array([[1.0, 0, 0, 0.0], [0, 1, 0, 0.0], [0, 0, 1, 2.0]], np.float32)
    f = 0.5 * w / np.tan(0.15)
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
    keep = torch.set_default_tensor_type
    torch.set_default_tensor_type = lambda *a, **k: None
    try:
        with torch.no_grad():
            frame = mr.nvs(pose, K, h, w)
    finally:
        torch.set_default_tensor_type = keep
    arr = {"nvs/" + k: v for k, v in frame.items()}
    # the surface points of the frame's rays (what the chunk loop shades), from the same reference methods
    i, j = torch.meshgrid(torch.linspace(0, w - 1, w), torch.linspace(0, h - 1, h), indexing="ij")
    i, j = i.t(), j.t()
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1).reshape(-1, 3)
    rays_d = torch.nn.functional.normalize(dirs @ torch.from_numpy(pose[:, :3]).t(), dim=-1)
    rays_o = torch.from_numpy(pose[:, 3]).expand(h * w, 3)
    with torch.no_grad():
        info = mr._get_trace_ray_batch_info({"rays_o": rays_o.clone(), "rays_d": rays_d.clone()}, is_train=False)
    hit = info["hit_mask"]
    print("material nvs: pixels on the object", int(hit.sum()), "of", h * w, "; mean visibility on them %.3f" % float(frame["occ_trace"].reshape(-1)[hit.numpy()].mean()))
    save("material_nvs_r32", verts=verts, faces=faces, nvs_pose=pose, nvs_K=K, nvs_hw=np.array([h, w]), unit_size=np.float32(unit),
         rays_o=rays_o, rays_d=rays_d, inters=info["inters"], normals=info["normals"], depth=info["depth"], hit=hit, **arr)


def gen_shading_grad():
    """Training-direction golden: MCShadingNetwork.forward(step=600) with the flow samplers active (use_flow_*_copy), loss =
    sum(colors*w) + loss_nis; gradients of every trainable tensor from the reference's autograd."""
    from network.fields import MCShadingNetwork
    from network.materialRenderer import MaterialRenderer
    from oracle.mesh import BruteForceRayTracer
    from tensoflow_amd.synth import sphere_surface_points
    verts, faces = small_mesh()
    host = types.SimpleNamespace(ray_tracer=BruteForceRayTracer(verts, faces), warned_normal=True)
    R = 32
    unit = float((2.0 / (R - 1)))
    trace = lambda o, d: MaterialRenderer.trace(host, o + 2 * unit * d, d)
    torch.manual_seed(4)
    cfg = dict(outer_light_version="envlight", light_exp_max=5.0, inner_light_exp_max=5.0, human_lights=False,
               gridSize=[R, R, R], light_reso=16, diffuse_sample_num=32, specular_sample_num=16, nis_diffuse_sample_num=16,
               nis_specular_sample_num=8)
    net = MCShadingNetwork(cfg, trace, AABB)
    g = torch.Generator().manual_seed(3)
    net.mat_plane = torch.nn.ParameterList([torch.nn.Parameter(0.3 * torch.randn(1, 36, R, R, generator=g)) for _ in range(3)])
    net.mat_line = torch.nn.ParameterList([torch.nn.Parameter(0.5 + 0.3 * torch.randn(1, 36, R, 1, generator=g)) for _ in range(3)])
    for fl in (net.flow_diffuse, net.flow_specular, net.flow_diffuse_copy, net.flow_specular_copy):
        perturb_(list(fl.nis_plane) + list(fl.nis_line), 0.1, 3)
        perturb_([p for n, p in fl.flows.named_parameters() if "weight" in n], 0.05, 5)
    with torch.no_grad():
        net.outer_light.base.add_(0.5 * torch.randn(net.outer_light.base.shape, generator=g))
    for fl in (net.flow_diffuse_copy, net.flow_specular_copy):
        for p in fl.parameters():
            p.requires_grad = False                      # fields.py:1054-1065
    net.use_flow_diffuse_copy = net.use_flow_specular_copy = True
    net.eval()
    pn = 40
    pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(pn, seed=8)]
    w = torch.rand(pn, 3, generator=g)
    net.zero_grad()
    colors, outputs = net(pts, view, nrm, None, 600, False)
    loss = (colors * w).sum() + outputs["loss_nis"]
    loss.backward()
    grads = {"grad/" + k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    grads.update(aux_arrays(outputs, "out600/"))
    grads.update({"out600/" + k: outputs[k] for k in ("diffuse_light", "specular_light", "diffuse_color", "specular_color", "visibility", "indirect_light")})
    save("shading_grad", sd=net.state_dict(), pts=pts, view_in=view, normals_in=nrm, colors=colors, bwd_w=w,
         loss_nis=outputs["loss_nis"], loss_nis_diffuse=outputs["loss_nis_diffuse"], loss_nis_specular=outputs["loss_nis_specular"],
         verts=verts, faces=faces, unit_size=np.float32(unit), sn=np.array([32, 16, 16, 8], np.int32), **grads)


def _gen_shading_variant(name, over, runs=(("flow600", True, True), ("fixed600", False, False)), patch=None, poses=None, more_sd=None):
    """A non-default cfg of MCShadingNetwork (`over`) on the network, mesh and points of `shading_grad` (same seeds: only outputs -- and the
    tensors whose shape the variant changes, sdx/* -- are stored): the eval forward (step None: fixed pass + flow pass), the training step with the flow copies sampling (step
    600) and the training step before the copies exist (step 600, NIS losses fitted on the fixed samples' own direction angles)."""
    from network.fields import MCShadingNetwork
    from network.materialRenderer import MaterialRenderer
    from oracle.mesh import BruteForceRayTracer
    from tensoflow_amd.synth import sphere_surface_points
    verts, faces = small_mesh()
    host = types.SimpleNamespace(ray_tracer=BruteForceRayTracer(verts, faces), warned_normal=True)
    R = 32
    unit = float((2.0 / (R - 1)))
    trace = lambda o, d: MaterialRenderer.trace(host, (o + 2 * unit * d).detach(), d.detach())
    base = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "shading_grad.npz")).items() if k.startswith("sd/")}
    cfg = {**dict(outer_light_version="envlight", light_exp_max=5.0, inner_light_exp_max=5.0, human_lights=False,
                  gridSize=[R, R, R], light_reso=16, diffuse_sample_num=32, specular_sample_num=16, nis_diffuse_sample_num=16,
                  nis_specular_sample_num=8), **over}
    pn = 40
    pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(pn, seed=8)]
    g = torch.Generator().manual_seed(9)
    w = torch.rand(pn, 3, generator=g)
    arrays = dict(pts=pts, view_in=view, normals_in=nrm, bwd_w=w)

    extra = {}

    def make():
        torch.manual_seed(4)
        net = MCShadingNetwork(cfg, trace, AABB)
        net.mat_plane = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(1, 36, R, R)) for _ in range(3)])
        net.mat_line = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(1, 36, R, 1)) for _ in range(3)])
        own = net.state_dict()
        fit = {k: v for k, v in {**base, **(more_sd or {})}.items() if k in own and own[k].shape == v.shape}     # (a lobe without its flow holds no flow tensors)
        missing, unexpected = net.load_state_dict(fit, strict=False)
        # tensors whose SHAPE the variant changes (another transform's coupling nets) or that it adds (shade_mixed_all's single flow) keep
        # the seeded initialisation: stored with the golden
        assert set(missing) >= {k for k in base if k in own and own[k].shape != base[k].shape}, missing
        missing = [k for k in missing if k not in (more_sd or {})]
        extra.update({k: own[k].detach().clone() for k in missing})
        for fl in [getattr(net, n) for n in ("flow_copy", "flow_diffuse_copy", "flow_specular_copy") if hasattr(net, n)]:
            for p in fl.parameters():
                p.requires_grad = False
        net.eval()
        if patch is not None:
            patch(net)
        return net
    net = make()
    with torch.no_grad():
        colors, outputs = net(pts, view, nrm, poses, None, False)
    arrays.update({"eval/colors": colors, "eval/rgb_pr_nis": outputs["rgb_pr_nis"], "eval/diffuse_color_nis": outputs["diffuse_color_nis"],
                   "eval/specular_color_nis": outputs["specular_color_nis"], "eval/visibility_nis": outputs["visibility_nis"]})
    for tag, copy_d, copy_s in runs:                 # (tag, use_flow_diffuse_copy, use_flow_specular_copy: update_step's state, :1050-1065)
        net = make()
        net.use_flow_diffuse_copy, net.use_flow_specular_copy = copy_d, copy_s
        net.use_flow_copy = copy_d                    # (shade_mixed_all's single copy)
        net.zero_grad()
        colors, outputs = net(pts, view, nrm, poses, 600, False)
        ((colors * w).sum() + outputs["loss_nis"]).backward()
        arrays.update({f"{tag}/colors": colors, f"{tag}/loss_nis": outputs["loss_nis"]} if "loss_nis_diffuse" not in outputs else
                      {f"{tag}/colors": colors, f"{tag}/loss_nis_diffuse": outputs["loss_nis_diffuse"],
                       f"{tag}/loss_nis_specular": outputs["loss_nis_specular"]})
        arrays.update({f"{tag}/grad/" + k: p.grad for k, p in net.named_parameters() if p.grad is not None})
    arrays.update({"sdx/" + k: v for k, v in extra.items()})
    save(name, verts=verts, faces=faces, unit_size=np.float32(unit), sn=np.array([32, 16, 16, 8], np.int32), **arrays)


def gen_shading_whole():
    """cfg use_half_diffuse = use_half_specular = False (fields.py:661-662): the flows sample the OUTGOING direction instead of the half
    vector (:1117-1134, :1190-1203; NIS losses :1276-1279, :1314-1317)."""
    _gen_shading_variant("shading_whole", dict(use_half_diffuse=False, use_half_specular=False))


def gen_shading_nonis():
    """A lobe without its flow (cfg use_nis_diffuse = False | use_nis_specular = False, fields.py:649-650, :1081, :1160): that lobe keeps
    its fixed sampler in every pass and has no NIS loss (:1257, :1294); and the default cfg in the states where ONE copy is active
    (nis_start_iter_diffuse != nis_start_iter_specular: update_step :1050-1065)."""
    _gen_shading_variant("shading_nonis_d", dict(use_nis_diffuse=False))
    _gen_shading_variant("shading_nonis_s", dict(use_nis_specular=False))
    _gen_shading_variant("shading_mixed", dict(), runs=(("copy_d600", True, False), ("copy_s600", False, True)))


def gen_shading_all():
    """cfg shade_fn = 'shade_mixed_all' with use_nis_all (fields.py:640-641, :1337-1451): one flow over both lobes, one direction set per
    point (its copy's nis_sample_num samples, or the fixed cosine set before the copy exists); half-vector and whole-direction flows."""
    _gen_shading_variant("shading_all", dict(shade_fn="shade_mixed_all", use_nis_all=True, nis_sample_num=16))
    _gen_shading_variant("shading_all_whole", dict(shade_fn="shade_mixed_all", use_nis_all=True, nis_sample_num=16, use_half_all=False))


def shading_realnvp_latent(shape):
    """The Gaussian prior's draws of the `shading_realnvp` runs: a function of the request's shape, so that the test side can repeat them."""
    return torch.randn(*shape, 2, generator=torch.Generator().manual_seed(1000 + int(shape[1])))


def gen_shading_realnvp():
    """cfg flow_diffuse = flow_specular = 'realnvp' (fields.py:653-654, :755-760 -> flow.py:645): Gaussian-prior affine flows in both lobes.
    The prior draws fresh normals on every call: every flow's prior returns `shading_realnvp_latent(shape)` here."""
    def patch(net):
        for n in ("flow_diffuse", "flow_diffuse_copy", "flow_specular", "flow_specular_copy"):
            prior = getattr(net, n).latent_prior
            prior.forward = (lambda shape, prior=prior: (lambda x: (x, -prior.log_prob(x)))(shading_realnvp_latent(shape)))
    _gen_shading_variant("shading_realnvp", dict(flow_diffuse="realnvp", flow_specular="realnvp"), patch=patch)


def gen_shading_envhuman():
    """cfg human_lights = True with the cube-map outer light (fields.py:727-729 + 'envlight' :929-930; blended at :962-968) -- a combination
    no shipped config uses: the capturer's net and the per-point poses of `shading_custom` on the network of `shading_grad`."""
    c = np.load(os.path.join(OUT, "shading_custom.npz"))
    human = {k[3:]: torch.from_numpy(c[k]) for k in c.files if k.startswith("sd/human_light.")}
    assert human, "shading_custom.npz holds the trained human_light net"
    _gen_shading_variant("shading_envhuman", dict(human_lights=True), poses=torch.from_numpy(c["human_poses"]), more_sd=human)


def gen_shading_smith():
    """cfg geometry_type = 'ggx_smith' (fields.py:626, :1026-1033): geometry_ggx_smith_correlated (:1000-1008) in the specular weights."""
    _gen_shading_variant("shading_smith", dict(geometry_type="ggx_smith"))


def gen_shading_pwlinear():
    """cfg flow_diffuse = flow_specular = 'pwlinear' (fields.py:653-654, :755-760 -> flow.py:174-312, :646): piecewise-linear coupling
    transforms in both lobes' flows (their coupling nets end in 10 outputs instead of 21: stored as sdx/*)."""
    _gen_shading_variant("shading_pwlinear", dict(flow_diffuse="pwlinear", flow_specular="pwlinear"))


def gen_shading_ablate():
    """cfg disable_tensorial = disable_reflected = True (fields.py:665-666 -> TensoFlow, flow.py:807-812, :838-843: the flows' tensorial
    feature and view-angle embedding zeroed -- the paper's ablation switches)."""
    _gen_shading_variant("shading_ablate", dict(disable_tensorial=True, disable_reflected=True))


def gen_shading_grad_fixed():
    """Training direction BEFORE the flow copies take over (use_flow_*_copy False: the first nis_start_iter = 1000 steps of the
    material stage, fields.py:1050-1065): MCShadingNetwork.forward at step 100 (fixed samplers only) and step 600 (NIS losses
    fitted on the fixed samples); loss = sum(colors * w) + loss_nis; reference autograd gradients of every trainable tensor --
    incl. the path through the roughness-warped GGX directions (sample_specular_directions, :858-903)."""
    from network.fields import MCShadingNetwork
    from network.materialRenderer import MaterialRenderer
    from oracle.mesh import BruteForceRayTracer
    from tensoflow_amd.synth import sphere_surface_points
    verts, faces = small_mesh()
    host = types.SimpleNamespace(ray_tracer=BruteForceRayTracer(verts, faces), warned_normal=True)
    R = 32
    unit = float((2.0 / (R - 1)))
    # the real `raytracing` extension is a CUDA kernel without a backward: its outputs carry no autograd history.  The brute-force
    # stand-in is written in torch ops, so its inputs are detached here -- otherwise hit points and depths would pick up a gradient
    # wrt the (roughness-dependent) specular directions that the reference does not have.
    trace = lambda o, d: MaterialRenderer.trace(host, (o + 2 * unit * d).detach(), d.detach())
    torch.manual_seed(4)
    cfg = dict(outer_light_version="envlight", light_exp_max=5.0, inner_light_exp_max=5.0, human_lights=False,
               gridSize=[R, R, R], light_reso=16, diffuse_sample_num=32, specular_sample_num=16, nis_diffuse_sample_num=16,
               nis_specular_sample_num=8)
    net = MCShadingNetwork(cfg, trace, AABB)
    g = torch.Generator().manual_seed(3)
    net.mat_plane = torch.nn.ParameterList([torch.nn.Parameter(0.3 * torch.randn(1, 36, R, R, generator=g)) for _ in range(3)])
    net.mat_line = torch.nn.ParameterList([torch.nn.Parameter(0.5 + 0.3 * torch.randn(1, 36, R, 1, generator=g)) for _ in range(3)])
    for fl in (net.flow_diffuse, net.flow_specular, net.flow_diffuse_copy, net.flow_specular_copy):
        perturb_(list(fl.nis_plane) + list(fl.nis_line), 0.1, 3)
        perturb_([p for n, p in fl.flows.named_parameters() if "weight" in n], 0.05, 5)
    with torch.no_grad():
        net.outer_light.base.add_(0.5 * torch.randn(net.outer_light.base.shape, generator=g))
    base = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "shading_grad.npz")).items() if k.startswith("sd/")}
    assert all(torch.equal(v, base[k]) for k, v in net.state_dict().items()), "state differs from shading_grad.npz"
    assert not net.use_flow_diffuse_copy and not net.use_flow_specular_copy
    net.eval()
    pn = 40
    pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(pn, seed=8)]
    w = torch.rand(pn, 3, generator=g)
    arr = dict(pts=pts, view_in=view, normals_in=nrm, bwd_w=w, unit_size=np.float32(unit), sn=np.array([32, 16, 16, 8], np.int32))
    for step in (100, 600):
        net.zero_grad()
        colors, outputs = net(pts, view, nrm, None, step, False)
        loss = (colors * w).sum() + outputs["loss_nis"]
        loss.backward()
        arr.update({f"colors_{step}": colors, f"loss_nis_{step}": outputs["loss_nis"], f"loss_nis_diffuse_{step}": outputs["loss_nis_diffuse"],
                    f"loss_nis_specular_{step}": outputs["loss_nis_specular"]})
        arr.update({f"grad{step}/" + k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
        arr.update(aux_arrays(outputs, f"out{step}/"))
        arr.update({f"out{step}/" + k: outputs[k] for k in ("diffuse_light", "specular_light", "diffuse_color", "specular_color", "visibility", "indirect_light")})
        print(step, "loss_nis", float(outputs["loss_nis"]), "tensors with grad", sum(p.grad is not None for p in net.parameters()))
    save("shading_grad_fixed", **arr)           # network state and mesh: shading_grad.npz (same seed)


def gen_shading_direction():
    """outer_light_version='direction' (configs/mat/syn/{lego,armadillo,horse}.yaml; fields.py:716-718, 913-916): the network of
    `shading_grad` (same seed, same sizes) with the outer light a 72-256-256-256-3 net on the IDE of the ray direction, made
    TRAINED (400 Adam steps on a synthetic sky, through the reference's own module).  Stored: only the tensors that differ from
    shading_grad.npz (`outer_light.*`); eval forward (fixed + flow pass), get_lights on random rays, and the reference's autograd
    gradients of the flow pass (step 1200, copies active) and of the fixed-sampler pass (step 100)."""
    from network.fields import MCShadingNetwork
    from network.materialRenderer import MaterialRenderer
    from oracle.mesh import BruteForceRayTracer
    from tensoflow_amd.synth import sphere_surface_points
    verts, faces = small_mesh()
    host = types.SimpleNamespace(ray_tracer=BruteForceRayTracer(verts, faces), warned_normal=True)
    R = 32
    unit = float((2.0 / (R - 1)))
    trace = lambda o, d: MaterialRenderer.trace(host, (o + 2 * unit * d).detach(), d.detach())      # (as gen_shading_grad_fixed)
    torch.manual_seed(4)
    cfg = dict(outer_light_version="direction", light_exp_max=5.0, inner_light_exp_max=5.0, human_lights=False,
               gridSize=[R, R, R], light_reso=16, diffuse_sample_num=32, specular_sample_num=16, nis_diffuse_sample_num=16,
               nis_specular_sample_num=8)
    net = MCShadingNetwork(cfg, trace, AABB)
    base = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "shading_grad.npz")).items() if k.startswith("sd/")}
    # every tensor shading_grad.npz holds is loaded (the construction order of the two variants differs, so the seeds do not line up)
    net.mat_plane = torch.nn.ParameterList([torch.nn.Parameter(base[f"mat_plane.{i}"].clone()) for i in range(3)])     # (32^2 planes, as there)
    net.mat_line = torch.nn.ParameterList([torch.nn.Parameter(base[f"mat_line.{i}"].clone()) for i in range(3)])
    own = net.state_dict()
    own.update({k: v for k, v in base.items() if k in own})
    net.load_state_dict(own)
    g = torch.Generator().manual_seed(13)
    # a TRAINED outer net instead of a freshly initialised one (which answers its bias, log 0.5, in every direction): 400 Adam steps
    # on a synthetic sky -- a sun lobe over a vertical gradient, log-radiance between about -1.6 and 1.2 -- through the reference's
    # own module and encoding
    sun = torch.nn.functional.normalize(torch.tensor([0.5, -0.3, 0.8]), dim=0)
    sky = lambda d: -1.0 + 2.2 * torch.exp(-6.0 * (1.0 - d @ sun))[:, None] + 0.6 * d[:, 2:3] * torch.tensor([1.0, 0.9, 0.7])
    prm = list(net.outer_light.parameters())
    m1, m2 = [torch.zeros_like(p) for p in prm], [torch.zeros_like(p) for p in prm]
    n_thr = torch.get_num_threads()
    torch.set_num_threads(1)       # the fit on ONE thread: a multi-threaded reduction order would make the trained weights differ from run to run
    for it in range(400):          # Adam written out (torch.optim pulls in modules the import shim cannot inspect)
        d = torch.nn.functional.normalize(torch.randn(2048, 3, generator=g), dim=-1)
        loss = ((net.predict_outer_lights_pts(d).log() - sky(d)) ** 2).mean()
        grads = torch.autograd.grad(loss, prm)
        with torch.no_grad():
            for p_, g_, a_, b_ in zip(prm, grads, m1, m2):
                a_.mul_(0.9).add_(g_, alpha=0.1)
                b_.mul_(0.999).addcmul_(g_, g_, value=0.001)
                p_.sub_(2e-3 * (a_ / (1 - 0.9 ** (it + 1))) / ((b_ / (1 - 0.999 ** (it + 1))).sqrt() + 1e-8))
    torch.set_num_threads(n_thr)
    print("outer net fitted: log-radiance mse %.4f" % float(loss))
    net.zero_grad()
    changed = {k: v for k, v in net.state_dict().items() if k not in base}
    assert all(k.startswith("outer_light.") for k in changed) and "outer_light.base" not in changed, list(changed)
    net.eval()
    pn = 40
    pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(pn, seed=8)]
    w = torch.rand(pn, 3, generator=g)
    arr = dict(pts=pts, view_in=view, normals_in=nrm, bwd_w=w, unit_size=np.float32(unit), sn=np.array([32, 16, 16, 8], np.int32))
    with torch.no_grad():
        colors, outputs = net(pts, view, nrm, None, None, False)
        dirs = torch.nn.functional.normalize(torch.randn(pn * 16, 3, generator=g), dim=-1)
        lights, _, inters, lnrm, hit = net.get_lights(pts.repeat_interleave(16, 0)[:, None], dirs[:, None], None)
        env = net.predict_outer_lights_pts(dirs)
    ml = lights[:, 0][~hit[:, 0]]
    print("miss rays", int((~hit).sum()), "log-radiance: mean %.3f std %.3f min %.3f max %.3f" % (
        float(ml.log().mean()), float(ml.log().std()), float(ml.log().min()), float(ml.log().max())))
    keep = ("albedo", "roughness", "metallic", "diffuse_light", "specular_light", "diffuse_color", "specular_color", "visibility",
            "indirect_light", "rgb_pr_nis", "diffuse_color_nis", "specular_color_nis", "visibility_nis", "indirect_light_nis",
            "diffuse_light_nis", "specular_light_nis")
    arr.update({"out/" + k: outputs[k] for k in keep})
    arr.update(colors=colors, gl_dirs=dirs, gl_lights=lights[:, 0], gl_hit=hit[:, 0], outer_pts=env)
    # training direction, fixed samplers (copies not active yet)
    assert not net.use_flow_diffuse_copy and not net.use_flow_specular_copy
    net.zero_grad()
    c, o = net(pts, view, nrm, None, 100, False)
    ((c * w).sum() + o["loss_nis"]).backward()
    arr.update({"colors_100": c})
    arr.update(aux_arrays(o, "out100/"))
    # (the flows' own gradients are pinned by shading_grad / shading_grad_fixed: not stored again)
    arr.update({"grad100/" + k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None and not k.startswith("flow_")})
    # the same pass in fp64 (the reference's module, default dtype switched): how far the reference's own fp32 gradients are from
    # the exact ones -- the colour gradient reaches the material grids through d IDE / d direction, degree-16 polynomials that
    # cancel badly in fp32 -- is the yardstick the parity test holds the HIP path to
    torch.set_default_dtype(torch.float64)
    cells = [c for c in net.sph_enc.__closure__ if torch.is_tensor(c.cell_contents)]     # the IDE closure's fp32 tables (the fp32-ROUNDED
    kept = [c.cell_contents for c in cells]                                              # coefficients are part of the function: kept, widened)
    try:
        for c in cells:
            c.cell_contents = c.cell_contents.double()
        net.double()
        net.zero_grad()
        c64, o64 = net(pts.double(), view.double(), nrm.double(), None, 100, False)
        ((c64 * w.double()).sum() + o64["loss_nis"]).backward()
        arr.update({"grad100_f64/" + k: p.grad.clone() for k, p in net.named_parameters()
                    if p.grad is not None and (k.startswith("mat_line") or k.startswith("outer_light.0") or k.startswith("roughness_predictor.2"))})
        arr["colors_100_f64"] = c64
    finally:
        torch.set_default_dtype(torch.float32)
        for c, v in zip(cells, kept):
            c.cell_contents = v
        net.float()
    for k in ("mat_line.0", "outer_light.0.parametrizations.weight.original1"):
        a32, a64 = arr["grad100/" + k].double(), arr["grad100_f64/" + k]
        print("fp32 reference vs fp64 reference, grad of", k, ": max %.2e l2 %.2e" % (float((a32 - a64).abs().max() / a64.abs().max()), float((a32 - a64).norm() / a64.norm())))
    # ... and with the flow copies sampling
    for fl in (net.flow_diffuse_copy, net.flow_specular_copy):
        for p in fl.parameters():
            p.requires_grad = False
    net.use_flow_diffuse_copy = net.use_flow_specular_copy = True
    net.zero_grad()
    c, o = net(pts, view, nrm, None, 1200, False)
    ((c * w).sum() + o["loss_nis"]).backward()
    arr.update({"colors_1200": c, "loss_nis_1200": o["loss_nis"]})
    arr.update(aux_arrays(o, "out1200/"))
    arr.update({"grad1200/" + k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None and not k.startswith("flow_")})
    print("grads:", sum(k.startswith("grad100/") for k in arr), sum(k.startswith("grad1200/") for k in arr))
    save("shading_direction", sd=changed, **arr)       # the rest of the state and the mesh: shading_grad.npz


def gen_shading_custom():
    """The real-capture variant of the material shader (configs/mat/custom/*.yaml): outer_light_version='sphere_direction' (144-input
    net: IDE of the direction | IDE of the unit-sphere exit point) + human_lights=True (the capturer's reflection, blended into the
    outer light of the rays that miss).  Network of `shading_grad` (same sizes); the two light nets trained for 300 Adam steps on
    synthetic targets so that they answer with a spread of values; per-point human poses = random rigid transforms that put the
    capturer's plane in front of the object.  Stored: the tensors that differ from shading_grad.npz, the poses, eval forward (fixed +
    flow pass), get_lights on random rays, gradients of the fixed pass (step 100) and of the flow pass (step 1200)."""
    from network.fields import MCShadingNetwork
    from network.materialRenderer import MaterialRenderer
    from oracle.mesh import BruteForceRayTracer
    from tensoflow_amd.synth import sphere_surface_points
    verts, faces = small_mesh()
    host = types.SimpleNamespace(ray_tracer=BruteForceRayTracer(verts, faces), warned_normal=True)
    R = 32
    unit = float((2.0 / (R - 1)))
    trace = lambda o, d: MaterialRenderer.trace(host, (o + 2 * unit * d).detach(), d.detach())
    torch.manual_seed(4)
    cfg = dict(outer_light_version="sphere_direction", light_exp_max=5.0, inner_light_exp_max=5.0, human_lights=True,
               gridSize=[R, R, R], light_reso=16, diffuse_sample_num=32, specular_sample_num=16, nis_diffuse_sample_num=16,
               nis_specular_sample_num=8)
    net = MCShadingNetwork(cfg, trace, AABB)
    base = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "shading_grad.npz")).items() if k.startswith("sd/")}
    net.mat_plane = torch.nn.ParameterList([torch.nn.Parameter(base[f"mat_plane.{i}"].clone()) for i in range(3)])
    net.mat_line = torch.nn.ParameterList([torch.nn.Parameter(base[f"mat_line.{i}"].clone()) for i in range(3)])
    own = net.state_dict()
    own.update({k: v for k, v in base.items() if k in own})
    net.load_state_dict(own)
    g = torch.Generator().manual_seed(17)
    pn = 40
    pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(pn, seed=8)]
    # human poses [pn,3,4]: x_h = R x + t; the capturer's plane z_h = 0 sits 2.5 units from the origin, looking at it
    def rand_pose(n):
        q = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)                  # direction to the capturer
        up = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
        xa = torch.nn.functional.normalize(torch.cross(up, q, dim=-1), dim=-1)
        ya = torch.cross(q, xa, dim=-1)
        Rm = torch.stack([xa, ya, q], 1)                                                            # rows: the capturer's axes
        t = -(Rm @ (2.5 * q)[:, :, None])[..., 0]
        return torch.cat([Rm, t[:, :, None]], -1)
    poses = rand_pose(pn)
    sun = torch.nn.functional.normalize(torch.tensor([0.5, -0.3, 0.8]), dim=0)
    sky = lambda d: -1.0 + 2.2 * torch.exp(-6.0 * (1.0 - d @ sun))[:, None] + 0.6 * d[:, 2:3] * torch.tensor([1.0, 0.9, 0.7])
    prm = list(net.outer_light.parameters()) + list(net.human_light.parameters())
    m1, m2 = [torch.zeros_like(p) for p in prm], [torch.zeros_like(p) for p in prm]
    n_thr = torch.get_num_threads()
    torch.set_num_threads(1)       # (one thread: reproducible weights, see gen_shading_direction)
    for it in range(300):          # Adam written out (see gen_shading_direction)
        d = torch.nn.functional.normalize(torch.randn(1024, 3, generator=g), dim=-1)
        o = 0.6 * torch.nn.functional.normalize(torch.randn(1024, 3, generator=g), dim=-1)
        ps = rand_pose(1024)
        outer = net.predict_outer_lights(o, d)
        hl, hw = net.get_human_light(o, d, ps)
        # targets: the sky for the outer net; a bright disc (weight 0.8 inside radius 0.5 of the plane's centre) for the capturer
        inter = (ps[:, :, :3] @ o[:, :, None] + ps[:, :, 3:])[..., 0]
        loss = ((outer.log() - sky(d)) ** 2).mean() + ((net.human_light(torch.zeros(1, 24))[:, 3:] - 0.6) ** 2).mean()
        loss = loss + ((hl - 0.5 * hw.detach()) ** 2).mean() + 0.0 * inter.sum()
        grads = torch.autograd.grad(loss, prm, allow_unused=True)
        with torch.no_grad():
            for p_, g_, a_, b_ in zip(prm, grads, m1, m2):
                if g_ is None:
                    continue
                a_.mul_(0.9).add_(g_, alpha=0.1)
                b_.mul_(0.999).addcmul_(g_, g_, value=0.001)
                p_.sub_(2e-3 * (a_ / (1 - 0.9 ** (it + 1))) / ((b_ / (1 - 0.999 ** (it + 1))).sqrt() + 1e-8))
    torch.set_num_threads(n_thr)
    net.zero_grad()
    changed = {k: v for k, v in net.state_dict().items() if k not in base}
    assert all(k.startswith("outer_light.") or k.startswith("human_light.") for k in changed), list(changed)
    net.eval()
    w = torch.rand(pn, 3, generator=g)
    arr = dict(pts=pts, view_in=view, normals_in=nrm, bwd_w=w, human_poses=poses, unit_size=np.float32(unit), sn=np.array([32, 16, 16, 8], np.int32))
    with torch.no_grad():
        colors, outputs = net(pts, view, nrm, poses, None, False)
        dirs = torch.nn.functional.normalize(torch.randn(pn * 16, 3, generator=g), dim=-1)
        lights, hlw, inters, lnrm, hit = net.get_lights(pts.repeat_interleave(16, 0), dirs, poses.repeat_interleave(16, 0))
    miss = ~hit
    hlw_n = hlw.norm(dim=-1)
    print("miss rays", int(miss.sum()), "of", hit.numel(), "; capturer seen by", int((hlw_n > 0).sum()), "of them; human term mean %.4f max %.4f" % (
        float(hlw_n.mean()), float(hlw_n.max())), "; log-radiance of misses: std %.3f" % float(lights[miss].log().std()))
    keep = ("albedo", "roughness", "metallic", "diffuse_light", "specular_light", "diffuse_color", "specular_color", "visibility",
            "indirect_light", "rgb_pr_nis", "diffuse_color_nis", "specular_color_nis", "visibility_nis", "indirect_light_nis",
            "diffuse_light_nis", "specular_light_nis")
    arr.update({"out/" + k: outputs[k] for k in keep})
    arr.update(aux_arrays(outputs))
    arr.update(colors=colors, gl_dirs=dirs, gl_lights=lights, gl_hit=hit, gl_human=hlw)
    net.zero_grad()
    c, o = net(pts, view, nrm, poses, 100, False)
    ((c * w).sum() + o["loss_nis"]).backward()
    arr.update({"colors_100": c})
    arr.update(aux_arrays(o, "out100/"))
    arr.update({"grad100/" + k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None and (k.startswith("outer_light") or k.startswith("human_light") or k.startswith("mat_line") or k.startswith("roughness_predictor.2"))})
    for fl in (net.flow_diffuse_copy, net.flow_specular_copy):
        for p in fl.parameters():
            p.requires_grad = False
    net.use_flow_diffuse_copy = net.use_flow_specular_copy = True
    net.zero_grad()
    c, o = net(pts, view, nrm, poses, 1200, False)
    ((c * w).sum() + o["loss_nis"]).backward()
    arr.update({"colors_1200": c, "loss_nis_1200": o["loss_nis"]})
    arr.update(aux_arrays(o, "out1200/"))
    arr.update({"grad1200/" + k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None and (k.startswith("outer_light") or k.startswith("human_light") or k.startswith("mat_line"))})
    print("grads:", sum(k.startswith("grad100/") for k in arr), sum(k.startswith("grad1200/") for k in arr))
    save("shading_custom", sd=changed, **arr)


def gen_shape_variants():
    """ShapeShadingNetwork.forward with the switches no shipped configs/shape file sets (fields.py:344,354-357,367-370,377-392,
    394-404,420-439): human_light (the capturer's reflection through IPE + a 24 -> 4 net), sphere_direction (144-input outer net:
    state-dict shape only, forward never calls it) and mat_pos_multires = 4 (embedded position behind the features of mat_mlp):
    both forms of forward (colour / intermediate dict) and the gradients of a weighted colour sum."""
    from network.fields import ShapeShadingNetwork
    torch.manual_seed(4417)
    cn = ShapeShadingNetwork(dict(human_light=True, sphere_direction=True, mat_pos_multires=4))
    perturb_([p for n, p in cn.named_parameters() if "original1" in n], 0.05, 11)
    g = torch.Generator().manual_seed(23)
    spec = [0.5 * torch.randn(6, s, s, 3, generator=g) - 0.7 for s in (16, 8, 4)]
    diff = 0.5 * torch.randn(6, 4, 4, 3, generator=g) - 0.7
    cn.envlight.specular, cn.envlight.diffuse = spec, diff
    u = torch.linspace(0, 1, 32)
    lut = torch.stack(torch.meshgrid(u, u, indexing="ij"), -1)
    cn.FG_LUT = torch.stack([0.9 * (1 - lut[..., 1]) * lut[..., 0] + 0.05, 0.1 * (1 - lut[..., 0]) ** 2], -1)[None].contiguous()
    n = 300
    pts = torch.rand(n, 3, generator=g) * 1.6 - 0.8
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    view = torch.nn.functional.normalize(nrm + 0.8 * torch.randn(n, 3, generator=g), dim=-1)
    feat = 0.5 * torch.randn(n, 128, generator=g)
    # capturer poses: random rotations, the camera plane 1.5-3 units away (a good share of the reflected rays reach it inside |m| < 1.5)
    q, _ = torch.linalg.qr(torch.randn(n, 3, 3, generator=g))
    poses = torch.cat([q, (torch.rand(n, 3, 1, generator=g) * 2 - 1) * torch.tensor([0.5, 0.5, 3.0])[None, :, None]], -1).contiguous()
    arr = dict(pts=pts, normals=nrm, view_dirs=view, feat=feat, human_poses=poses)
    with torch.no_grad():
        col, none, occ = cn(pts, nrm.clone(), view, feat, poses, step=100)
        assert none is None
        col2, occ2, inter = cn(pts, nrm.clone(), view, feat, poses, inter_results=True, step=100)
        met, rough, alb = cn.predict_materials(pts, feat)
    arr.update(color=col, occ_prob=occ["occ_prob"], roughness=occ["roughness"], reflective=occ["reflective"],
               pm_metallic=met, pm_roughness=rough, pm_albedo=alb, **{"inter/" + k: v for k, v in inter.items()})
    arr["frac_human_hits"] = (inter["human_light"].abs().sum(-1) > 0).float().mean()
    w = torch.rand(n, 3, generator=g)
    cn.zero_grad()
    nr, ft = nrm.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    col, _, occ = cn(pts, nr, view, ft, poses, step=100)
    ((col * w).sum() + occ["occ_prob"].sum()).backward()
    grads = {"grad/" + k: p.grad for k, p in cn.named_parameters() if p.grad is not None and "envlight" not in k}
    sd = {k: v for k, v in cn.state_dict().items() if "FG_LUT" not in k and "envlight.base" not in k}
    save("shape_variants", sd=sd, bwd_w=w, g_normals=nr.grad, g_feat=ft.grad, fg_lut=cn.FG_LUT, env_diffuse=diff, env_spec0=spec[0],
         env_spec1=spec[1], env_spec2=spec[2], **arr, **grads)


def gen_march_grad():
    """Geometry-only training direction of the ray-march: loss over compute_sdf_alpha + nerfacc compositing outputs
    (shapeRenderer.py:995-1025, :1166-1206); gradients of the SDF field, decoder and variance from the reference autograd."""
    from network.shapeRenderer import ShapeRenderer
    import nerfacc
    from tensoflow_amd.synth import pinhole_rays
    R = 32
    cfg = dict(gridSize=[R, R, R], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False, isBGWhite=True,
               has_radiance_field=False, clip_sample_variance=False, device="cpu", database_name="tensoSDF/compressor",
               nerfDataType=True, inv_s_init=0.3)
    torch.manual_seed(6033)
    r = ShapeRenderer(cfg, training=False)
    perturb_(list(r.sdf_network.sdf_plane) + list(r.sdf_network.sdf_line), 0.02, 1)
    g = torch.Generator().manual_seed(17)
    rn = 64
    o, d, radii, cos = [torch.from_numpy(a) for a in pinhole_rays(rn, seed=4)]
    near, far = r.near_far_from_sphere(o, d)
    with torch.no_grad():
        t0, t1, ridx = r.sample_ray(o, d, near, far, 0, radiis=radii, rays_cos=cos)
    mid = (t0 + t1) * 0.5
    pts = o[ridx] + d[ridx] * mid[:, None]
    lv = torch.log2(r.compute_ball_radii(mid[:, None], radii[ridx], cos[ridx]) / r.base_radii)
    N = pts.shape[0]
    wa, wn, wf = torch.rand(rn, 1, generator=g), torch.randn(rn, 3, generator=g), torch.randn(rn, 8, generator=g)
    r.zero_grad()
    alpha, grad, feat, inv_s, sdf, hess = r.compute_sdf_alpha(pts, lv, t1 - t0, d[ridx], 0.5, 100, True)
    weights, _ = nerfacc.render_weight_from_alpha(alpha, ray_indices=ridx, n_rays=rn)
    acc = nerfacc.accumulate_along_rays(weights, values=None, ray_indices=ridx, n_rays=rn)
    nrm = nerfacc.accumulate_along_rays(weights, values=grad, ray_indices=ridx, n_rays=rn)
    fac = nerfacc.accumulate_along_rays(weights, values=feat[:, :8], ray_indices=ridx, n_rays=rn)
    loss = (acc * wa).sum() + (nrm * wn).sum() + (fac * wf).sum() + 0.1 * ((grad.norm(dim=-1) - 1.0) ** 2).mean() \
        + 0.01 * hess.abs().mean() + torch.exp(-20.0 * sdf.abs()).mean() + torch.mean(1 / inv_s)
    loss.backward()
    grads = {"grad/" + k: p.grad for k, p in r.named_parameters() if p.grad is not None}
    sd = {k: v for k, v in r.state_dict().items() if k.startswith("sdf_network.") and "gaussian" not in k or k.startswith("deviation")}
    save("march_grad", sd=sd, pts=pts, level=lv, dists=t1 - t0, dirs=d[ridx], ray_indices=ridx, n_rays=np.int64(rn), wa=wa, wn=wn, wf=wf,
         loss=loss.detach(), alpha=alpha, **grads)


def gen_trainer():
    """Pins the trainer harness to the reference's OWN code: the loss classes of network/loss.py applied to
    a seeded render-output dict at several steps (shape- and material-stage loss lists), ShapeRenderer / MaterialRenderer
    .compute_rgb_loss, compute_diffuse_light_regularization, TrainerInv.update_learning_rate / N_voxel_list / N_to_reso over the
    step range, MaterialRenderer._construct_ray_batch_nerf (+ get_human_coordinate_poses) on a tiny image set."""
    from network.loss import name2loss
    from network.materialRenderer import MaterialRenderer
    from network.shapeRenderer import ShapeRenderer
    from train.trainer_inv import TrainerInv
    g = torch.Generator().manual_seed(101)
    rn, N = 24, 200
    R = lambda *sh: torch.rand(*sh, generator=g)
    pr = {"ray_rgb": R(rn, 3), "radiance": R(rn, 3), "roughness_weights": R(rn), "gradient_error": R(N), "std": R(()) + 0.1,
          "sdf_pts": torch.randn(N, 3, generator=g) * 0.7, "sdf_vals": torch.randn(N, generator=g) * 0.3, "loss_occ": R(rn, 1),
          "loss_sparse": R(()), "loss_hessian": R(()), "loss_tv_sdf": R(3), "loss_gaussian": R(()), "acc": R(rn), "loss_nis": R(()),
          "loss_mat_reg": R(1), "diffuse_light": R(rn, 3)}
    gt = {"rgbs": R(rn, 3), "masks": (R(rn) > 0.4).float()}
    arr = {"pr/" + k: v for k, v in pr.items()}
    arr.update({"gt/" + k: v for k, v in gt.items()})
    # compute_rgb_loss of both renderers, every kind
    for kind in ("l2", "l1", "smooth_l1", "charbonier"):
        host = types.SimpleNamespace(cfg={"rgb_loss": kind})
        arr[f"shape_rgb_loss/{kind}"] = ShapeRenderer.compute_rgb_loss(host, pr["ray_rgb"], gt["rgbs"])
        if kind in ("l1", "charbonier"):
            arr[f"mat_rgb_loss/{kind}"] = MaterialRenderer.compute_rgb_loss(host, pr["ray_rgb"], gt["rgbs"])
    arr["diffuse_light_reg"] = MaterialRenderer.compute_diffuse_light_regularization(
        types.SimpleNamespace(cfg={"reg_diffuse_light_lambda": 0.1}), pr["diffuse_light"])
    # the shape stage's loss list (configs/shape/syn/compressor.yaml) through the reference's Loss classes
    upsample = [2000, 5000, 10000, 20000]
    shape_cfg = {"loss": ["nerf_render", "eikonal", "std", "init_sdf_reg", "occ", "Sparse", "Hessian", "TV", "mask", "Gaussian"],
                 "eikonal_weight": 0.1, "eikonal_weight_anneal_begin": 1000, "eikonal_weight_anneal_end": 4000,
                 "sparse_update_list": upsample, "sparse_ratio": [1.0, 0.5, 0.25, 0.1], "hessian_update_list": upsample,
                 "hessian_ratio": [1.0, 0.8, 0.3, 0.0], "apply_std_loss": True, "std_loss_weight": 0.05}
    host = types.SimpleNamespace(cfg={"rgb_loss": "charbonier"})
    steps = [0, 500, 999, 1000, 2500, 5000, 12000, 30000]
    arr["steps"] = np.array(steps)
    for name, cfg, extra in (("shape", shape_cfg, {}), ("mat", {"loss": ["nerf_render", "mat_reg", "nis"]}, {})):
        losses = [name2loss[n](cfg) for n in cfg["loss"]]
        for st in steps:
            d = dict(pr)
            if name == "shape":      # what ShapeRenderer.train_step adds before the Loss classes see the dict (:787-793)
                d["loss_rgb"] = ShapeRenderer.compute_rgb_loss(host, pr["ray_rgb"], gt["rgbs"])
                if st > 20000:
                    d["loss_radiance"] = ShapeRenderer.compute_rgb_loss(host, pr["radiance"], gt["rgbs"]) * pr["roughness_weights"]
                    d["loss_rgb"] = d["loss_rgb"] * (1.0 - pr["roughness_weights"])
                d["loss_mask"] = torch.nn.functional.binary_cross_entropy(pr["acc"].clip(1e-3, 1.0 - 1e-3), (gt["masks"] > 0.5).float())
            else:                    # MaterialRenderer.train_step (:555-565)
                d["loss_rgb"] = MaterialRenderer.compute_rgb_loss(host, pr["ray_rgb"], gt["rgbs"])
                d["loss_diffuse_light"] = arr["diffuse_light_reg"]
            log = {}
            for L in losses:
                log.update(L(d, gt, st))
            total = 0
            for k, v in log.items():
                if k.startswith("loss"):
                    total = total + torch.mean(torch.as_tensor(v, dtype=torch.float32))
                    arr[f"{name}/{st}/{k}"] = torch.mean(torch.as_tensor(v, dtype=torch.float32))
            arr[f"{name}/{st}/total"] = total
    # learning-rate factor, voxel schedule, N_to_reso (trainer_inv.py:118-127, :339-353)
    tr = types.SimpleNamespace(cfg={"lr_decay_iters": 40000, "lr_decay_target_ratio": 5e-2}, lr_factor=1.0, pre_lr_factor=1.0)
    fac = []
    for st in range(0, 40000, 997):
        TrainerInv.update_learning_rate(tr, st)
        fac.append([st, tr.lr_factor, tr.pre_lr_factor])
    arr["lr_factor"] = np.array(fac, np.float64)
    nv = (np.round(np.exp(np.linspace(np.log(128 ** 3), np.log(400 ** 3), len(upsample) + 1))).astype(np.int32)).tolist()
    arr["N_voxel_list"] = np.array(nv, np.int64)
    bbox = [[-1.0, -0.8, -1.2], [1.0, 1.1, 0.9]]
    arr["N_to_reso"] = np.array([TrainerInv.N_to_reso(tr, n, bbox) for n in nv], np.int64)
    arr["N_to_reso_bbox"] = np.array(bbox, np.float32)
    # the material stage's own ray constructor
    imn, h, w = 2, 5, 6
    poses = torch.eye(4)[None].repeat(imn, 1, 1)
    rot = torch.linalg.qr(torch.randn(imn, 3, 3, generator=g))[0]
    poses[:, :3, :3] = rot
    poses[:, :3, 3] = torch.randn(imn, 3, generator=g)
    info = {"imgs": R(imn, 3, h, w), "Ks": torch.tensor([[[7.5, 0, 3.0], [0, 7.5, 2.5], [0, 0, 1]]]).repeat(imn, 1, 1), "poses": poses}
    mhost = types.SimpleNamespace(cfg={"fixed_camera": False}, _warn_ray_tracing=lambda c: None)
    mhost.get_human_coordinate_poses = types.MethodType(MaterialRenderer.get_human_coordinate_poses, mhost)
    rb = MaterialRenderer._construct_ray_batch_nerf(mhost, info)
    arr.update({"rays/imgs": info["imgs"], "rays/Ks": info["Ks"], "rays/poses": poses, **{"rays/out_" + k: v for k, v in rb.items()}})
    save("trainer", **arr)


def gen_alpha_mask():
    """ShapeRenderer.updateAlphaMask / compute_gridAlpha / compute_grid_alpha (shapeRenderer.py:257-325) on the march_r32 network:
    first call (no previous mask), second call on a finer lattice with the first mask in place; AlphaGridMask volumes, the raw
    alpha lattice and the shrunk aabb.  (torch.set_default_tensor_type('torch.cuda.FloatTensor') inside updateAlphaMask is made a
    no-op for the call: there is no CUDA device here.)"""
    from network.shapeRenderer import ShapeRenderer
    R = 32
    cfg = dict(gridSize=[R, R, R], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False,
               isBGWhite=True, has_radiance_field=False, clip_sample_variance=False, apply_occ_loss=True,
               occ_loss_step=10000, device="cpu", database_name="tensoSDF/compressor", nerfDataType=True,
               apply_gaussian_loss=False, inv_s_init=0.3)
    torch.manual_seed(6033)
    r = ShapeRenderer(cfg, training=False)
    perturb_(list(r.sdf_network.sdf_plane) + list(r.sdf_network.sdf_line), 0.02, 1)
    r.eval()
    keep = torch.set_default_tensor_type
    torch.set_default_tensor_type = lambda *a, **k: None
    try:
        arr = {}
        alpha0, xyz0 = r.compute_gridAlpha((24, 20, 28))
        arr["alpha_24x20x28"] = alpha0
        aabb1 = r.updateAlphaMask((24, 20, 28))
        arr["mask1"], arr["aabb1"] = r.alphaMask.alpha_volume[0, 0], aabb1
        alpha2, _ = r.compute_gridAlpha((40, 40, 40))          # sampled through mask1
        arr["alpha_40_masked"] = alpha2
        aabb2 = r.updateAlphaMask((40, 40, 40))
        arr["mask2"], arr["aabb2"] = r.alphaMask.alpha_volume[0, 0], aabb2
    finally:
        torch.set_default_tensor_type = keep
    arr["thres"], arr["mul_length"] = np.float32(r.alphaMask_thres), np.float32(r.cfg["mul_length"])
    arr["inv_s"] = r.deviation_network(torch.zeros(1, 3))[0, 0]
    print("mask1 kept %.3f, mask2 kept %.3f" % (float(arr["mask1"].float().mean()), float(arr["mask2"].float().mean())), "aabb1", aabb1.tolist())
    sd = {k: v for k, v in r.state_dict().items() if k.startswith("sdf_network.") and "gaussian" not in k or k.startswith("deviation")}
    base = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "march_r32.npz")).items() if k.startswith("sd/")}
    same = all(k in base and torch.equal(base[k], v) for k, v in sd.items())
    save("alpha_mask_r32", sd=None if same else sd, sd_is_march_r32=np.bool_(same), **arr)


def main():
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['tensosdf', 'tensosdf_multires', 'pwquad', 'flow', 'flow_variants', 'flow_realnvp', 'encodings', 'shading', 'shading_wide', 'march', 'march_eval', 'march_late', 'refine', 'material_nvs', 'shading_grad', 'shading_whole', 'shading_ablate', 'shading_smith', 'shading_pwlinear', 'shading_nonis', 'shading_all', 'shading_realnvp', 'shading_envhuman', 'shading_grad_fixed', 'shading_direction', 'shading_custom', 'march_grad', 'trainer', 'alpha_mask', 'shape_variants']      # every generator, in dependency order
    with ref_shim.reference():
        for w in which:
            globals()["gen_" + w]()


if __name__ == "__main__":
    main()
