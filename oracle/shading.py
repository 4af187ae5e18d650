"""Oracle: the flow-sampled Monte-Carlo rendering integral (TEST INFRASTRUCTURE).

Follows network/fields.py (MCShadingNetwork):
  tenso_feature :776-810, get_orthogonal_directions :812-822,
  sample_diffuse_directions :824-856, sample_specular_directions :858-903,
  get_inner_lights :905-911, predict_outer_lights('envlight') :929-930 / ('direction') :913-916,
  get_lights :951-975, GGX terms :977-1033, predict_materials :1010-1017,
  direction_to_angle :1035-1048, shade_mixed :1075-1335, forward :1453-1473;
network/light.py:125-162 (EnvLight.direct_light); network/materialRenderer.py:221-223,253-263
(ray_trace_fun / trace); utils/base_utils.py:869-882 (sample_sphere).

Eval mode (`is_train=False`) unless random tensors are injected.  Parameters come as the
reference `state_dict` of the MCShadingNetwork (keys `mat_plane.*`, `mat_line.*`,
`{metallic,roughness,albedo}_predictor.*`, `inner_light.*`, `outer_light.base`,
`flow_{diffuse,specular}[_copy].*`).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import flow as oflow
from .encodings import ide5, linear_to_srgb, mlp, posenc
from .mesh import ray_triangles, MAX_DIST
from .segments import segment_coo
from .texture import cube_bilinear
from .vm_field import vm_feature

EPS = 1e-6


def fibonacci_az_el(num_samples, begin_elevation=0):
    """utils/base_utils.py:869-882 -> (azimuth [n], elevation [n]) float64 numpy."""
    ratio = (begin_elevation + 90) / 180
    num_points = int(num_samples // (1 - ratio))
    g = (np.sqrt(5) - 1.0) / 2.0
    n = np.arange(num_points - num_samples, num_points, dtype=np.float64)
    z = 2.0 * n / num_points - 1.0
    return (2 * np.pi * n * g) % (2 * np.pi), np.arcsin(z)


def fixed_direction_samples(n):
    """fields.py:734-742 -> [n,2] float32 (az/2pi, 1-2el/pi)."""
    az, el = fibonacci_az_el(n, 0)
    return torch.from_numpy(np.stack([az * 0.5 / np.pi, 1 - 2 * el / np.pi], -1).astype(np.float32)).to(torch.get_default_dtype())


def sat_dot(a, b):
    return (a * b).sum(-1, keepdim=True).clamp(0.0, 1.0)


def tangent_frame(n):
    """(x, y) with z = n  (fields.py:812-822 + cross)."""
    nx, ny, nz = n[:, 0:1], n[:, 1:2], n[:, 2:3]
    zero = torch.zeros_like(nx)
    o0 = torch.cat([ny, -nx, zero], -1)
    o1 = torch.cat([-nz, zero, nx], -1)
    use0 = (o0.norm(dim=-1) > o1.norm(dim=-1))[:, None]
    x = F.normalize(torch.where(use0, o0, o1), dim=-1)
    y = torch.cross(n, x, dim=-1)
    return x, y


def dir_to_angle(n, x, y, d):
    """d [pn,sn,3] -> (phi, theta) [pn,sn,2]  (fields.py:1035-1048)."""
    cx = (x[:, None] * d).sum(-1, keepdim=True)
    cy = (y[:, None] * d).sum(-1, keepdim=True)
    cz = (n[:, None] * d).sum(-1, keepdim=True).clamp(-1 + EPS, 1 - EPS)
    phi = (torch.atan2(cy, cx) + 2 * np.pi) % (2 * np.pi)
    return torch.cat([phi, torch.acos(cz)], -1)


def ggx_d(NoH, a):
    a2 = a ** 2
    den = NoH ** 2 * (a2 - 1.0) + 1.0
    return a2 / (np.pi * den ** 2).clamp_min(EPS)


def schlick_g1(c, a):
    k = a / 2
    return c / (c * (1 - k) + k + 1e-5)


def predict_materials(sd, pts, aabb):
    planes = [sd[f"mat_plane.{i}"] for i in range(3)]
    lines = [sd[f"mat_line.{i}"] for i in range(3)]
    feat = vm_feature(planes, lines, pts, aabb, None, 3)
    sig = torch.sigmoid
    metallic = mlp(sd, "metallic_predictor", (0, 2), feat, F.relu, sig)
    rough = mlp(sd, "roughness_predictor", (0, 2), feat, F.relu, sig)
    rough = rough * (1.0 - 0.04 ** 2) + 0.04 ** 2
    albedo = mlp(sd, "albedo_predictor", (0, 2), feat, F.relu, sig)
    return metallic, rough, albedo


def env_direct_light(base, d):
    """EnvLight.direct_light (light.py:125-162): exp(bilinear cube lookup of log-radiance)."""
    return torch.exp(cube_bilinear(base, d))


def inner_light(sd, pts, view, nrm, exp_max=5.0):
    """get_inner_lights (fields.py:905-911): pos_enc8(51) + IDE5(refl) (72) -> 4-layer MLP."""
    nrm = F.normalize(nrm, dim=-1)
    view = F.normalize(view, dim=-1)
    refl = (view * nrm).sum(-1, keepdim=True) * nrm * 2 - view
    h = torch.cat([posenc(pts, 8), ide5(refl, 0)], -1)
    return mlp(sd, "inner_light", (0, 2, 4, 6), h, F.relu, lambda t: torch.exp(t.clamp(max=exp_max)))


class MeshTracer:
    """materialRenderer.trace (:253-263) over a triangle soup: brute force, or -- `bvh` = an oracle.mesh.BvhRayTracer over the
    same triangles -- through the CPU BVH (same hits, t within an ulp or two; the only way to trace the 265 k-triangle bench scene on the host)."""

    def __init__(self, tri, bvh=None):
        self.tri = tri
        self.bvh = bvh
        n = torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=-1)
        self.nrm = F.normalize(n, dim=-1)

    def __call__(self, o, d):
        t, f = self.bvh.first_hit(o, d) if self.bvh is not None else ray_triangles(o, d, self.tri)
        pos = o + t[:, None] * d
        fn = torch.where((f >= 0)[:, None], self.nrm[f.clamp(min=0)], torch.zeros_like(pos))
        nrm = F.normalize(-fn, dim=-1)
        return pos, nrm, t[:, None], (t < MAX_DIST)


def outer_light_direction(sd, dirs, light_exp_max=5.0):
    """predict_outer_lights, outer_light_version='direction' (fields.py:913-916; the net: :716-718): the 4-layer predictor on
    sph_enc(directions, 0) -- the rows as they are."""
    return mlp(sd, "outer_light", (0, 2, 4, 6), ide5(dirs, 0), F.relu, lambda t: torch.exp(t.clamp(max=light_exp_max)))


def outer_light_sphere_direction(sd, pts, dirs, light_exp_max=5.0):
    """predict_outer_lights, 'sphere_direction' (fields.py:917-928): IDE of the direction | IDE of the point where the ray leaves the unit
    sphere (get_sphere_intersection, utils/network_utils.py:108-114; points outside 0.999 are pulled in first)."""
    p = pts.clone()
    far = p.norm(dim=-1) > 0.999
    p[far] = p[far] * 0.999
    dtx = (p * dirs).sum(-1, keepdim=True)
    dist = -dtx + torch.sqrt(dtx ** 2 - (p ** 2).sum(-1, keepdim=True) + 1 + 1e-6)
    enc = torch.cat([ide5(dirs, 0), ide5(p + dirs * dist, 0)], -1)
    return mlp(sd, "outer_light", (0, 2, 4, 6), enc, F.relu, lambda t: torch.exp(t.clamp(max=light_exp_max)))


def human_light(sd, pts, dirs, poses):
    """get_human_light (fields.py:935-949) with get_camera_plane_intersection (utils/network_utils.py:69-88) and IPE(mean, 0, 0, 6)
    (:56-61: sin of [2^k mean | 2^k mean + pi/2]) -> (human_lights [n,3], human_weights [n,1])."""
    R, t = poses[:, :, :3], poses[:, :, 3:]
    p_ = (R @ pts[:, :, None] + t)[..., 0]
    d_ = (R @ dirs[:, :, None])[..., 0].clone()
    hits = d_[:, 2].abs() > 1e-4
    d_[~hits, 2] = 1e-4                                   # the reference writes through a view of dirs_
    dist = -p_[:, 2] / d_[:, 2]
    mean = (p_ + dist[:, None] * d_)[:, :2] * 0.3
    hits = (hits & (mean.norm(dim=-1) < 1.5) & (dist > 0)).to(pts.dtype)[:, None]
    mean = mean * hits
    scaled = (mean[:, None, :] * (2.0 ** torch.arange(6, dtype=pts.dtype))[:, None]).reshape(-1, 12)
    pe = torch.sin(torch.cat([scaled, scaled + 0.5 * np.pi], -1))
    h = mlp(sd, "human_light", (0, 2, 4, 6), pe, F.relu, lambda t: torch.exp(t.clamp(max=0.0))) * hits
    return h[:, :3], h[:, 3:].clamp(0.0, 1.0)


def get_lights(sd, tracer, unit_size, pts, dirs, exp_max=5.0, light_exp_max=5.0, poses=None, return_human=False):
    """fields.py:951-975; the variant follows the state dict: 'envlight' (`outer_light.base`), 'direction' / 'sphere_direction'
    (`outer_light.0.*` with 72 / 144 inputs), human lights (`human_light.*`; poses [M,3,4]).
    pts, dirs [M,3] -> lights [M,3], hit [M] bool, inters [M,3]."""
    eps = 1e-5
    o = pts + dirs * eps
    inters, nrm, depth, hit = tracer(o + 2 * unit_size * dirs, dirs)
    lights = torch.zeros_like(pts)
    miss = ~hit
    hlw = torch.zeros(int(miss.sum()), 3, dtype=pts.dtype) if miss.any() else torch.zeros(1, 3, dtype=pts.dtype)     # (:960,966)
    if miss.any():
        if "outer_light.base" in sd:
            outer = env_direct_light(sd["outer_light.base"], dirs[miss])
        elif sd["outer_light.0.parametrizations.weight.original1"].shape[1] == 144:
            outer = outer_light_sphere_direction(sd, pts[miss], dirs[miss], light_exp_max)
        else:
            outer = outer_light_direction(sd, dirs[miss], light_exp_max)
        if "human_light.0.bias" in sd:
            hl, hw = human_light(sd, pts[miss], dirs[miss], poses[miss])
            outer = outer * (1 - hw) + hl * hw
            hlw = hl * hw
        lights[miss] = outer
    if hit.any():
        lights[hit] = inner_light(sd, inters[hit], -dirs[hit], nrm[hit], exp_max)
    lights = lights * (depth > eps).to(lights.dtype)
    if return_human:
        return lights, hit, inters, hlw
    return lights, hit, inters


def half_to_dirs(angles01, n, x, y, view):
    """flow sample (half-vector angles in [0,1]^2) -> H, outgoing dirs, HoV, theta."""
    phi = angles01[..., :1] * (2 * np.pi)
    theta = angles01[..., 1:2] * (0.5 * np.pi)
    H = (torch.sin(theta) * torch.cos(phi)) * x[:, None] + (torch.sin(theta) * torch.sin(phi)) * y[:, None] \
        + torch.cos(theta) * n[:, None]
    HoV = sat_dot(view[:, None], H)
    dirs = HoV * H * 2 - view[:, None]
    return H, dirs, HoV, phi, theta


def whole_to_dirs(angles01, n, x, y):
    """use_half_* False (fields.py:1117-1134, :1190-1203): the flow sample IS the outgoing direction's (phi, theta) in the shading
    frame -> dirs, theta; pdf Jacobian pi^2 sin(theta)."""
    phi = angles01[..., :1] * (2 * np.pi)
    theta = angles01[..., 1:2] * (0.5 * np.pi)
    dirs = (torch.sin(theta) * torch.cos(phi)) * x[:, None] + (torch.sin(theta) * torch.sin(phi)) * y[:, None] + torch.cos(theta) * n[:, None]
    return dirs, theta


def fixed_diffuse_dirs(n, x, y, view, samples, az_jitter=None):
    """sample_diffuse_directions (fields.py:824-856), eval mode -> dirs, pdf."""
    az, el = samples[None, :, 0:1] * np.pi * 2, samples[None, :, 1:2]
    if az_jitter is not None:
        az = (az + az_jitter * np.pi * 2) % (2 * np.pi)
    el_sqrt = torch.sqrt(el + 1e-7)
    cz = torch.sqrt(1 - el + 1e-7)
    cx = el_sqrt * torch.cos(az)
    cy = el_sqrt * torch.sin(az)
    dirs = cx * x[:, None] + cy * y[:, None] + cz * n[:, None]
    pdf = sat_dot(dirs, n[:, None]) / np.pi * (torch.cos((1 - el) * np.pi / 2) * np.pi / 2)
    return dirs, pdf


def fixed_specular_dirs(n, x, y, view, rough, samples):
    """sample_specular_directions (fields.py:858-903), eval mode -> dirs, pdf."""
    az, el = samples[None, :, 0:1], samples[None, :, 1:2]
    phi = np.pi * 2 * az
    a = rough[:, None]
    cos_t = ((1.0 - el) / (1.0 + (a ** 2 - 1.0) * el).clamp_min(EPS)).clamp_min(EPS).sqrt()
    sin_t = (1 - cos_t ** 2).clamp_min(EPS).sqrt()
    H = (torch.cos(phi) * sin_t) * x[:, None] + (torch.sin(phi) * sin_t) * y[:, None] + cos_t * n[:, None]
    VoH = sat_dot(view[:, None], H)
    dirs = VoH * H * 2 - view[:, None]
    NoH = cos_t.clamp_min(0.0)
    pdf = ggx_d(NoH, rough[:, None]) * NoH / (4 * VoH).clamp_min(EPS) * (torch.cos((1 - el) * np.pi / 2) * np.pi / 2)
    return dirs, pdf


def shade(sd, tracer, unit_size, aabb, pts, view, nrm, sn_diffuse, sn_specular, n_fixed_diffuse=512,
          n_fixed_specular=256, use_flow=True, exp_max=5.0, flow_sfx="_copy", human_poses=None, use_half=(True, True), flow_ablate=(False, False), geometry_type="schlick"):
    """MCShadingNetwork.forward -> shade_mixed, eval; outer-light variant and human lights follow the state dict (get_lights).
    use_half = cfg (use_half_diffuse, use_half_specular): False -> that lobe's flow samples the outgoing direction itself;
    flow_ablate = cfg (disable_tensorial, disable_reflected); geometry_type = cfg geometry_type ('schlick' | 'ggx_smith', fields.py:1026-1033).
    Returns dict(colors, diffuse_colors(lin), specular_colors(lin), metallic, roughness, albedo,
                 specular_rays_id, specular_mask, visibility, ...)."""
    view = F.normalize(view, dim=-1)
    nrm = F.normalize(nrm, dim=-1)
    pn = pts.shape[0]
    metallic, rough, albedo = predict_materials(sd, pts, aabb)
    x, y = tangent_frame(nrm)
    va = dir_to_angle(nrm, x, y, view[:, None])[:, 0] / torch.tensor([2 * np.pi, 0.5 * np.pi])
    out = {}

    # ---- diffuse lobe
    ddirs_fix, dpdf_fix = fixed_diffuse_dirs(nrm, x, y, view, fixed_direction_samples(n_fixed_diffuse))
    if use_flow:
        ang, logq = oflow.flow_sample(sd, pts, va, rough, sn_diffuse, aabb, pfx=f"flow_diffuse{flow_sfx}.", ablate=flow_ablate)
        if use_half[0]:
            H, ddirs, HoV, phi, theta = half_to_dirs(ang, nrm, x, y, view)
            dpdf = torch.exp(-logq.clamp(-8, 8)) / (4 * np.pi ** 2 * HoV * torch.sin(theta)).clamp_min(EPS)
        else:
            ddirs, theta = whole_to_dirs(ang, nrm, x, y)
            dpdf = torch.exp(-logq.clamp(-8, 8)) / (np.pi ** 2 * torch.sin(theta)).clamp_min(EPS)
        ddirs = torch.cat([ddirs, ddirs_fix], 1)
        dpdf = torch.cat([dpdf, dpdf_fix], 1)
        out["diffuse_flow_angles"] = ang
        out["diffuse_flow_logq"] = logq
    else:
        ddirs, dpdf = ddirs_fix, dpdf_fix
    dn = ddirs.shape[1]
    kd = 1 - metallic[:, None]
    dl, dhit, _ = get_lights(sd, tracer, unit_size, pts[:, None].expand(pn, dn, 3).reshape(-1, 3),
                             ddirs.reshape(-1, 3), exp_max,
                             poses=human_poses[:, None].expand(pn, dn, 3, 4).reshape(-1, 3, 4) if human_poses is not None else None)
    dl = dl.reshape(pn, dn, 3)
    dw = albedo[:, None] * kd * (sat_dot(ddirs, nrm[:, None]) / np.pi)
    diffuse = torch.mean(dw * dl / dpdf.clamp_min(EPS), 1)

    # ---- specular lobe
    if use_flow:
        ang, logq = oflow.flow_sample(sd, pts, va, rough, sn_specular, aabb, pfx=f"flow_specular{flow_sfx}.", ablate=flow_ablate)
        if use_half[1]:
            H, sdirs, HoVs, phi, theta = half_to_dirs(ang, nrm, x, y, view)
            spdf = torch.exp(-logq.clamp(-8, 8)) / (4 * np.pi ** 2 * HoVs * torch.sin(theta)).clamp_min(EPS)
        else:
            sdirs, theta = whole_to_dirs(ang, nrm, x, y)
            spdf = torch.exp(-logq.clamp(-8, 8)) / (np.pi ** 2 * torch.sin(theta)).clamp_min(EPS)
        out["specular_flow_angles"] = ang
        out["specular_flow_logq"] = logq
    else:
        sdirs, spdf = fixed_specular_dirs(nrm, x, y, view, rough, fixed_direction_samples(n_fixed_specular))
    sn = sdirs.shape[1]
    smask = (sdirs * nrm[:, None]).sum(-1) > 0
    rid = torch.arange(pn)[:, None].repeat(1, sn)[smask]
    sd_, sp_ = sdirs[smask], spdf[smask]
    F0 = 0.04 * (1 - metallic) + metallic * albedo
    Hs = F.normalize(view[rid] + sd_, dim=-1)
    HoV = (Hs * view[rid]).sum(-1, keepdim=True).clamp(0.0, 1.0)
    fres = F0[rid] + (1.0 - F0[rid]) * (1.0 - HoV).clamp(0.0, 1.0) ** 5.0
    NoV = sat_dot(nrm, view)[rid]
    NoL = sat_dot(nrm[rid], sd_)
    if geometry_type == "ggx_smith":            # geometry_ggx_smith_correlated (fields.py:1000-1008)
        lam = lambda a2, c: 0.5 * torch.sqrt(1 + a2 * (1 - c ** 2) / (c ** 2 + 1e-7)) - 0.5
        geo = 1.0 / (1.0 + lam(rough[rid] ** 2, NoV) + lam(rough[rid] ** 2, NoL))
    else:
        geo = schlick_g1(NoV, rough[rid]) * schlick_g1(NoL, rough[rid])
    NoH = sat_dot(nrm[rid], Hs)
    dist = ggx_d(NoH, rough[rid])
    sl, shit, sinter, shlw = get_lights(sd, tracer, unit_size, pts[rid], sd_, exp_max, poses=human_poses[rid] if human_poses is not None else None,
                                        return_human=True)
    sw = dist * fres * geo / (4 * NoV).clamp_min(EPS)
    specular = segment_coo(sw * sl / sp_.clamp_min(EPS), rid, torch.zeros(pn, 3)) / sn
    colors = linear_to_srgb(diffuse + specular)
    out.update(colors=colors, diffuse_lin=diffuse, specular_lin=specular, metallic=metallic,
               roughness=rough, albedo=albedo, specular_rays_id=rid, specular_mask=smask,
               diffuse_hit=dhit.reshape(pn, dn), specular_hit=shit,
               visibility=1 - segment_coo(shit.to(sl.dtype), rid, torch.zeros(pn))[:, None] / sn,
               indirect_light=segment_coo(sl * shit[:, None].to(sl.dtype), rid, torch.zeros(pn, 3)) / sn,
               diffuse_light=torch.clamp(linear_to_srgb(dl.mean(1)), 0, 1),
               specular_light=torch.clamp(linear_to_srgb(segment_coo(sl, rid, torch.zeros(pn, 3)) / sn), 0, 1),
               diffuse_dirs=ddirs, diffuse_pdf=dpdf, specular_dirs=sdirs, specular_pdf=spdf)
    # the rest of the reference's dict (fields.py:1241-1256, :1288-1291).  approximate_light adds the ALREADY sRGB-encoded, clamped
    # specular colour to the linear diffuse term (:1246-1248: `specular_colors` was reassigned two lines earlier); `variance` is the
    # specular one (:1289 overwrites :1255); variance_diffuse_vis divides by cfg diffuse_sample_num, not by the ray count (:1256)
    gd = (dw * dl).mean(-1, keepdim=True) / dpdf.clamp_min(EPS)
    gs = (sw * sl).mean(-1, keepdim=True) / sp_.clamp_min(EPS)
    spec_srgb = torch.clamp(linear_to_srgb(specular), 0, 1)
    out.update(human_lights=shlw.reshape(-1, 3), inter=sinter,
               approximate_light=torch.clamp(linear_to_srgb(torch.mean(kd * dl, 1) + spec_srgb), 0, 1),
               variance=torch.var(gs), variance_diffuse_vis=torch.var(gd, dim=1, unbiased=True) / n_fixed_diffuse,
               variance_specular_vis=(segment_coo(gs ** 2, rid, torch.zeros(pn, 1)) / sn - (segment_coo(gs, rid, torch.zeros(pn, 1)) / sn) ** 2) / sn)
    return out
