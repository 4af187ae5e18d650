"""Oracle: NeuS-style ray-march over the TensoSDF field (TEST INFRASTRUCTURE).

Follows network/shapeRenderer.py:
  near_far_from_sphere :676-684, upsample :820-849, cat_z_vals :851-869, sample_ray :871-932,
  compute_ball_radii :966-970, compute_sdf_alpha :995-1025, render_core :1105-1277 (train branch);
utils/network_utils.py:117-147 (sample_pdf); network/other_field.py:193-207 (inv_s);
network/fields.py:419-567 (ShapeShadingNetwork.forward, split-sum shading);
network/light.py:72-80,95-122 (EnvLight.get_mip / __call__ on an injected pre-filtered stack).

State-dict keys are those of ShapeRenderer: `sdf_network.*`, `deviation_network.variance`,
`color_network.*`.  Eval/perturb=0 sampling; stochastic pieces are out of the oracle.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import texture as tex
from .encodings import ide5, linear_to_srgb, mlp, posenc
from .segments import accumulate_along_rays, render_weight_from_alpha
from .vm_field import sdf_forward, sdf_gradient, sdf_units


def near_far_from_sphere(o, d, radius=1.0):
    a = (d ** 2).sum(-1, keepdim=True)
    b = 2.0 * (o * d).sum(-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return (mid - radius).clamp(min=1e-3), mid + radius


def ball_radii(t, radii, cos):
    inv = 1.0 / cos
    tmp = (inv * inv - 1).sqrt() - radii
    return t * radii * cos / (tmp * tmp + 1.0).sqrt()


def sample_pdf_det(bins, weights, n):
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0.5 / n, 1.0 - 0.5 / n, steps=n).expand(list(cdf.shape[:-1]) + [n]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b0 + (u - c0) / den * (b1 - b0), inds


def _upsample(o, d, z, sdf, n_imp, inv_s):
    pts = o[:, None] + d[:, None] * z[..., None]
    rad = pts.norm(dim=-1)
    inside = (rad[:, :-1] < 1.0) | (rad[:, 1:] < 1.0)
    ps, ns = sdf[:, :-1], sdf[:, 1:]
    pz, nz = z[:, :-1], z[:, 1:]
    mid = (ps + ns) * 0.5
    cos = (ns - ps) / (nz - pz + 1e-5)
    prev = torch.cat([torch.zeros(z.shape[0], 1), cos[:, :-1]], -1)
    cos = torch.minimum(prev, cos).clip(-1e3, 0.0) * inside
    dist = nz - pz
    pc = torch.sigmoid((mid - cos * dist * 0.5) * inv_s)
    nc = torch.sigmoid((mid + cos * dist * 0.5) * inv_s)
    alpha = (pc - nc + 1e-5) / (pc + 1e-5)
    w = alpha * torch.cumprod(torch.cat([torch.ones(z.shape[0], 1), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    return sample_pdf_det(z, w, n_imp)[0]


def sample_ray(sd, o, d, near, far, radiis, rays_cos, aabb, grid_size, n_levels, base_radii,
               n_samples=64, n_importance=64, up_steps=4):
    """perturb=0, clip_sample_variance=False  -> t_starts, t_ends, ray_indices (int64), all packed."""
    rn = o.shape[0]
    sdf_fn = lambda p, lv: sdf_forward(sd, p, lv, aabb, n_levels, "sdf_network.")[:, 0]
    vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
    ra, rb = (aabb[1] - o) / vec, (aabb[0] - o) / vec
    tmin = torch.minimum(ra, rb).amax(-1).clamp(min=near[:, 0], max=far[:, 0])[:, None]
    tmax = torch.maximum(ra, rb).amin(-1).clamp(min=near[:, 0], max=far[:, 0])[:, None]
    t = tmin + (tmax - tmin) * torch.linspace(0.0, 1.0, n_samples)[None]
    pts = o[:, None] + d[:, None] * t[..., None]
    lv = torch.log2(ball_radii(t[..., None], radiis[:, None], rays_cos[:, None]) / base_radii)
    sdf = sdf_fn(pts.reshape(-1, 3), lv.reshape(-1, 1)).reshape(rn, n_samples)
    for i in range(up_steps):
        inv_s = torch.ones(rn, t.shape[1] - 1) * 64 * 2 ** i
        new_t = _upsample(o, d, t, sdf, n_importance // up_steps, inv_s)
        last = i + 1 == up_steps
        npts = o[:, None] + d[:, None] * new_t[..., None]
        nlv = torch.log2(ball_radii(new_t[..., None], radiis[:, None], rays_cos[:, None]) / base_radii)
        t_all, index = torch.sort(torch.cat([t, new_t], -1), -1)
        if not last:
            nsdf = sdf_fn(npts.reshape(-1, 3), nlv.reshape(-1, 1)).reshape(rn, -1)
            sdf = torch.gather(torch.cat([sdf, nsdf], -1), -1, index)
        t = t_all
    dists = t[:, 1:] - t[:, :-1]
    dists = torch.cat([dists, dists[:, -1:]], -1)
    mid = t + dists * 0.5
    ridx = torch.arange(rn)[:, None].expand(rn, t.shape[1])
    p = o[:, None] + d[:, None] * mid[..., None]
    inner = ~((aabb[0] > p) | (p > aabb[1])).any(-1)
    return t[inner], (t + dists)[inner], ridx[inner]


def sdf_alpha(sd, pts, level, dists, dirs, cos_anneal, aabb, grid_size, n_levels, training=True):
    """compute_sdf_alpha -> alpha, grad, feat, inv_s, sdf, normal_hessian."""
    out = sdf_forward(sd, pts, level, aabb, n_levels, "sdf_network.")
    sdf, feat = out[:, 0], out[:, 1:]
    grad, nh = sdf_gradient(sd, pts, level, aabb, n_levels, grid_size, sdf=sdf[:, None], training=training,
                            prefix="sdf_network.")
    inv_s = (torch.ones(pts.shape[0]) * torch.exp(sd["deviation_network.variance"] * 10.0)).clip(1e-6, 1e6)
    true_cos = (dirs * grad).sum(-1)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal) + F.relu(-true_cos) * cos_anneal)
    pc = torch.sigmoid((sdf - iter_cos * dists * 0.5) * inv_s)
    nc = torch.sigmoid((sdf + iter_cos * dists * 0.5) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)
    return alpha, grad, feat, inv_s, sdf, nh


def env_lookup(levels, d, rough=None, diffuse=None, min_r=0.08, max_r=0.5):
    """EnvLight.__call__ (light.py:95-122) on an injected pre-filtered stack."""
    if rough is None:
        return torch.exp(tex.cube_bilinear(diffuse, d))
    n = len(levels)
    r = rough[:, 0]
    mip = torch.where(r < max_r, (r.clamp(min_r, max_r) - min_r) / (max_r - min_r) * (n - 2),
                      (r.clamp(max_r, 1.0) - max_r) / (1.0 - max_r) + n - 2)
    val = tex._mip_lerp(levels, lambda t: tex.cube_bilinear(t, d), mip, n - 1)
    return torch.exp(val)


def shape_shade(sd, env, fg_lut, pts, normals, view, feat, inter_results=False):
    """ShapeShadingNetwork.forward (no radiance field, no human light) -> color, occ_prob, roughness, reflective
    (+ the `inter_results` dictionary of fields.py:541-560 when asked)."""
    P = "color_network."
    normals = F.normalize(normals, dim=-1).clone()
    normals[normals[:, :2].sum(-1) == 0.0] = torch.tensor([0.0, 1e-6, 1.0])
    view = F.normalize(view, dim=-1)
    refl = (view * normals).sum(-1, keepdim=True) * normals * 2 - view
    NoV = (normals * view).sum(-1, keepdim=True)
    mat = mlp(sd, P + "mat_mlp", (0, 2, 4), feat, F.relu, torch.sigmoid)
    albedo, rough, metal = mat[:, :3] * 0.77 + 0.03, mat[:, 3:4] * 0.9 + 0.09, mat[:, 4:]
    diffuse = (1 - metal) * albedo * env_lookup(None, normals, diffuse=env["diffuse"])
    spec_alb = 0.04 * (1 - metal) + metal * albedo
    ref_r = ide5(refl, rough)
    direct = env_lookup(env["specular"], refl, rough)
    pe = posenc(pts, 8)
    expo = lambda t: torch.exp(t.clamp(max=0.0))
    indirect = mlp(sd, P + "inner_light", (0, 2, 4), torch.cat([pe, ref_r], -1), F.relu, expo)
    occ = mlp(sd, P + "inner_weight", (0, 2, 4), torch.cat([pe, posenc(refl, 6)], -1), F.relu) * 0.5 + 0.5
    occ_c = occ.clamp(0, 1)
    light = indirect * occ_c + direct * (1 - occ_c)
    uv = torch.cat([NoV.clamp(0, 1), rough.clamp(0, 1)], -1)
    fg = tex.bilinear_2d(fg_lut[0], uv, "clamp")
    spec = (spec_alb * fg[:, 0:1] + fg[:, 1:2]) * light
    color = linear_to_srgb(diffuse + spec).clamp(0.0, 1.0)
    if inter_results:
        c01 = lambda t: t.clamp(0.0, 1.0)
        diff_light = env_lookup(None, normals, diffuse=env["diffuse"])
        inter = dict(specular_albedo=spec_alb, specular_ref=c01(spec_alb * fg[:, 0:1] + fg[:, 1:2]), specular_direct_light=direct,
                     specular_light=c01(linear_to_srgb(light)), specular_color=c01(linear_to_srgb(spec)), diffuse_albedo=(1 - metal) * albedo,
                     diffuse_light=c01(linear_to_srgb(diff_light)), diffuse_color=c01(linear_to_srgb(diffuse)), metallic=metal, roughness=rough,
                     albedo=albedo, occ_prob=occ_c, indirect_light=indirect * occ_c)      # fields.py:438: the indirect light is returned weighted
        return color, occ, rough, refl, inter
    return color, occ, rough, refl


def render_core(sd, env, fg_lut, o, d, radiis, rays_cos, t0, t1, ridx, aabb, grid_size, n_levels, base_radii,
                cos_anneal):
    """Train branch of render_core, white background -> dict."""
    rn = o.shape[0]
    mid = (t0 + t1) * 0.5
    dists = t1 - t0
    pts = o[ridx] + d[ridx] * mid[:, None]
    lv = torch.log2(ball_radii(mid[:, None], radiis[ridx], rays_cos[ridx]) / base_radii)
    alpha, grad, feat, inv_s, sdf, nh = sdf_alpha(sd, pts, lv, dists, d[ridx], cos_anneal, aabb, grid_size, n_levels)
    color, occ, rough, refl = shape_shade(sd, env, fg_lut, pts, F.normalize(grad, dim=-1), -d[ridx], feat)
    w, _ = render_weight_from_alpha(alpha, ray_indices=ridx, n_rays=rn)
    acc = accumulate_along_rays(w, None, ridx, rn)
    rgb = accumulate_along_rays(w, color, ridx, rn) + (1 - acc)
    nrm = accumulate_along_rays(w, grad, ridx, rn)
    nrm = F.normalize(nrm * acc + (1.0 - acc) * torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    return dict(ray_rgb=rgb, acc=acc, normal=nrm, gradient_error=(grad.norm(dim=-1) - 1.0) ** 2,
                std=torch.mean(1 / inv_s), loss_sparse=torch.exp(-20.0 * sdf.abs()).mean(),
                loss_hessian=nh.abs().mean(), alpha=alpha, weights=w, color=color, sdf=sdf, grad=grad)


def _neus_weights(sd, inv_s, z, p, d, aabb, n_levels):
    """get_weights (utils/network_utils.py:149-170): NeuS weights of the sections of z along (p, d); field level None (sdf_inter_fun)."""
    pn, sn = z.shape
    pts = z[..., None] * d[:, None, :] + p[:, None, :]
    sdf = sdf_forward(sd, pts.reshape(-1, 3), None, aabb, n_levels, "sdf_network.")[:, 0].reshape(pn, sn)
    prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
    prev_z, next_z = z[:, :-1], z[:, 1:]
    mid_sdf = (prev_sdf + next_sdf) * 0.5
    cos_val = (next_sdf - prev_sdf) / (next_z - prev_z + 1e-5)
    surface = cos_val < 0
    cos_val = cos_val.clamp(max=0)
    dist = next_z - prev_z
    prev_cdf = torch.sigmoid((mid_sdf - cos_val * dist * 0.5) * inv_s)
    next_cdf = torch.sigmoid((mid_sdf + cos_val * dist * 0.5) * inv_s)
    alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5) * surface.float()
    return alpha * torch.cumprod(torch.cat([torch.ones(pn, 1), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]


def traced_occlusion(sd, pts, dirs, aabb, n_levels, sn0=128, sn1=9):
    """get_intersection (utils/network_utils.py:172-202) -> sum of the hit weights along (pts, dirs) up to the unit sphere [pn,1]
    (0 for points outside r = 0.999): the `occ_prob_gt` of render_core's validation branch (shapeRenderer.py:1268-1270)."""
    occ = torch.zeros(pts.shape[0], 1)
    inside = pts.norm(dim=-1) < 0.999
    if bool(inside.any()):
        p, d = pts[inside], dirs[inside]
        dtx, xtx = (p * d).sum(-1, keepdim=True), (p ** 2).sum(-1, keepdim=True)
        max_dist = -dtx + torch.sqrt((dtx ** 2 - xtx + 1).clamp(min=0) + 1e-6)           # get_sphere_intersection (network_utils.py:93-103)
        inv_s = torch.exp(sd["deviation_network.variance"] * 10.0).clip(1e-6, 1e6)
        z = max_dist * torch.linspace(0, 1, sn0)[None]
        w = _neus_weights(sd, inv_s, z, p, d, aabb, n_levels)
        z_new = sample_pdf_det(z, w, sn1)[0]
        occ[inside] = _neus_weights(sd, inv_s, z_new, p, d, aabb, n_levels).sum(-1, keepdim=True)
    return occ


def render_core_validation(sd, env, fg_lut, o, d, radiis, rays_cos, t0, t1, ridx, aabb, grid_size, n_levels, base_radii, cos_anneal=1.0):
    """Validation branch of ShapeRenderer.render_core (is_train=False, nerfDataType, shapeRenderer.py:1246-1275) on top of the train
    branch's composite: expected depth -> surface point -> re-evaluated finite-difference normal, materials / split-sum lights there
    (the shading net's inter_results, masked to the aabb), traced occlusion along the reflected direction."""
    out = render_core(sd, env, fg_lut, o, d, radiis, rays_cos, t0, t1, ridx, aabb, grid_size, n_levels, base_radii, cos_anneal)
    rn = o.shape[0]
    mid = (t0 + t1) * 0.5
    acc = out["acc"]
    t_depth = accumulate_along_rays(out["weights"], mid[:, None], ridx, rn)
    pts = t_depth * d + o
    level = torch.log2(ball_radii(t_depth, radiis, rays_cos) / base_radii)
    grad, _ = sdf_gradient(sd, pts, level, aabb, n_levels, grid_size, training=False, prefix="sdf_network.")
    normals = F.normalize(grad, dim=-1)
    inner = ~((aabb[0] > pts) | (pts > aabb[1])).any(-1)[:, None]
    feat = sdf_forward(sd, pts, level, aabb, n_levels, "sdf_network.")[:, 1:]
    _, occ, rough, refl, inter = shape_shade(sd, env, fg_lut, pts, normals, -d, feat, inter_results=True)
    val = dict(ray_rgb=out["ray_rgb"], acc=acc, normal=out["normal"], normal_vis=((out["normal"] + 1.0) * 0.5) * acc + (1.0 - acc),
               depth=t_depth * rays_cos, occ_prob_gt=traced_occlusion(sd, pts, refl, aabb, n_levels, 128, 9))
    val.update({k: v * inner for k, v in inter.items()})
    return val


def alpha_mask_sample(volume, aabb, pts):
    """AlphaGridMask.sample_alpha (shapeRenderer.py:78-97): trilinear align_corners=True fetch of a [D,H,W] volume."""
    size = aabb[1] - aabb[0]
    g = (pts - aabb[0]) * (1.0 / size * 2) - 1
    vol = volume.float().view(1, 1, *volume.shape[-3:])
    return F.grid_sample(vol, g.view(1, -1, 1, 1, 3), align_corners=True).view(-1)


def march_uniform(o, d, near, far, aabb, n_steps, step_size=0.0, volume=None, mask_aabb=None, cells=False, t_jitter=None):
    """Fixed-step sampler of the build (stands in for nerfacc's OccGridEstimator.sampling, which is third-party and
    unpinned): slab test / clamp as sample_ray (shapeRenderer.py:878-884); n_steps uniform intervals of [tmin, tmax]
    (step_size <= 0) or intervals of step_size from tmin while t < tmax; a sample is kept when its mid-point is inside the
    aabb and, with an occupancy volume, sample_alpha(mid) > 0.  PARITY: pinned only by the build's own fixtures."""
    rn = o.shape[0]
    vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
    ra, rb = (aabb[1] - o) / vec, (aabb[0] - o) / vec
    tmin = torch.minimum(ra, rb).amax(-1).clamp(min=near.reshape(-1), max=far.reshape(-1))[:, None]
    tmax = torch.maximum(ra, rb).amin(-1).clamp(min=near.reshape(-1), max=far.reshape(-1))[:, None]
    if t_jitter is not None:                       # stratified start: added after the near / far clamp
        tmin = tmin + t_jitter.reshape(-1, 1)
    step = torch.full_like(tmin, step_size) if step_size > 0 else (tmax - tmin) / n_steps
    i = torch.arange(n_steps, dtype=torch.float32)[None]
    t0 = tmin + step * i
    t1 = t0 + step
    mid = (t0 + t1) * 0.5
    p = o[:, None] + d[:, None] * mid[..., None]
    alive = (step > 0) & (tmax > tmin) & (t0 < tmax) & ~((aabb[0] > p) | (p > aabb[1])).any(-1)
    if volume is not None and cells:               # occupancy grid [rx,ry,rz]: the cell that holds the mid-point
        mb = aabb if mask_aabb is None else mask_aabb
        res = torch.tensor(volume.shape[-3:], dtype=torch.float32)
        u = (p - mb[0]) * (1.0 / (mb[1] - mb[0]) * 2) * 0.5
        ijk = torch.minimum(torch.floor(u * res).long().clamp(min=0), (res - 1).long())
        alive = alive & (volume.reshape(*volume.shape[-3:])[ijk[..., 0], ijk[..., 1], ijk[..., 2]] > 0)
    elif volume is not None:
        a = alpha_mask_sample(volume, aabb if mask_aabb is None else mask_aabb, p.reshape(-1, 3)).reshape(rn, n_steps)
        alive = alive & (a > 0)
    ridx = torch.arange(rn)[:, None].expand(rn, n_steps)
    return t0[alive], t1[alive], ridx[alive]


def occ_grid_update(occs, binaries_shape, idx, occ, occ_thre=1e-2, ema_decay=0.95):
    """One OccGridEstimator._update given the evaluated cells `idx` and their opacities `occ` (nerfacc's published rule; nerfacc
    itself is absent: PARITY UNPINNED): occs[idx] = max(occs[idx] * ema_decay, occ); binaries = occs > min(mean(occs), occ_thre)."""
    occs = occs.clone()
    occs[idx] = torch.maximum(occs[idx] * ema_decay, occ)
    thre = torch.clamp(occs[occs >= 0].mean(), max=occ_thre)
    return occs, (occs > thre).view(binaries_shape)
