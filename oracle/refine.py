"""Oracle: SDF refinement of mesh hits (TEST INFRASTRUCTURE).

Follows network/materialRenderer.py:281-343 (get_intersection_around_mesh, trace_sdf_with_mesh), :345-355
(near_far_from_sphere) and utils/network_utils.py:149-170 (get_weights), :117-147 (sample_pdf, det=True):
around the BVH hit depth, 32 uniform t in depth +- 4 voxels -> NeuS weights -> 9 importance samples -> weights
again -> weighted-mean depth; normal = normalised FD gradient of the SDF, flipped to face the ray.
"""
import torch
import torch.nn.functional as F

from .march import near_far_from_sphere, sample_pdf_det
from .vm_field import sdf_forward, sdf_gradient


def neus_weights(sdf_fn, inv_s, z, o, d):
    """get_weights -> weights [pn,sn-1], mid_sdf [pn,sn-1]."""
    pts = z[..., None] * d[:, None] + o[:, None]
    pn, sn = z.shape
    sdf = sdf_fn(pts.reshape(-1, 3)).reshape(pn, sn)
    ps, ns = sdf[:, :-1], sdf[:, 1:]
    pz, nz = z[:, :-1], z[:, 1:]
    mid = (ps + ns) * 0.5
    cos = (ns - ps) / (nz - pz + 1e-5)
    surf = cos < 0
    cos = cos.clamp(max=0)
    dist = nz - pz
    pc = torch.sigmoid((mid - cos * dist * 0.5) * inv_s)
    nc = torch.sigmoid((mid + cos * dist * 0.5) * inv_s)
    alpha = (pc - nc + 1e-5) / (pc + 1e-5) * surf.float()
    w = alpha * torch.cumprod(torch.cat([torch.ones(pn, 1), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    mid = torch.where(surf, mid, -torch.ones_like(mid))
    return w, mid


def refine_hits(sd, tracer, o, d, aabb, grid_size, n_levels, inv_s, unit_size, sn0=32, sn1=9, prefix="sdf_network."):
    """trace_sdf_with_mesh -> inters [rn,3], normals [rn,3], depth [rn,1], hit [rn,1] bool."""
    inters, normals, depth, hit = tracer(o, d)
    inters, normals, depth = inters.clone(), normals.clone(), depth.clone()
    if hit.any():
        oo, dd, md = o[hit], d[hit], depth[hit]
        near, far = near_far_from_sphere(oo, dd)
        tmin = torch.maximum(torch.minimum(md - unit_size * 4, far), near)
        tmax = torch.maximum(torch.minimum(md + unit_size * 4, far), near)
        sdf_fn = lambda p: sdf_forward(sd, p, None, aabb, n_levels, prefix)[:, 0]
        z = tmin + (tmax - tmin) * torch.linspace(0.0, 1.0, sn0)[None]
        w, _ = neus_weights(sdf_fn, inv_s, z, oo, dd)
        z_new, _ = sample_pdf_det(z, w, sn1)
        w, _ = neus_weights(sdf_fn, inv_s, z_new, oo, dd)
        z_mid = (z_new[:, 1:] + z_new[:, :-1]) * 0.5
        w = w / w.sum(-1, keepdim=True)
        w = torch.where(torch.isnan(w), torch.full_like(w, 1.0 / (sn1 - 1)), w)
        dep = (w * z_mid).sum(-1, keepdim=True)
        depth[hit] = dep
        p = oo + dep * dd
        inters[hit] = p
        g, _ = sdf_gradient(sd, p, None, aabb, n_levels, grid_size, prefix=prefix)
        n = F.normalize(g, dim=-1)
        flip = (n * dd).sum(-1) >= 0
        n = torch.where(flip[:, None], -n, n)
        normals[hit] = n
    return inters, normals, depth, hit[:, None]


def material_nvs(shader_sd, sdf_sd, tracer, pose, K, h, w, aabb, grid_size, n_levels, inv_s, unit_size, sn_fixed, sn_flow,
                 n_fixed_specular):
    """MaterialRenderer.nvs (materialRenderer.py:641-752) with nerfDataType rays (:647-672): pixel grid -> camera rays (no half-pixel
    offset, unit directions) -> trace_sdf_with_mesh(32, 9) -> MCShadingNetwork.forward(step=None) on the pixels that see the object
    (the un-suffixed maps come from the fixed-sampler pass) -> 15 maps [h,w,C]: white background (:743), normal (0,0,1) on pixels that
    miss in 512-ray chunks with a hit (:725), sqrt of the squared roughness (:739), the four variance maps left at zero (their assignments are commented out,
    :732-735).  sn_fixed = cfg diffuse_sample_num, sn_flow = (nis_diffuse_sample_num, nis_specular_sample_num)."""
    from . import shading as osh
    pose = torch.as_tensor(pose, dtype=torch.float32)
    K = torch.as_tensor(K, dtype=torch.float32)
    i, j = torch.meshgrid(torch.linspace(0, w - 1, w), torch.linspace(0, h - 1, h), indexing="ij")
    i, j = i.t(), j.t()
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1).reshape(-1, 3)
    rays_d = F.normalize((pose[:3, :3] @ dirs.t()).t(), dim=-1)
    rays_o = pose[:3, 3].expand(h * w, 3)
    inters, normals, depth, hit = refine_hits(sdf_sd, tracer, rays_o, rays_d, aabb, grid_size, n_levels, inv_s, unit_size)
    hit = hit[:, 0]
    rn = h * w
    out = {k: torch.zeros(rn, c) for k, c in (("color", 3), ("normal", 3), ("spec_light", 3), ("diff_light", 3), ("indirect_light", 3),
                                              ("spec_color", 3), ("diff_color", 3), ("albedo", 3), ("roughness", 1), ("metallic", 1),
                                              ("occ_trace", 1), ("variance_diffuse_vis", 1), ("variance_specular_vis", 1),
                                              ("variance_diffuse_vis_nis", 1), ("variance_specular_vis_nis", 1))}
    out["color"][:] = 1.0
    # (:725 sits inside `if torch.sum(hit_mask) > 0` of the 512-ray chunk loop: pixels that miss get the normal (0,0,1) only in chunks
    # that contain at least one hit; in a chunk without any they stay zero)
    for c0 in range(0, rn, 512):
        if hit[c0:c0 + 512].any():
            out["normal"][c0:c0 + 512, 2] = 1.0
    if hit.any():
        sh = osh.shade(shader_sd, tracer, unit_size, aabb, inters[hit], -rays_d[hit], normals[hit], sn_flow[0], sn_flow[1],
                       n_fixed_diffuse=sn_fixed, n_fixed_specular=n_fixed_specular, use_flow=False)
        c01 = lambda t: torch.clamp(osh.linear_to_srgb(t), 0, 1)
        out["color"][hit] = sh["colors"]
        out["normal"][hit] = normals[hit]
        out["spec_light"][hit], out["diff_light"][hit] = sh["specular_light"], sh["diffuse_light"]
        out["indirect_light"][hit], out["occ_trace"][hit] = sh["indirect_light"], sh["visibility"]
        out["spec_color"][hit], out["diff_color"][hit] = c01(sh["specular_lin"]), c01(sh["diffuse_lin"])
        out["albedo"][hit], out["metallic"][hit] = sh["albedo"], sh["metallic"]
        out["roughness"][hit] = torch.sqrt(sh["roughness"])
    return {k: v.reshape(h, w, -1) for k, v in out.items()}, (inters, normals, depth, hit)
