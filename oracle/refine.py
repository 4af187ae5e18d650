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
