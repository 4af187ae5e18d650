"""Surface-point finder and full-frame inference of the material stage on the device.

Mirrors MaterialRenderer.trace_sdf_with_mesh / get_intersection_around_mesh (network/materialRenderer.py:281-343),
utils/network_utils.py:149-170 (get_weights) and the chunk loop of MaterialRenderer.nvs (:641-752): primary rays ->
BVH first hit -> SDF refinement (32 + 9 field evaluations per hit ray in tf_sdf_forward, FD normal in
tf_sdf_alpha_fwd) -> flow-sampled shading (MCShader) -> image.  The per-ray bookkeeping of the 9-sample
importance step (cumprod / searchsorted over <= 32 entries) is device-resident torch, as in march.sample_ray.
"""
import torch
import torch.nn.functional as F

from . import ops
from .march import SdfField, _sample_pdf_det, near_far_from_sphere


def _neus_weights(field: SdfField, inv_s, z, o, d):
    pts = z[..., None] * d[:, None] + o[:, None]
    pn, sn = z.shape
    sdf = field.sdf(pts.reshape(-1, 3)).reshape(pn, sn)
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
    return alpha * torch.cumprod(torch.cat([torch.ones(pn, 1, device=z.device), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]


@torch.no_grad()
def trace_sdf_with_mesh(bvh: ops.Bvh, field: SdfField, o, d, inv_s, unit_size, sn0=32, sn1=9):
    """-> inters [rn,3], normals [rn,3], depth [rn,1], hit [rn,1] bool   (materialRenderer.py:316-343)."""
    inters, normals, depth, hit = bvh.trace(o, d)
    depth = depth[:, None]
    idx = torch.nonzero(hit, as_tuple=False)[:, 0]
    if idx.numel() > 0:
        oo, dd, md = o[idx], d[idx], depth[idx]
        near, far = near_far_from_sphere(oo, dd)
        tmin = torch.maximum(torch.minimum(md - unit_size * 4, far), near)
        tmax = torch.maximum(torch.minimum(md + unit_size * 4, far), near)
        z = tmin + (tmax - tmin) * torch.linspace(0.0, 1.0, sn0, device=o.device)[None]
        w = _neus_weights(field, inv_s, z, oo, dd)
        z_new = _sample_pdf_det(z, w, sn1)
        w = _neus_weights(field, inv_s, z_new, oo, dd)
        z_mid = (z_new[:, 1:] + z_new[:, :-1]) * 0.5
        w = w / w.sum(-1, keepdim=True)
        w = torch.where(torch.isnan(w), torch.full_like(w, 1.0 / (sn1 - 1)), w)
        dep = (w * z_mid).sum(-1, keepdim=True)
        p = oo + dep * dd
        zeros = torch.zeros(p.shape[0], device=o.device)
        _, g, _, _, _ = field.sdf_alpha(p, None, zeros, torch.zeros_like(p), 1.0, 0.0, want_feat=False, want_hess=False)
        n = F.normalize(g, dim=-1)
        n = torch.where(((n * dd).sum(-1) >= 0)[:, None], -n, n)
        depth = depth.index_copy(0, idx, dep)
        inters = inters.index_copy(0, idx, p)
        normals = normals.index_copy(0, idx, n)
    return inters, normals, depth, hit[:, None]


@torch.no_grad()
def render_frame(shader, field: SdfField, rays_o, rays_d, inv_s, unit_size, sn_diffuse, sn_specular, chunk=65536):
    """Full-frame material-stage inference (MaterialRenderer.nvs, :705-750): -> dict(color [rn,3], normal, hit, ...).
    Misses are white; `chunk` rays are processed per pass (the reference uses 512: launch-bound)."""
    rn = rays_o.shape[0]
    dev = rays_o.device
    color = torch.ones(rn, 3, device=dev)
    normal = torch.zeros(rn, 3, device=dev)
    normal[:, 2] = 1.0
    albedo = torch.zeros(rn, 3, device=dev)
    hit_all = torch.zeros(rn, dtype=torch.bool, device=dev)
    for s in range(0, rn, chunk):
        o, d = rays_o[s:s + chunk].contiguous(), rays_d[s:s + chunk].contiguous()
        inters, nrm, depth, hit = trace_sdf_with_mesh(shader.bvh, field, o, d, inv_s, unit_size)
        idx = torch.nonzero(hit[:, 0], as_tuple=False)[:, 0]
        if idx.numel() == 0:
            continue
        out = shader.shade(inters[idx], -d[idx], nrm[idx], sn_diffuse, sn_specular)
        color[s + idx] = out["colors"]
        normal[s + idx] = nrm[idx]
        albedo[s + idx] = out["albedo"]
        hit_all[s + idx] = True
    return dict(color=color, normal=normal, albedo=albedo, hit=hit_all)
