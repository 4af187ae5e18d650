"""Oracle: VM-decomposed tensorial field lookup + TensoSDF decoder (TEST INFRASTRUCTURE).

Follows network/fields.py:262-299 (TensoSDF.forward), :227-260 (gradient),
network/fields.py:776-810 (MCShadingNetwork.tenso_feature), network/flow.py:709-744
(TensoFlow.tenso_feature) and utils/network_utils.py:90-91 (contraction).

Parameters are passed as a `state_dict`-style mapping with the reference's key names
(`sdf_plane.{0,1,2}` [1,C,H,W], `sdf_line.{0,1,2}` [1,C,L,1], `sdf_mat.{0,2}.{weight,bias}`).
"""
import torch
import torch.nn.functional as F

from . import texture as tex

MAT_MODE = ((0, 1), (0, 2), (1, 2))   # fields.py:28
VEC_MODE = (2, 1, 0)                  # fields.py:29


def contraction(xyz, aabb):
    return (xyz - aabb[0]) / (aabb[1] - aabb[0])


def vm_gather(planes, lines, xyz01, level, n_levels):
    """planes: 3 x [1,C,H,W]; lines: 3 x [1,C,L,1]; xyz01 [N,3] in aabb-normalised
    coordinates; level [N] or None  ->  (plane_feat [N,3C], line_feat [N,3C]).

    Plane i is sampled at uv = (xyz[MAT_MODE[i][0]], xyz[MAT_MODE[i][1]]) (u -> W axis,
    v -> H axis); line i at uv = (0, xyz[VEC_MODE[i]]) on an [L,1] texture.
    """
    N = xyz01.shape[0]
    lv = torch.zeros(N, dtype=xyz01.dtype) if level is None else level.reshape(-1)
    pf, lf = [], []
    for i in range(3):
        p = planes[i].permute(0, 2, 3, 1)     # [1,H,W,C]
        l = lines[i].permute(0, 2, 3, 1)      # [1,L,1,C]
        uvp = torch.stack((xyz01[:, MAT_MODE[i][0]], xyz01[:, MAT_MODE[i][1]]), -1).detach()
        uvl = torch.stack((torch.zeros(N, dtype=xyz01.dtype), xyz01[:, VEC_MODE[i]]), -1).detach()
        pf.append(tex.texture(p, uvp[None, :, None, :], mip_level_bias=lv[None, :, None],
                              boundary_mode="clamp", max_mip_level=n_levels - 1)[0, :, 0, :])
        lf.append(tex.texture(l, uvl[None, :, None, :], mip_level_bias=lv[None, :, None],
                              boundary_mode="clamp", max_mip_level=n_levels - 1)[0, :, 0, :])
    return torch.cat(pf, -1), torch.cat(lf, -1)


def vm_feature(planes, lines, xyz, aabb, level, n_levels):
    pf, lf = vm_gather(planes, lines, contraction(xyz, aabb).reshape(-1, 3), level, n_levels)
    return pf * lf


def softplus100(x):
    return F.softplus(x, beta=100)


def sdf_forward(sd, xyz, level, aabb, n_levels, prefix=""):
    """TensoSDF.forward (network/fields.py:262-299) -> [N, 1+app_dim].  sdf_multires = m is read off the first layer's width
    (3C + 3 + 6m inputs): m > 0 appends get_embedder(m)(x) in place of the raw point -- of the CONTRACTED point when m == 3, of the raw
    one otherwise (:294)."""
    planes = [sd[f"{prefix}sdf_plane.{i}"] for i in range(3)]
    lines = [sd[f"{prefix}sdf_line.{i}"] for i in range(3)]
    feat = vm_feature(planes, lines, xyz, aabb, level, n_levels)
    m = (sd[f"{prefix}sdf_mat.0.weight"].shape[1] - feat.shape[1] - 3) // 6
    pos = xyz
    if m > 0:
        from .encodings import posenc
        pos = posenc(contraction(xyz, aabb).reshape(-1, 3) if m == 3 else xyz, m)
    h = torch.cat([feat, pos], -1)
    h = F.linear(h, sd[f"{prefix}sdf_mat.0.weight"], sd[f"{prefix}sdf_mat.0.bias"])
    h = softplus100(h)
    return F.linear(h, sd[f"{prefix}sdf_mat.2.weight"], sd[f"{prefix}sdf_mat.2.bias"])


def sdf_units(aabb, grid_size):
    return (aabb[1] - aabb[0]) / (torch.as_tensor(grid_size, dtype=torch.float32) - 1)


def sdf_gradient(sd, xyz, level, aabb, n_levels, grid_size, sdf=None, training=False, prefix=""):
    """Central differences with eps = aabbSize/(R-1)  (fields.py:227-260)."""
    eps = sdf_units(aabb, grid_size)
    taps = []
    for ax in range(3):
        e = torch.zeros(3)
        e[ax] = eps[ax]
        sp = sdf_forward(sd, xyz + e, level, aabb, n_levels, prefix)[..., :1]
        sn = sdf_forward(sd, xyz - e, level, aabb, n_levels, prefix)[..., :1]
        taps.append((sp, sn))
    grad = torch.cat([(sp - sn) / (2 * eps[ax]) for ax, (sp, sn) in enumerate(taps)], -1)
    nh = None
    if training:
        hess = torch.cat([(sp + sn - 2 * sdf) / (eps[ax] ** 2) for ax, (sp, sn) in enumerate(taps)], -1)
        nh = (grad * hess).sum(-1) / ((grad ** 2).sum(-1) + 1e-5)
    return grad, nh
