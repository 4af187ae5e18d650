"""Oracle restatement of `nvdiffrast.torch.texture` (TEST INFRASTRUCTURE).

nvdiffrast is a third-party CUDA extension that is NOT vendored under /root/reference
and is not version-pinned by the reference.  The reference calls it
at network/fields.py:276-288, :790-802, network/flow.py:723-735 (2-D, clamp, mip bias),
network/fields.py:522 (2-D linear clamp), network/light.py:107,111-118,135 (cube).
This file restates its published filtering rules with differentiable torch ops:

  * texel (i, j) of an [H, W] texture has its centre at uv = ((j + .5) / W, (i + .5) / H);
    a sample at uv is taken at texel coordinate (u*W - .5, v*H - .5), bilinear, indices
    clamped to the edge for boundary_mode="clamp" (== grid_sample(align_corners=False,
    padding_mode="border"));
  * the mip chain is a 2x2 box filter (axes of size 1 stay 1 and average 2x1);
    dimensions > 1 must stay even until `max_mip_level` is reached;
  * with `mip_level_bias` and no `uv_da` the level is the bias itself, clamped to
    [0, max level]; l0 = floor(level), f = level - l0, result = (1-f)*L[l0] + f*L[l0+1]
    (second level fetched only when f > 0);
  * cube maps: major-axis face selection, GL face order +x,-x,+y,-y,+z,-z with the face
    parametrisation of network/light_utils.py:24-31; bilinear taps that leave the face are
    re-projected onto the neighbouring face; the (at most one) tap that falls off a cube
    corner is dropped and the remaining three weights renormalised.

PARITY UNPINNED: no reference test exercises dr.texture and the real extension cannot run
here (no CUDA).  `tests/test_oracle_texture.py` cross-checks the 2-D path against
`F.grid_sample` and the cube path against a brute-force direction re-projection.
"""
import torch

__all__ = ["texture", "box_mips", "bilinear_2d", "cube_face_uv", "cube_bilinear"]


def box_mips(tex, n_levels, cube=False):
    """tex [..., H, W, C] -> list of n_levels tensors (level 0 = tex)."""
    out = [tex]
    for _ in range(1, n_levels):
        t = out[-1]
        H, W = t.shape[-3], t.shape[-2]
        if (H > 1 and H % 2) or (W > 1 and W % 2):
            raise ValueError(f"cannot build mip of odd texture {H}x{W}")
        if H > 1:
            t = 0.5 * (t[..., 0::2, :, :] + t[..., 1::2, :, :])
        if W > 1:
            t = 0.5 * (t[..., :, 0::2, :] + t[..., :, 1::2, :])
        out.append(t)
        if t.shape[-3] == 1 and t.shape[-2] == 1:
            break
    return out


def bilinear_2d(tex, uv, boundary="clamp"):
    """tex [H, W, C]; uv [M, 2] -> [M, C]."""
    H, W, C = tex.shape
    u = uv[:, 0] * W - 0.5
    v = uv[:, 1] * H - 0.5
    if boundary == "wrap":
        u = u - torch.floor(uv[:, 0]) * W
        v = v - torch.floor(uv[:, 1]) * H
    iu0 = torch.floor(u)
    iv0 = torch.floor(v)
    fu = (u - iu0).unsqueeze(-1)
    fv = (v - iv0).unsqueeze(-1)
    iu0 = iu0.long()
    iv0 = iv0.long()
    iu1, iv1 = iu0 + 1, iv0 + 1
    if boundary == "clamp":
        iu0, iu1 = iu0.clamp(0, W - 1), iu1.clamp(0, W - 1)
        iv0, iv1 = iv0.clamp(0, H - 1), iv1.clamp(0, H - 1)
    elif boundary == "wrap":
        iu0, iu1, iv0, iv1 = iu0 % W, iu1 % W, iv0 % H, iv1 % H
    else:
        raise NotImplementedError(boundary)
    flat = tex.reshape(H * W, C)
    t00 = flat[iv0 * W + iu0]
    t10 = flat[iv0 * W + iu1]
    t01 = flat[iv1 * W + iu0]
    t11 = flat[iv1 * W + iu1]
    top = t00 * (1 - fu) + t10 * fu
    bot = t01 * (1 - fu) + t11 * fu
    return top * (1 - fv) + bot * fv


def cube_face_uv(d):
    """d [M,3] -> face [M] long, x [M], y [M] in [-1,1] (light_utils.py:24-31 inverted)."""
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    ax, ay, az = dx.abs(), dy.abs(), dz.abs()
    is_z = az > torch.maximum(ax, ay)
    is_y = (~is_z) & (ay > ax)
    is_x = ~(is_z | is_y)
    c = torch.where(is_z, dz, torch.where(is_y, dy, dx))
    face = torch.where(is_z, 4, torch.where(is_y, 2, 0)) + (c < 0).long()
    m = 1.0 / c.abs()
    # per-face (x, y)
    x = torch.where(is_x, torch.where(c < 0, dz, -dz),
                    torch.where(is_y, dx, torch.where(c < 0, -dx, dx))) * m
    y = torch.where(is_x, -dy,
                    torch.where(is_y, torch.where(c < 0, -dz, dz), -dy)) * m
    return face, x, y


def _face_dir(face, x, y):
    """inverse of cube_face_uv (un-normalised direction); face long [M], x,y [M]."""
    one = torch.ones_like(x)
    dirs = [
        torch.stack((one, -y, -x), -1),
        torch.stack((-one, -y, x), -1),
        torch.stack((x, one, y), -1),
        torch.stack((x, -one, -y), -1),
        torch.stack((x, -y, one), -1),
        torch.stack((-x, -y, -one), -1),
    ]
    out = torch.zeros(x.shape[0], 3, dtype=x.dtype)
    for s in range(6):
        out = torch.where((face == s).unsqueeze(-1), dirs[s], out)
    return out


def cube_bilinear(tex, d):
    """tex [6, R, R, C]; d [M, 3] (need not be normalised) -> [M, C]."""
    six, R, R2, C = tex.shape
    assert six == 6 and R == R2
    face, x, y = cube_face_uv(d)
    u = (x * 0.5 + 0.5) * R - 0.5
    v = (y * 0.5 + 0.5) * R - 0.5
    iu0 = torch.floor(u)
    iv0 = torch.floor(v)
    fu = u - iu0
    fv = v - iv0
    flat = tex.reshape(6 * R * R, C)
    acc = torch.zeros(d.shape[0], C, dtype=tex.dtype)
    wsum = torch.zeros(d.shape[0], dtype=tex.dtype)
    for dv in (0, 1):
        for du in (0, 1):
            iu = iu0 + du
            iv = iv0 + dv
            w = (fu if du else 1 - fu) * (fv if dv else 1 - fv)
            out_u = (iu < 0) | (iu > R - 1)
            out_v = (iv < 0) | (iv > R - 1)
            corner = out_u & out_v
            # texel centre on the (extended) face plane -> direction -> owning face/texel
            tx = (iu + 0.5) / R * 2 - 1
            ty = (iv + 0.5) / R * 2 - 1
            f2, x2, y2 = cube_face_uv(_face_dir(face, tx, ty))
            inside = ~(out_u | out_v)
            f2 = torch.where(inside, face, f2)
            ju = torch.where(inside, iu, torch.floor((x2 * 0.5 + 0.5) * R)).clamp(0, R - 1).long()
            jv = torch.where(inside, iv, torch.floor((y2 * 0.5 + 0.5) * R)).clamp(0, R - 1).long()
            w = torch.where(corner, torch.zeros_like(w), w)
            acc = acc + w.unsqueeze(-1) * flat[(f2 * R + jv) * R + ju]
            wsum = wsum + w
    return acc / wsum.unsqueeze(-1)


def _mip_lerp(levels, fetch, level, max_level):
    """levels: list of textures; fetch(tex)->[M,C]; level [M] float or None."""
    if level is None or len(levels) == 1:
        return fetch(levels[0])
    max_level = min(max_level, len(levels) - 1)
    level = level.clamp(0.0, float(max_level))
    l0 = torch.floor(level).long().clamp(max=max_level)
    f = (level - l0.to(level.dtype)).unsqueeze(-1)
    l1 = (l0 + 1).clamp(max=max_level)
    out = None
    for li, tex in enumerate(levels[: max_level + 1]):
        need = ((l0 == li) | ((l1 == li) & (f[:, 0] > 0)))
        if not bool(need.any()):
            continue
        val = fetch(tex)
        w = torch.where((l0 == li).unsqueeze(-1), 1 - f, torch.zeros_like(f)) + \
            torch.where(((l1 == li) & (l0 != li)).unsqueeze(-1), f, torch.zeros_like(f))
        out = val * w if out is None else out + val * w
    return out


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto",
            boundary_mode="wrap", max_mip_level=None):
    """Drop-in for the subset of nvdiffrast.torch.texture the reference uses."""
    if uv_da is not None:
        raise NotImplementedError("uv_da is never passed on the reference path")
    if filter_mode == "auto":
        filter_mode = "linear-mipmap-linear" if mip_level_bias is not None else "linear"
    cube = boundary_mode == "cube"
    B = tex.shape[0]
    if B != 1:
        raise NotImplementedError("reference always passes minibatch 1")
    out_shape = uv.shape[:-1]
    uvf = uv.reshape(-1, uv.shape[-1])
    t0 = tex[0]
    level = None
    levels = [t0]
    max_level = 0
    if filter_mode == "linear-mipmap-linear":
        level = mip_level_bias.reshape(-1)
        if mip is not None:
            levels = [t0] + [m[0] for m in mip]
            max_level = len(levels) - 1
            if max_mip_level is not None:
                max_level = min(max_level, max_mip_level)
        else:
            if max_mip_level is None:
                n = 1
                h, w = t0.shape[-3], t0.shape[-2]
                while (h > 1 or w > 1) and not ((h > 1 and h % 2) or (w > 1 and w % 2)):
                    h, w = max(h // 2, 1), max(w // 2, 1)
                    n += 1
                max_level = n - 1
            else:
                max_level = max_mip_level
            levels = box_mips(t0, max_level + 1, cube=cube)
            max_level = len(levels) - 1
    elif filter_mode != "linear":
        raise NotImplementedError(filter_mode)
    if cube:
        fetch = lambda t: cube_bilinear(t, uvf)
    else:
        fetch = lambda t: bilinear_2d(t, uvf, boundary_mode)
    out = _mip_lerp(levels, fetch, level, max_level)
    return out.reshape(*out_shape, tex.shape[-1])
