"""Synthetic scene pieces used by tests, smoke() and bench.py (no dataset exists offline).

* analytic sphere-union-torus triangle mesh,
* surface points / normals / view directions on it,
* pinhole rays of an 800x800 TensoSDF-synthetic style camera (dataset/database.py:545-559
  conventions: camera on a sphere of radius 2 looking at the origin).
"""
import math

import numpy as np


def uv_sphere(radius=0.5, n_lat=16, n_lon=32, center=(0.0, 0.0, 0.0)):
    verts = [(0.0, 0.0, radius)]
    for i in range(1, n_lat):
        th = math.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * math.pi * j / n_lon
            verts.append((radius * math.sin(th) * math.cos(ph), radius * math.sin(th) * math.sin(ph),
                          radius * math.cos(th)))
    verts.append((0.0, 0.0, -radius))
    faces = []
    for j in range(n_lon):
        faces.append((0, 1 + j, 1 + (j + 1) % n_lon))
    for i in range(n_lat - 2):
        r0, r1 = 1 + i * n_lon, 1 + (i + 1) * n_lon
        for j in range(n_lon):
            a, b = r0 + j, r0 + (j + 1) % n_lon
            c, d = r1 + j, r1 + (j + 1) % n_lon
            faces.append((a, c, d))
            faces.append((a, d, b))
    last = len(verts) - 1
    r0 = 1 + (n_lat - 2) * n_lon
    for j in range(n_lon):
        faces.append((last, r0 + (j + 1) % n_lon, r0 + j))
    v = np.asarray(verts, np.float32) + np.asarray(center, np.float32)
    return v, np.asarray(faces, np.int32)


def torus(R=0.75, r=0.12, n_major=48, n_minor=16, center=(0.0, 0.0, 0.0)):
    verts, faces = [], []
    for i in range(n_major):
        u = 2 * math.pi * i / n_major
        for j in range(n_minor):
            w = 2 * math.pi * j / n_minor
            verts.append(((R + r * math.cos(w)) * math.cos(u), (R + r * math.cos(w)) * math.sin(u), r * math.sin(w)))
    for i in range(n_major):
        for j in range(n_minor):
            a = i * n_minor + j
            b = ((i + 1) % n_major) * n_minor + j
            c = ((i + 1) % n_major) * n_minor + (j + 1) % n_minor
            d = i * n_minor + (j + 1) % n_minor
            faces.append((a, b, c))
            faces.append((a, c, d))
    v = np.asarray(verts, np.float32) + np.asarray(center, np.float32)
    return v, np.asarray(faces, np.int32)


def sphere_torus_mesh(n_lat=16, n_lon=32, n_major=48, n_minor=16, torus_r=0.12, torus_R=0.75):
    """Sphere r = 0.5 inside a torus of major radius `torus_R`, tube radius `torus_r` -> vertices [V,3] f32, triangles [T,3] i32
    (outward-facing winding)."""
    v0, f0 = uv_sphere(0.5, n_lat, n_lon)
    v1, f1 = torus(torus_R, torus_r, n_major, n_minor)
    return np.concatenate([v0, v1], 0), np.concatenate([f0, f1 + len(v0)], 0)


def sphere_surface_points(n, seed=6, radius=0.5, cam_dist=2.0, n_cams=8):
    """Points on the sphere with analytic normals and view dirs toward one of `n_cams`
    cameras for which the point is front-facing. -> pts, normals, view_dirs  (f32 [n,3])."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    pts = d * radius
    k = np.arange(n_cams)
    cams = np.stack([np.cos(2 * np.pi * k / n_cams) * math.cos(0.5), np.sin(2 * np.pi * k / n_cams) * math.cos(0.5),
                     np.full(n_cams, math.sin(0.5))], -1) * cam_dist
    score = d @ cams.T
    pick = np.argmax(score + rng.uniform(0, 0.3, size=score.shape), axis=-1)
    view = cams[pick] - pts
    view /= np.linalg.norm(view, axis=-1, keepdims=True)
    return pts.astype(np.float32), d.astype(np.float32), view.astype(np.float32)


def torus_surface_points(n, seed=6, R=0.75, r=0.12, cam_dist=2.0, n_cams=8):
    """Points on the torus of `sphere_torus_mesh` (area-uniform: rejection on the (R + r cos w) area element) with analytic normals
    and view dirs toward the camera the normal faces best. -> pts, normals, view_dirs  (f32 [n,3])."""
    rng = np.random.default_rng(seed)
    u = rng.uniform(0, 2 * np.pi, size=3 * n + 64)
    w = rng.uniform(0, 2 * np.pi, size=3 * n + 64)
    keep = rng.uniform(0, R + r, size=u.shape) < R + r * np.cos(w)
    u, w = u[keep][:n], w[keep][:n]
    assert len(u) == n
    nrm = np.stack([np.cos(w) * np.cos(u), np.cos(w) * np.sin(u), np.sin(w)], -1)
    pts = np.stack([R * np.cos(u), R * np.sin(u), np.zeros_like(u)], -1) + r * nrm
    k = np.arange(n_cams)
    cams = np.stack([np.cos(2 * np.pi * k / n_cams) * math.cos(0.5), np.sin(2 * np.pi * k / n_cams) * math.cos(0.5),
                     np.full(n_cams, math.sin(0.5))], -1) * cam_dist
    pick = np.argmax(nrm @ cams.T + rng.uniform(0, 0.3, size=(n, n_cams)), axis=-1)
    view = cams[pick] - pts
    view /= np.linalg.norm(view, axis=-1, keepdims=True)
    return pts.astype(np.float32), nrm.astype(np.float32), view.astype(np.float32)


def scene_surface_points(n, seed=6, torus_r=0.12, torus_R=0.75):
    """Points over the WHOLE scene of sphere_torus_mesh (sphere and torus, split by surface area: 47 % / 53 % at R = 0.75, r = 0.12),
    shuffled."""
    a_s, a_t = 4 * math.pi * 0.25, 4 * math.pi ** 2 * torus_R * torus_r
    n_s = int(round(n * a_s / (a_s + a_t)))
    parts = [sphere_surface_points(n_s, seed=seed), torus_surface_points(n - n_s, seed=seed + 1, R=torus_R, r=torus_r)]
    perm = np.random.default_rng(seed + 2).permutation(n)
    return tuple(np.concatenate([a[i] for a in parts], 0)[perm] for i in range(3))


def pinhole_rays(n, seed=2, h=800, w=800, focal=1111.1, cam_dist=2.0, az=0.7, el=0.5):
    """n random pixels of an h x w pinhole looking at the origin.
    -> rays_o, rays_d (unit), radiis [n,1], rays_cos [n,1]  (shapeRenderer.py:479-486,510 style)."""
    rng = np.random.default_rng(seed)
    idx = rng.permutation(h * w)[:n]
    py, px = (idx // w).astype(np.float64), (idx % w).astype(np.float64)
    c = np.array([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)]) * cam_dist
    fwd = -c / np.linalg.norm(c)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    dx = (px + 0.5 - w / 2) / focal
    dy = (py + 0.5 - h / 2) / focal
    dirs = fwd[None] + dx[:, None] * right[None] - dy[:, None] * up[None]
    nrm = np.linalg.norm(dirs, axis=-1, keepdims=True)
    cos = 1.0 / nrm
    dirs = dirs / nrm
    radii = np.full((n, 1), 1.0 / (focal * math.sqrt(math.pi)))  # sqrt(pixel area / pi)
    o = np.broadcast_to(c, dirs.shape)
    return o.astype(np.float32).copy(), dirs.astype(np.float32), radii.astype(np.float32), cos.astype(np.float32)


# --------------------------------------------------------------------------- synthetic parameters
def _wn_linear(gen, fan_in, fan_out, sd, prefix, bias_fill=None):
    """nn.utils.parametrizations.weight_norm(nn.Linear) state: original0 = g [out,1], original1 = v [out,in]."""
    import torch
    bound = 1.0 / math.sqrt(fan_in)
    v = (torch.rand(fan_out, fan_in, generator=gen) * 2 - 1) * bound
    sd[prefix + ".parametrizations.weight.original1"] = v
    sd[prefix + ".parametrizations.weight.original0"] = v.norm(dim=1, keepdim=True)
    b = (torch.rand(fan_out, generator=gen) * 2 - 1) * bound
    if bias_fill is not None:
        b = torch.full((fan_out,), float(bias_fill))
    sd[prefix + ".bias"] = b


def _linear(gen, fan_in, fan_out, sd, prefix):
    import torch
    bound = 1.0 / math.sqrt(fan_in)
    sd[prefix + ".weight"] = (torch.rand(fan_out, fan_in, generator=gen) * 2 - 1) * bound
    sd[prefix + ".bias"] = (torch.rand(fan_out, generator=gen) * 2 - 1) * bound


def random_flow_state(sd, prefix, gen, R=512, C=12):
    """TensoFlow state_dict entries (network/flow.py:683-698, :755-764 shapes and init scales, plus the
    plane / net perturbation SURVEY.md 8(d) config 3 prescribes so the flow is non-trivial)."""
    import torch
    for i in range(3):
        sd[f"{prefix}nis_plane.{i}"] = 1e-4 * (2 * torch.rand(1, C, R, R, generator=gen) - 1) + 0.1 * torch.randn(1, C, R, R, generator=gen)
        sd[f"{prefix}nis_line.{i}"] = torch.full((1, C, R, 1), 1.0 / (C * 3)) + 0.1 * torch.randn(1, C, R, 1, generator=gen)
    _linear(gen, 3 * C + 21, 64, sd, prefix + "nis_mat.0")
    _linear(gen, 64, 16, sd, prefix + "nis_mat.2")
    for b in range(2):
        dims = [(44, 64), (64, 64), (64, 64), (64, 21)]
        for l, (fi, fo) in zip((1, 3, 5, 7), dims):
            _linear(gen, fi, fo, sd, f"{prefix}flows.{b}.nn.{l}")


def random_mc_state(seed=4, R=512, flow_R=512, env_res=128):
    """Random-init MCShadingNetwork parameters in the reference's state_dict layout
    (network/fields.py:668-760; configs/mat/syn/compressor.yaml:15-21)."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for i in range(3):
        sd[f"mat_plane.{i}"] = 1e-4 * (2 * torch.rand(1, 36, R, R, generator=gen) - 1) + 0.1 * torch.randn(1, 36, R, R, generator=gen)
        sd[f"mat_line.{i}"] = torch.full((1, 36, R, 1), 1.0 / 108) + 0.1 * torch.randn(1, 36, R, 1, generator=gen)
    for name, out in (("metallic", 1), ("roughness", 1), ("albedo", 3)):
        _wn_linear(gen, 108, 128, sd, f"{name}_predictor.0")
        _wn_linear(gen, 128, out, sd, f"{name}_predictor.2")
    dims = [(123, 256), (256, 256), (256, 256), (256, 3)]
    for l, (fi, fo) in zip((0, 2, 4, 6), dims):
        _wn_linear(gen, fi, fo, sd, f"inner_light.{l}", bias_fill=math.log(0.5) if l == 6 else None)
    sd["outer_light.base"] = math.log(0.5) + 0.5 * torch.randn(6, env_res, env_res, 3, generator=gen)
    for name in ("flow_diffuse_copy.", "flow_specular_copy."):
        random_flow_state(sd, name, gen, R=flow_R)
    return sd


def random_sdf_state(seed=1, R=300, C=36, hidden=256, app=128):
    """TensoSDF parameters in the reference layout (network/fields.py:78-131): circle init of radius 0.2
    plus 0.02*N(0,1) on the planes."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    lin = torch.linspace(-1, 1, R)
    xx, yy = torch.meshgrid(lin, lin, indexing="ij")
    circ = (torch.sqrt(xx ** 2 + yy ** 2) - 0.2)[None, None].expand(1, C, R, R)
    for i in range(3):
        sd[f"sdf_plane.{i}"] = (circ + 0.02 * torch.randn(1, C, R, R, generator=gen)).contiguous()
        sd[f"sdf_line.{i}"] = torch.full((1, C, R, 1), 1.0 / (C * 3))
    sd["sdf_mat.0.weight"] = torch.randn(hidden, 3 * C + 3, generator=gen) * (math.sqrt(2) / math.sqrt(hidden))
    sd["sdf_mat.0.bias"] = torch.zeros(hidden)
    sd["sdf_mat.2.weight"] = math.sqrt(math.pi) / math.sqrt(hidden) + 1e-4 * torch.randn(1 + app, hidden, generator=gen)
    sd["sdf_mat.2.bias"] = torch.full((1 + app,), -0.2)
    return sd


def random_shape_shader_state(seed=8, app=128):
    """ShapeShadingNetwork parameters in the reference layout (network/fields.py:360-371; MLPs of other_field.py:50-84)."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for name, dims in (("mat_mlp", [(app, 128), (128, 128), (128, 5)]), ("inner_light", [(123, 128), (128, 128), (128, 3)]),
                       ("inner_weight", [(90, 128), (128, 128), (128, 1)])):
        for l, (fi, fo) in zip((0, 2, 4), dims):
            _wn_linear(gen, fi, fo, sd, f"color_network.{name}.{l}")
    sd["color_network.envlight.base"] = math.log(0.5) + 0.5 * torch.randn(6, 128, 128, 3, generator=gen)
    return sd


def synthetic_fg_lut(n=256):
    """Stand-in for assets/bsdf_256_256.bin (the split-sum DFG table, [1,n,n,2], u = NoV, v = roughness): the analytic
    fit of Karis' "Real Shading in UE4" mobile approximation -- same shape, range and smoothness; synthetic data for benches."""
    import torch
    nov = (torch.arange(n, dtype=torch.float32) + 0.5) / n
    rough = (torch.arange(n, dtype=torch.float32) + 0.5) / n
    r, c = torch.meshgrid(rough, nov, indexing="ij")
    c0 = torch.tensor([-1.0, -0.0275, -0.572, 0.022])
    c1 = torch.tensor([1.0, 0.0425, 1.04, -0.04])
    rr = [r * c0[i] + c1[i] for i in range(4)]
    a004 = torch.minimum(rr[0] * rr[0], torch.exp2(-9.28 * c)) * rr[0] + rr[1]
    A = -1.04 * a004 + rr[2]
    B = 1.04 * a004 + rr[3]
    return torch.stack([A, B], -1)[None].clamp(0, 1).contiguous()
