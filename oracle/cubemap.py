"""Oracle restatement of the env-light prefilter (TEST INFRASTRUCTURE -- never imported by the product path).

Follows EnvLight.build_mips (network/light.py:52-64), cubemap_mip (network/light_utils.py:66-80),
ru.diffuse_cubemap / ru.specular_cubemap + __ndfBounds (network/renderutils/ops.py:391-458) and the semantics of the
CUDA kernels they call (network/renderutils/c_src/cubemap.cu: pixel_area :17-31, cube_to_dir :33-46, diffuse :112-140,
GGX filter :170-176,239-287 and their adjoints :142-168,289-349).

The CUDA texel bounding boxes (`specular_bounds`, cubemap.cu:178-237) only prune texels that fail the
`dot(L, V) >= cutoff` test that the filter re-applies itself, so they do not change the result: this restatement sums over
every texel that passes the test.  float32 throughout, same operation order per pair (direction = (fx,fy,+-1)/length,
H = (L+V)/|L+V|, d = (c*a2 - c)*c + 1).

PARITY UNPINNED: the CUDA extension cannot be built or run here (needs nvcc/-lcuda) and the reference's own test of it
(renderutils/tests/test_cubemap.py) is broken, so no golden vectors exist for these five functions; the pin is the source
semantics above plus the self-consistency tests in tests/test_oracle_cubemap.py (adjointness, energy, constant maps).
The GGX weight at roughness 0.08 (a2 = 4.1e-5) is ill-conditioned at the lobe centre: d = 1 - c^2 (1 - a2) loses 3 digits
when c -> 1, so the last-ulp rounding of dot(V, H) for L == V moves that texel's weight by up to ~6e-3 relative.
"""
import functools

import numpy as np
import torch

f32 = np.float32


def texel_dirs(res):
    """cubemap.cu:33-46 cube_to_dir(x, y, side, N) for every texel -> [6, res, res, 3] float32 unit vectors."""
    c = (f32(2.0) * ((np.arange(res, dtype=f32) + f32(0.5)) / f32(res)) - f32(1.0)).astype(f32)
    fy, fx = np.meshgrid(c, c, indexing="ij")
    one = np.ones_like(fx)
    faces = [(one, -fy, -fx), (-one, -fy, fx), (fx, one, fy), (fx, -one, -fy), (fx, -fy, one), (-fx, -fy, -one)]
    v = np.stack([np.stack(f, -1) for f in faces]).astype(f32)
    l = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1] + v[..., 2] * v[..., 2]).astype(f32)).astype(f32)
    return (v / l[..., None]).astype(f32)


def texel_area(res):
    """cubemap.cu:17-31 pixel_area (including its |x-H| asymmetry) -> [res, res] float32 (same for all faces)."""
    if res <= 1:
        return np.ones((res, res), f32)
    H = res // 2
    a = np.abs(np.arange(res) - H).astype(f32)
    d = (np.arctan((a + f32(1)) / f32(H)).astype(f32) - np.arctan(a / f32(H)).astype(f32)).astype(f32)
    return (d[None, :] * d[:, None]).astype(f32)          # [y, x] = dx(x) * dy(y)


def mip(cub):
    """light_utils.py:66-70: 2x2 average pool of a [6,R,R,C] map."""
    cub = np.asarray(cub, f32)
    s = ((cub[:, 0::2, 0::2] + cub[:, 0::2, 1::2]) + cub[:, 1::2, 0::2]) + cub[:, 1::2, 1::2]
    return (s * f32(0.25)).astype(f32)


def mip_bwd(dout):
    """light_utils.py:72-80: the reference's adjoint surrogate -- cube-bilinear fetch of 0.25*dout at the fine texel centres."""
    from .texture import cube_bilinear
    dout = torch.as_tensor(np.asarray(dout, f32))
    res = dout.shape[1] * 2
    out = torch.zeros(6, res, res, dout.shape[-1])
    lin = torch.linspace(-1.0 + 1.0 / res, 1.0 - 1.0 / res, res)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    one = torch.ones_like(gx)
    faces = [(one, -gy, -gx), (-one, -gy, gx), (gx, one, gy), (gx, -one, -gy), (gx, -gy, one), (-gx, -gy, -one)]
    for s, f in enumerate(faces):
        v = torch.stack(f, -1)
        v = v / torch.sqrt(torch.clamp((v * v).sum(-1, keepdim=True), min=1e-20))
        out[s] = cube_bilinear(dout * 0.25, v.reshape(-1, 3)).reshape(res, res, -1)
    return out.numpy()


def _diffuse_weights(res):
    d = texel_dirs(res).reshape(-1, 3)
    cos = np.minimum(np.maximum((d @ d.T).astype(f32), f32(0)), f32(0.999))
    area = np.broadcast_to(texel_area(res), (6, res, res)).reshape(-1)
    return (cos * area[None, :] / f32(3.141592)).astype(f32)          # [out texel, source texel]


def diffuse(cub):
    """cubemap.cu:112-140: irradiance-like cosine filter over ALL texels, weight clamp(N.L,0,.999)*area/3.141592."""
    cub = np.asarray(cub, f32)
    res = cub.shape[1]
    return (_diffuse_weights(res) @ cub.reshape(-1, cub.shape[-1])).reshape(cub.shape).astype(f32)


def diffuse_bwd(dout):
    dout = np.asarray(dout, f32)
    res = dout.shape[1]
    return (_diffuse_weights(res).T @ dout.reshape(-1, dout.shape[-1])).reshape(dout.shape).astype(f32)


@functools.lru_cache(maxsize=None)
def ndf_cutoff(roughness, cutoff=0.99, n=1000000):
    """ops.py:428-441 __ndfBounds: cos(theta) that keeps `cutoff` of the GGX NDF mass (float64 numpy, as the reference)."""
    a2 = roughness ** 4
    cos = np.cos(np.linspace(0, np.pi / 2.0, n))
    c = np.clip(cos, 0.0, 1.0)
    d = (c * a2 - c) * c + 1.0
    D = np.cumsum(a2 / (d * d * np.pi))
    return float(cos[np.argmax(D >= D[-1] * cutoff)])


def _specular_weights(res, roughness, cos_cutoff, rows=None):
    d = texel_dirs(res).reshape(-1, 3)
    V = d if rows is None else d[rows]
    area = np.broadcast_to(texel_area(res), (6, res, res)).reshape(-1)
    alpha = f32(roughness) * f32(roughness)
    a2 = f32(alpha * alpha)
    # per pair, in cubemap.cu's order: dot(L, V); H = (L+V)/|L+V|; dot(V, H)
    LV = ((d[None, :, 0] * V[:, None, 0] + d[None, :, 1] * V[:, None, 1]).astype(f32) + d[None, :, 2] * V[:, None, 2]).astype(f32)
    S = (d[None, :, :] + V[:, None, :]).astype(f32)
    ln = np.sqrt(((S[..., 0] * S[..., 0] + S[..., 1] * S[..., 1]).astype(f32) + S[..., 2] * S[..., 2]).astype(f32)).astype(f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        Hn = np.where(ln[..., None] > 0, S / ln[..., None], f32(0)).astype(f32)
    VH = ((V[:, None, 0] * Hn[..., 0] + V[:, None, 1] * Hn[..., 1]).astype(f32) + V[:, None, 2] * Hn[..., 2]).astype(f32)
    c = np.clip(np.maximum(VH, f32(0)), f32(0), f32(1))
    dd = ((c * a2 - c).astype(f32) * c + f32(1)).astype(f32)
    D = (np.float64(a2) / ((dd * dd).astype(f32).astype(np.float64) * np.pi)).astype(f32)      # M_PI is a double in cubemap.cu:170-176
    w = (np.maximum(LV, f32(0)) * D * area[None, :] / f32(4)).astype(f32)
    return np.where(LV >= f32(cos_cutoff), w, f32(0)).astype(f32)


def specular_rows(cub, roughness, rows, cutoff=0.99):
    """`specular` for a subset of output texels (flat indices) -- lets tests probe a 128^2 map without the full 1e10 pairs."""
    cub = np.asarray(cub, f32)
    flat = cub.reshape(-1, cub.shape[-1])
    out = []
    for r0 in range(0, len(rows), 64):
        w = _specular_weights(cub.shape[1], roughness, ndf_cutoff(roughness, cutoff), np.asarray(rows[r0:r0 + 64]))
        out.append((w @ flat) / w.sum(-1, dtype=f32)[:, None])
    return np.concatenate(out).astype(f32)


def specular(cub, roughness, cutoff=0.99, chunk=None, return_wsum=False):
    """ru.specular_cubemap (ops.py:446-458): GGX-lobe filter normalised by the weight sum (cubemap.cu:239-287)."""
    cub = np.asarray(cub, f32)
    res = cub.shape[1]
    flat = cub.reshape(-1, cub.shape[-1])
    chunk = chunk or max(16, (1 << 22) // flat.shape[0])
    cc = ndf_cutoff(roughness, cutoff)
    out = np.zeros_like(flat)
    ws = np.zeros(flat.shape[0], f32)
    for r0 in range(0, flat.shape[0], chunk):
        rows = np.arange(r0, min(r0 + chunk, flat.shape[0]))
        w = _specular_weights(res, roughness, cc, rows)
        ws[rows] = w.sum(-1, dtype=f32)
        out[rows] = (w @ flat) / ws[rows, None]
    out = out.reshape(cub.shape).astype(f32)
    return (out, ws.reshape(6, res, res)) if return_wsum else out


def specular_bwd(dout, res, roughness, cutoff=0.99, chunk=None):
    """Adjoint of `specular` wrt the cube map (ops.py:421-425 + the python division by the weight sum)."""
    dout = np.asarray(dout, f32).reshape(-1, dout.shape[-1])
    chunk = chunk or max(16, (1 << 22) // dout.shape[0])
    cc = ndf_cutoff(roughness, cutoff)
    g = np.zeros_like(dout)
    for r0 in range(0, dout.shape[0], chunk):
        rows = np.arange(r0, min(r0 + chunk, dout.shape[0]))
        w = _specular_weights(res, roughness, cc, rows)
        g += w.T @ (dout[rows] / w.sum(-1, dtype=f32)[:, None])
    return g.reshape(6, res, res, -1).astype(f32)


def build_mips(base, min_res=16, min_roughness=0.08, max_roughness=0.5, cutoff=0.99):
    """EnvLight.build_mips (light.py:52-64) -> (specular levels list, diffuse)."""
    spec = [np.asarray(base, f32)]
    while spec[-1].shape[1] > min_res:
        spec.append(mip(spec[-1]))
    diff = diffuse(spec[-1])
    for i in range(len(spec) - 1):
        r = (i / (len(spec) - 2)) * (max_roughness - min_roughness) + min_roughness
        spec[i] = specular(spec[i], r, cutoff)
    spec[-1] = specular(spec[-1], 1.0, cutoff)
    return spec, diff
