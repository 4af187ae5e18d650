"""EnvLight (reference: network/light.py:8-162): learnable log-radiance cube map; prefiltered mip stack (build_mips) and the
diffuse / roughness-indexed specular fetch of the shape stage, direct fetch of the material stage -- all with autograd."""
import numpy as np
import torch

from .. import ops


class _CubeLookup(torch.autograd.Function):
    """exp(bilinear cube fetch): differentiable wrt the map (tf_cube_lookup_bwd) and -- as dr.texture is -- wrt the lookup direction
    (tf_cube_lookup_bwd_dirs; needed while the material stage still trains on the roughness-warped GGX directions)."""

    @staticmethod
    def forward(ctx, base, dirs):
        ctx.save_for_backward(base, dirs)
        return ops.cube_lookup(base, dirs, apply_exp=True)

    @staticmethod
    def backward(ctx, g):
        base, dirs = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            g_base, g_dirs = ops.cube_lookup_bwd_dirs(base, dirs, g.contiguous(), apply_exp=True, want_base=ctx.needs_input_grad[0])
            return g_base, g_dirs.view_as(dirs)
        return ops.cube_lookup_bwd(base, dirs, g.contiguous(), apply_exp=True), None


def ndf_cutoff(roughness, cutoff=0.99, n=1000000):
    """cos(theta) keeping `cutoff` of the GGX NDF mass -- renderutils/ops.py:428-441 (__ndfBounds), float64 on the host, cached."""
    key = (float(roughness), float(cutoff))
    if key not in _NDF_CUTOFF:
        a2 = roughness ** 4
        cos = np.cos(np.linspace(0, np.pi / 2.0, n))
        c = np.clip(cos, 0.0, 1.0)
        d = (c * a2 - c) * c + 1.0
        D = np.cumsum(a2 / (d * d * np.pi))
        _NDF_CUTOFF[key] = float(cos[np.argmax(D >= D[-1] * cutoff)])
    return _NDF_CUTOFF[key]


_NDF_CUTOFF = {}


_CENTRE_DIRS = {}


def _texel_centre_dirs(res, device):
    """Directions of the texel centres as light_utils.py:72-80 builds them (torch.linspace grid + safe_normalize).  Cached per
    (resolution, device): rebuilding them in every _CubeMip.backward was 75 launches per shape-stage training step."""
    key = (int(res), str(device))
    if key not in _CENTRE_DIRS:
        _CENTRE_DIRS[key] = _texel_centre_dirs_build(res, device)
    return _CENTRE_DIRS[key]


def _texel_centre_dirs_build(res, device):
    lin = torch.linspace(-1.0 + 1.0 / res, 1.0 - 1.0 / res, res, device=device)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    one = torch.ones_like(gx)
    faces = [(one, -gy, -gx), (-one, -gy, gx), (gx, one, gy), (gx, -one, -gy), (gx, -gy, one), (-gx, -gy, -one)]
    v = torch.stack([torch.stack(f, -1) for f in faces])
    return (v / torch.sqrt(torch.clamp((v * v).sum(-1, keepdim=True), min=1e-20))).reshape(-1, 3).contiguous()


class _CubeMip(torch.autograd.Function):
    """light_utils.cubemap_mip (:66-80): 2x2 box forward; the backward is the reference's surrogate (cube-bilinear fetch of
    0.25*dout at the fine texel centres), not the exact adjoint."""

    @staticmethod
    def forward(ctx, cube):
        return ops.cubemap_mip(cube)

    @staticmethod
    def backward(ctx, g):
        res = g.shape[1] * 2
        out = ops.cube_lookup((g * 0.25).contiguous(), _texel_centre_dirs(res, g.device), apply_exp=False)
        return out.view(6, res, res, 3)


class _CubeDiffuse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cube):
        return ops.cubemap_diffuse(cube)

    @staticmethod
    def backward(ctx, g):
        return ops.cubemap_diffuse(g.contiguous(), adjoint=True)


class _CubeSpecular(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cube, roughness, cos_cutoff):
        out, wsum = ops.cubemap_specular(cube, roughness, cos_cutoff)
        ctx.save_for_backward(wsum)
        ctx.cfg = (roughness, cos_cutoff)
        return out

    @staticmethod
    def backward(ctx, g):
        (wsum,) = ctx.saved_tensors
        return ops.cubemap_specular_bwd(g.contiguous(), wsum, *ctx.cfg), None, None


class _CubeLookupLinear(torch.autograd.Function):
    """bilinear cube fetch without the exp (one mip level of EnvLight.__call__); gradient wrt the map and -- when the
    directions carry a graph (shape stage: normals / reflective come from the SDF) -- wrt the direction."""

    @staticmethod
    def forward(ctx, tex, dirs):
        ctx.save_for_backward(tex, dirs)
        return ops.cube_lookup(tex, dirs, apply_exp=False)

    @staticmethod
    def backward(ctx, g):
        tex, dirs = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            return ops.cube_lookup_bwd_dirs(tex, dirs, g.contiguous(), apply_exp=False, want_base=ctx.needs_input_grad[0])
        return ops.cube_lookup_bwd(tex, dirs, g.contiguous(), apply_exp=False), None


def latlong_to_cubemap(latlong, res, device="cuda"):
    """light_utils.latlong_to_cubemap (:34-47): a lat-long map [H,W,C] resampled at the texel-centre directions of a cube map
    [6,res,res,C] -- u = atan2(x, -z) / 2pi + 1/2, v = acos(y) / pi, bilinear taps with nvdiffrast's default 'wrap' boundary in both
    axes (texel centres at (i + 1/2) / size).  Set-up code (one call per environment map), plain torch on the device."""
    img = torch.as_tensor(latlong, dtype=torch.float32, device=device)
    H, W, C = img.shape
    v = _texel_centre_dirs(res, device)
    tu = torch.atan2(v[:, 0], -v[:, 2]) / (2 * np.pi) + 0.5
    tv = torch.acos(v[:, 1].clamp(-1, 1)) / np.pi
    x, y = tu * W - 0.5, tv * H - 0.5
    x0, y0 = torch.floor(x), torch.floor(y)
    fx, fy = (x - x0)[:, None], (y - y0)[:, None]
    x0, y0 = x0.long(), y0.long()
    tap = lambda yy, xx: img[yy % H, xx % W]
    out = (tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx) * (1 - fy) + (tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx) * fy
    return out.reshape(6, res, res, C).contiguous()


class _CubeLookupMips(torch.autograd.Function):
    """exp of the trilinear fetch over the specular stack (EnvLight.__call__ with a roughness): one launch each way; gradient wrt every
    level of the stack, the direction and the (clamped) mip coordinate."""

    @staticmethod
    def forward(ctx, dirs, mip, *texs):
        ctx.save_for_backward(dirs, mip, *texs)
        return ops.cube_lookup_mips(texs, dirs, mip, apply_exp=True)

    @staticmethod
    def backward(ctx, g):
        dirs, mip, *texs = ctx.saved_tensors
        g_texs, g_dirs, g_mip = ops.cube_lookup_mips_bwd(texs, dirs, mip, g.contiguous(), apply_exp=True,
                                                         want_texs=any(ctx.needs_input_grad[2:]), want_dirs=ctx.needs_input_grad[0],
                                                         want_mip=ctx.needs_input_grad[1])
        return (g_dirs, g_mip, *(g_texs if g_texs is not None else [None] * len(texs)))


class EnvLight(torch.nn.Module):
    def __init__(self, path=None, device=None, scale=1.0, min_res=16, start_res=16, max_res=512, min_roughness=0.08,
                 max_roughness=0.5, trainable=False):
        super().__init__()
        self.device = device if device is not None else "cuda"
        self.scale, self.min_res, self.max_res = scale, min_res, max_res
        self.min_roughness, self.max_roughness, self.trainable, self.start_res = min_roughness, max_roughness, trainable, start_res
        self.composed_lookup = False          # True: the specular lookup as the per-level torch composition (tests)
        self.base = torch.nn.Parameter(torch.full((6, max_res, max_res, 3), np.log(0.5), dtype=torch.float32, device=self.device),
                                       requires_grad=trainable)
        if path is not None:
            self.load(path)
        self.level = max(0, int(np.log2(max_res / start_res)) + 0.5)

    def load(self, path):
        """light.py:39-49: a lat-long environment picture (.hdr / .exr / .npy / 8-bit) x scale, resampled onto the cube, becomes `base`."""
        from ..hdr_io import imread_float
        image = imread_float(path)[..., :3] * self.scale
        self.base.data = latlong_to_cubemap(image, self.max_res, self.device)

    def upsample(self):
        if self.level > 0:
            self.level = max(self.level - 1, 0)

    def build_mips(self, cutoff=0.99):
        """light.py:52-64: box mips down to min_res, cosine filter of the coarsest level, GGX filter per level
        (roughness min..max over levels 0..M-2, 1.0 for the last).  Differentiable wrt `base`."""
        spec = [self.base]
        while spec[-1].shape[1] > self.min_res:
            spec.append(_CubeMip.apply(spec[-1]))
        self.diffuse = _CubeDiffuse.apply(spec[-1])
        for idx in range(len(spec) - 1):
            r = (idx / (len(spec) - 2)) * (self.max_roughness - self.min_roughness) + self.min_roughness
            spec[idx] = _CubeSpecular.apply(spec[idx], r, ndf_cutoff(r, cutoff))
        spec[-1] = _CubeSpecular.apply(spec[-1], 1.0, ndf_cutoff(1.0, cutoff))
        self.specular = spec

    def get_mip(self, roughness):
        n = len(self.specular)
        return torch.where(roughness < self.max_roughness,
                           (roughness.clamp(self.min_roughness, self.max_roughness) - self.min_roughness)
                           / (self.max_roughness - self.min_roughness) * (n - 2),
                           (roughness.clamp(self.max_roughness, 1.0) - self.max_roughness) / (1.0 - self.max_roughness) + n - 2)

    def forward(self, l, roughness=None, mip=None):
        """light.py:95-122: exp of the cube fetch of the diffuse map, or of the trilinear fetch over the specular stack.
        mip: get_mip(roughness).clamp(0, n - 1) computed by the caller (autograd.ShapeGluePreFn emits it with the roughness)."""
        prefix = l.shape[:-1]
        d = l.reshape(-1, 3).contiguous()
        if roughness is None:
            return torch.exp(_CubeLookupLinear.apply(self.diffuse, d)).view(*prefix, -1)
        n = len(self.specular)
        mip = self.get_mip(roughness.reshape(-1)).clamp(0, n - 1) if mip is None else mip.reshape(-1)
        if d.is_cuda and not self.composed_lookup:
            return _CubeLookupMips.apply(d, mip, *self.specular).view(*prefix, -1)
        # the per-level composition of rounds 1-4 (every level fetched for every sample): the checker of the fused lookup, and the CPU form
        l0 = mip.floor().clamp(max=n - 1)
        f = (mip - l0)[:, None]
        l0 = l0.long()
        l1 = (l0 + 1).clamp(max=n - 1)
        f = torch.where((l1 == l0)[:, None], torch.zeros_like(f), f)
        out = torch.zeros(d.shape[0], 3, device=d.device)
        for li, tex in enumerate(self.specular):
            w = torch.where((l0 == li)[:, None], 1 - f, torch.zeros_like(f)) \
                + torch.where(((l1 == li) & (l0 != li))[:, None], f, torch.zeros_like(f))
            out = out + w * _CubeLookupLinear.apply(tex, d)
        return torch.exp(out).view(*prefix, -1)

    def build_mips_direct(self, cutoff=0.99):
        """light.py:66-70: the material stage only ever looks up `base` (direct_light ignores the mips)."""
        self.base_mip = [self.base]

    def direct_light(self, l, roughness=None):
        """light.py:125-162 -> exp(bilinear cube lookup), any prefix shape [...,3]."""
        prefix = l.shape[:-1]
        out = _CubeLookup.apply(self.base, l.reshape(-1, 3).contiguous())
        return out.view(*prefix, -1)
