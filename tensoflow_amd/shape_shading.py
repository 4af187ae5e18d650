"""Split-sum shading of the shape stage on the device: ShapeShadingNetwork.forward (network/fields.py:448-567,
:419-439) with EnvLight.__call__ (network/light.py:72-80,95-122) over a pre-filtered cube-map stack.

Cube-map taps run in tf_cube_lookup_fwd (one launch per mip level touched, lerped on the device); the three 128-wide
per-sample MLPs (mat_mlp 128-128-128-5, inner_light 123-128-128-3, inner_weight 90-128-128-1) are plain library GEMMs
through torch (rocBLAS/hipBLASLt); FG LUT lookup is a bilinear clamp fetch of a [1,256,256,2] table.
EnvLight.build_mips (the renderutils cubemap prefilter, network/light.py:52-64) is NOT part of round 1: the
pre-filtered stack is an input here.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .shading import posenc, wn_weight


def _ide_tables(deg=5):
    ms, ls = [], []
    for i in range(deg):
        l = 2 ** i
        for m in range(l + 1):
            ms.append(m)
            ls.append(l)
    mat = np.zeros((2 ** (deg - 1) + 1, len(ms)))
    for i, (m, l) in enumerate(zip(ms, ls)):
        for k in range(l - m + 1):
            a = 0.5 * (l + k + m - 1.0)
            gb = float(np.prod(a - np.arange(l))) / math.factorial(l)
            leg = (-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m) * gb
            mat[k, i] = math.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * math.pi * math.factorial(l + m))) * leg
    return ms, ls, mat.astype(np.float32)


class Ide5:
    """Integrated directional encoding, degree 5 (utils/ref_utils.py:53-117), real arithmetic."""

    def __init__(self, device):
        ms, ls, mat = _ide_tables(5)
        self.ms = ms
        self.sigma = (0.5 * torch.tensor(ls, dtype=torch.float32) * (torch.tensor(ls, dtype=torch.float32) + 1)).to(device)
        self.mat = torch.from_numpy(mat).to(device)

    def __call__(self, xyz, kappa_inv):
        x, y, z = xyz[:, 0:1], xyz[:, 1:2], xyz[:, 2:3]
        vmz = torch.cat([z ** i for i in range(self.mat.shape[0])], -1)
        poly = vmz @ self.mat
        re, im = [torch.ones_like(x)], [torch.zeros_like(x)]
        for _ in range(16):
            re.append(re[-1] * x - im[-1] * y)
            im.append(re[-2] * y + im[-1] * x)
        cre = torch.cat([re[m] for m in self.ms], -1)
        cim = torch.cat([im[m] for m in self.ms], -1)
        att = torch.exp(-self.sigma * kappa_inv)
        return torch.cat([cre * poly * att, cim * poly * att], -1)


class ShapeShader:
    def __init__(self, sd, env_specular, env_diffuse, fg_lut, device="cuda", prefix="color_network.",
                 min_roughness=0.08, max_roughness=0.5, light_exp_max=0.0):
        sdd = {k: v.to(device).float() for k, v in sd.items() if k.startswith(prefix) and v.is_floating_point()}
        L = lambda name, ids: [(wn_weight(sdd, f"{prefix}{name}.{i}").contiguous(), sdd[f"{prefix}{name}.{i}.bias"]) for i in ids]
        self.mat_mlp = L("mat_mlp", (0, 2, 4))
        self.inner_light = L("inner_light", (0, 2, 4))
        self.inner_weight = L("inner_weight", (0, 2, 4))
        self.spec = [s.to(device).float().contiguous() for s in env_specular]
        self.diff = env_diffuse.to(device).float().contiguous()
        self.fg = fg_lut.to(device).float()                # [1,H,W,2]
        self.ide = Ide5(device)
        self.min_r, self.max_r, self.exp_max = min_roughness, max_roughness, light_exp_max

    @staticmethod
    def _mlp(layers, x, out_act=None):
        for i, (W, b) in enumerate(layers):
            x = F.linear(x, W, b)
            if i + 1 < len(layers):
                x = F.relu(x)
        return out_act(x) if out_act is not None else x

    def env_specular(self, d, rough):
        n = len(self.spec)
        r = rough[:, 0]
        mip = torch.where(r < self.max_r, (r.clamp(self.min_r, self.max_r) - self.min_r) / (self.max_r - self.min_r) * (n - 2),
                          (r.clamp(self.max_r, 1.0) - self.max_r) / (1.0 - self.max_r) + n - 2).clamp(0, n - 1)
        l0 = mip.floor().clamp(max=n - 1)
        f = (mip - l0)[:, None]
        l0 = l0.long()
        l1 = (l0 + 1).clamp(max=n - 1)
        f = torch.where((l1 == l0)[:, None], torch.zeros_like(f), f)
        out = torch.zeros(d.shape[0], 3, device=d.device)
        for li, tex in enumerate(self.spec):
            w = torch.where((l0 == li)[:, None], 1 - f, torch.zeros_like(f)) + torch.where(((l1 == li) & (l0 != li))[:, None], f, torch.zeros_like(f))
            out = out + w * ops.cube_lookup(tex, d, apply_exp=False)
        return torch.exp(out)

    @torch.no_grad()
    def __call__(self, pts, normals, view, feat):
        normals = F.normalize(normals, dim=-1).clone()
        normals[normals[:, :2].sum(-1) == 0.0] = torch.tensor([0.0, 1e-6, 1.0], device=pts.device)
        view = F.normalize(view, dim=-1)
        refl = (view * normals).sum(-1, keepdim=True) * normals * 2 - view
        NoV = (normals * view).sum(-1, keepdim=True)
        mat = self._mlp(self.mat_mlp, feat, torch.sigmoid)
        albedo, rough, metal = mat[:, :3] * 0.77 + 0.03, mat[:, 3:4] * 0.9 + 0.09, mat[:, 4:]
        diffuse = (1 - metal) * albedo * ops.cube_lookup(self.diff, normals, apply_exp=True)
        spec_alb = 0.04 * (1 - metal) + metal * albedo
        direct = self.env_specular(refl.contiguous(), rough)
        pe = posenc(pts, 8)
        expo = lambda t: torch.exp(t.clamp(max=self.exp_max))
        indirect = self._mlp(self.inner_light, torch.cat([pe, self.ide(refl, rough)], -1), expo)
        occ = self._mlp(self.inner_weight, torch.cat([pe, posenc(refl, 6)], -1)) * 0.5 + 0.5
        occ_c = occ.clamp(0, 1)
        light = indirect * occ_c + direct * (1 - occ_c)
        uv = torch.cat([NoV.clamp(0, 1), rough.clamp(0, 1)], -1)
        fg = F.grid_sample(self.fg.permute(0, 3, 1, 2), (uv * 2 - 1)[None, :, None, :], mode="bilinear", padding_mode="border",
                           align_corners=False)[0, :, :, 0].T
        spec = (spec_alb * fg[:, 0:1] + fg[:, 1:2]) * light
        lin = diffuse + spec
        eps = torch.finfo(torch.float32).eps
        color = torch.where(lin <= 0.0031308, 323 / 25 * lin, (211 * lin.clamp(min=eps) ** (5 / 12) - 11) / 200).clamp(0.0, 1.0)
        return color, occ, rough, refl
