"""EnvLight (reference: network/light.py:8-162): learnable log-radiance cube map, direct lookup with autograd."""
import numpy as np
import torch

from .. import ops


class _CubeLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, dirs):
        ctx.save_for_backward(base, dirs)
        return ops.cube_lookup(base, dirs, apply_exp=True)

    @staticmethod
    def backward(ctx, g):
        base, dirs = ctx.saved_tensors
        return ops.cube_lookup_bwd(base, dirs, g.contiguous(), apply_exp=True), None


class EnvLight(torch.nn.Module):
    def __init__(self, path=None, device=None, scale=1.0, min_res=16, start_res=16, max_res=512, min_roughness=0.08,
                 max_roughness=0.5, trainable=False):
        super().__init__()
        if path is not None:
            raise NotImplementedError("loading lat-long HDR files is outside the hot path (light.py:39-49)")
        self.device = device if device is not None else "cuda"
        self.scale, self.min_res, self.max_res = scale, min_res, max_res
        self.min_roughness, self.max_roughness, self.trainable, self.start_res = min_roughness, max_roughness, trainable, start_res
        self.base = torch.nn.Parameter(torch.full((6, max_res, max_res, 3), np.log(0.5), dtype=torch.float32, device=self.device),
                                       requires_grad=trainable)
        self.level = max(0, int(np.log2(max_res / start_res)) + 0.5)

    def upsample(self):
        if self.level > 0:
            self.level = max(self.level - 1, 0)

    def build_mips_direct(self, cutoff=0.99):
        """light.py:66-70: the material stage only ever looks up `base` (direct_light ignores the mips)."""
        self.base_mip = [self.base]

    def direct_light(self, l, roughness=None):
        """light.py:125-162 -> exp(bilinear cube lookup), any prefix shape [...,3]."""
        prefix = l.shape[:-1]
        out = _CubeLookup.apply(self.base, l.reshape(-1, 3).contiguous())
        return out.view(*prefix, -1)
