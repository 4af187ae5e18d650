"""Split-sum shading of the shape stage on the device: ShapeShadingNetwork.forward (network/fields.py:448-567,
:419-439) with EnvLight.__call__ (network/light.py:72-80,95-122) over a pre-filtered cube-map stack.

Everything per sample -- the three 128-wide MLPs (mat_mlp 128-128-128-5, inner_light 123-128-128-3, inner_weight
90-128-128-1), the encodings, the diffuse / two-mip specular cube lookups, the FG LUT fetch and the sRGB transfer --
runs in ONE launch of tf_shape_shade_fwd (csrc/shape_shade.hip).  The pre-filtered stack comes from EnvLight.build_mips
(tensoflow_amd/network/light.py, tf_cubemap_*).
"""
import torch

from . import ops
from .shading import wn_weight


class ShapeShader:
    """Eval-mode ShapeShadingNetwork over a fixed pre-filtered environment stack; parameters from a reference-layout
    state_dict (same keys as ShapeShadingNetwork).  One HIP launch per call (ops.ShapeShade)."""

    def __init__(self, sd, env_specular, env_diffuse, fg_lut, device="cuda", prefix="color_network.",
                 min_roughness=0.08, max_roughness=0.5, light_exp_max=0.0):
        sdd = {k: v.to(device).float() for k, v in sd.items() if k.startswith(prefix) and v.is_floating_point()}
        L = lambda name, ids: [(wn_weight(sdd, f"{prefix}{name}.{i}").contiguous(), sdd[f"{prefix}{name}.{i}.bias"].contiguous()) for i in ids]
        nets = {name: L(name, (0, 2, 4)) for name in ("mat_mlp", "inner_light", "inner_weight")}
        self.op = ops.ShapeShade(nets, [s.to(device) for s in env_specular], env_diffuse.to(device), fg_lut.to(device),
                                 min_roughness, max_roughness, light_exp_max)

    @torch.no_grad()
    def __call__(self, pts, normals, view, feat):
        """-> color [n,3], occ_prob [n,1], roughness [n,1], reflective [n,3]"""
        return self.op(pts, normals, view, feat)
