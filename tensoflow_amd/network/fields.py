"""TensoSDF and MCShadingNetwork (reference: network/fields.py:20-317, :618-1595), forward direction, on the HIP kernels."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..autograd import LightsFn, ShadeWeightsFn, VmGatherFn
from ..shading import MCShader, fibonacci_samples, sphere_latent
from .flow import TensoFlow, _check_no_grad
from .light import EnvLight


class TVLoss(nn.Module):
    """other_field.py:170-191: squared-difference total variation of a [B,C,H,W] grid (parameter-only regulariser)."""

    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x):
        b, c, h, w = x.shape
        if x.is_cuda and b == 1 and x.dtype == torch.float32 and x.is_contiguous():
            from ..autograd import TvLossFn
            return TvLossFn.apply(x, self.TVLoss_weight)         # tf_tv_fwd / tf_tv_bwd (round 4)
        count_h, count_w = c * (h - 1) * w, c * h * (w - 1)
        total = 0.0
        if count_h != 0:
            total = total + torch.pow(x[:, :, 1:, :] - x[:, :, :h - 1, :], 2).sum() / count_h
        if count_w != 0:
            total = total + torch.pow(x[:, :, :, 1:] - x[:, :, :, :w - 1], 2).sum() / count_w
        return self.TVLoss_weight * 2 * total / b


_CONST3 = {}


def _const3(a, b, c, device):
    """A three-vector constant resident on `device` (built per call it is a host-to-device copy, i.e. a host synchronisation)."""
    key = (a, b, c, str(device))
    if key not in _CONST3:
        _CONST3[key] = torch.tensor([a, b, c], device=device)
    return _CONST3[key]


def _gauss_kernel(kernel_size, sigma, dims):
    x = torch.arange(-kernel_size // 2 + 1.0, kernel_size // 2 + 1.0)
    if dims == 1:
        k = torch.exp(-x ** 2 / (2 * sigma ** 2))
    else:
        xx, yy = torch.meshgrid(x, x, indexing="ij")
        k = torch.exp(-(xx ** 2 + yy ** 2) / (2 * sigma ** 2))
    return k[None, None, ...] / k.sum()


class GaussianBlur2D(nn.Module):
    """other_field.py:146-156 (buffer `kernel` is part of the reference state_dict)."""

    def __init__(self, kernel_size=5, sigma=1.0, stride=2, device="cuda"):
        super().__init__()
        self.kernel_size, self.sigma, self.stride = kernel_size, sigma, stride
        self.register_buffer("kernel", _gauss_kernel(kernel_size, sigma, 2).to(device))

    def forward(self, x):
        return F.conv2d(x, self.kernel, stride=self.stride, padding=self.kernel_size // 2)


class GaussianBlur1D(nn.Module):
    """other_field.py:158-168."""

    def __init__(self, kernel_size=5, sigma=1.0, stride=2, device="cuda"):
        super().__init__()
        self.kernel_size, self.sigma, self.stride = kernel_size, sigma, stride
        self.register_buffer("kernel", _gauss_kernel(kernel_size, sigma, 1).to(device))

    def forward(self, x):
        return F.conv1d(x, self.kernel, stride=self.stride, padding=self.kernel_size // 2)


class TensoSDF(nn.Module):
    def __init__(self, gridSize, aabb, device="cuda", sdf_n_comp=36, sdf_dim=256, app_dim=128, init_n_levels=3, sdf_multires=0):
        super().__init__()
        self.kernel_size, self.sigma = 5, 0.5                       # fields.py:34-36
        self.gaussian1d = GaussianBlur1D(self.kernel_size, self.sigma, stride=1, device=device)
        self.gaussian2d = GaussianBlur2D(self.kernel_size, self.sigma, stride=1, device=device)
        # sdf_multires = m > 0 (fields.py:66-91; the reference class default is 3, the renderers' and every shipped config's 0): the
        # positional encoding of the point (3 + 6 m values; of the CONTRACTED point when m == 3, :294) joins the VM features in front of
        # the decoder.  The fused kernels instantiate m = 0; m > 0 runs as a composition of the gather / encoding / dense-layer kernels
        # (ops.sdf_forward / ops.sdf_alpha / autograd.SdfAlphaFn dispatch on the width of the first layer) -- correct, not fast.
        self.sdf_multires = int(sdf_multires)
        self.sdf_n_comp, self.sdf_dim, self.app_dim, self.device = sdf_n_comp, sdf_dim, app_dim, device
        self.matMode, self.vecMode, self.nplane, self.init_radius = [[0, 1], [0, 2], [1, 2]], [2, 1, 0], 3, 0.2
        self.update_gridSize_aabb(torch.as_tensor(gridSize), aabb, init_n_levels)
        planes, lines = [], []
        for i in range(3):
            ps, ls = self.gridSize[self.matMode[i]], self.gridSize[self.vecMode[i]]
            x, y = torch.meshgrid(torch.linspace(-1, 1, int(ps[0])), torch.linspace(-1, 1, int(ps[1])), indexing="ij")
            init = (torch.sqrt(x ** 2 + y ** 2) - self.init_radius)[None, None].expand(1, sdf_n_comp, -1, -1)
            planes.append(nn.Parameter(init.clone()))
            lines.append(nn.Parameter(torch.ones(1, sdf_n_comp, int(ls), 1) * (1.0 / (sdf_n_comp * 3))))
        self.sdf_plane, self.sdf_line = nn.ParameterList(planes).to(device), nn.ParameterList(lines).to(device)
        in_ch = 3 + 6 * self.sdf_multires
        self.sdf_mat = nn.Sequential(nn.Linear(3 * sdf_n_comp + in_ch, sdf_dim), nn.Softplus(beta=100), nn.Linear(sdf_dim, 1 + app_dim)).to(device)
        nn.init.constant_(self.sdf_mat[0].bias, 0.0)
        if self.sdf_multires > 0:             # (:84-86) only the columns of the raw coordinates start non-zero
            nn.init.constant_(self.sdf_mat[0].weight, 0.0)
            nn.init.normal_(self.sdf_mat[0].weight[:, -in_ch:-(in_ch - 3)], 0.0, np.sqrt(2) / np.sqrt(sdf_dim))
        else:
            nn.init.normal_(self.sdf_mat[0].weight, 0.0, np.sqrt(2) / np.sqrt(sdf_dim))
        nn.init.constant_(self.sdf_mat[-1].bias, -self.init_radius)
        nn.init.normal_(self.sdf_mat[-1].weight, mean=np.sqrt(np.pi) / np.sqrt(sdf_dim), std=0.0001)
        self._packed, self._packed_version = None, None

    def update_gridSize_aabb(self, gridSize, aabb, n_levels):
        self.gridSize, self.aabb = gridSize, aabb
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.units = self.aabbSize / (self.gridSize - 1)
        self.n_levels = n_levels

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        """fields.py:155-178: bilinear (align_corners) resampling of planes / lines to `res_target` rounded down to a multiple of
        2^(levels-1), one more mip level.  The optimizer has to be rebuilt by the caller (new Parameters), as in the reference."""
        new_levels = self.n_levels + 1
        res_target = torch.as_tensor(res_target).cpu()
        res_target = (res_target / 2 ** (new_levels - 1)).int() * 2 ** (new_levels - 1)
        for i in range(3):
            m0, m1 = self.matMode[i]
            self.sdf_plane[i] = nn.Parameter(F.interpolate(self.sdf_plane[i].data, size=(int(res_target[m1]), int(res_target[m0])),
                                                           mode="bilinear", align_corners=True))
            self.sdf_line[i] = nn.Parameter(F.interpolate(self.sdf_line[i].data, size=(int(res_target[self.vecMode[i]]), 1),
                                                          mode="bilinear", align_corners=True))
        self.update_gridSize_aabb(res_target, self.aabb, new_levels)
        self._packed = None
        return res_target, self.n_levels

    def _field(self):
        ver = tuple(p._version for p in list(self.sdf_plane) + list(self.sdf_line)) + (self.n_levels,)
        if self._packed is None or ver != self._packed_version:
            self._packed = ops.VmPacked(list(self.sdf_plane), list(self.sdf_line), self.n_levels)
            self._packed_version = ver
        return self._packed

    def _w(self):
        return [self.sdf_mat[0].weight, self.sdf_mat[0].bias, self.sdf_mat[2].weight, self.sdf_mat[2].bias]

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        return [{"params": self.sdf_line, "lr": lr_init_spatialxyz}, {"params": self.sdf_plane, "lr": lr_init_spatialxyz},
                {"params": self.sdf_mat.parameters(), "lr": lr_init_network}]

    # parameter-only regularisers (device-resident torch; they never touch a ray)
    def TV_loss_sdf(self, reg):
        """fields.py:133-138."""
        grids = [g for i in range(self.nplane) for g in (self.sdf_plane[i], self.sdf_line[i])]
        if isinstance(reg, TVLoss) and all(g.is_cuda and g.dim() == 4 and g.shape[0] == 1 and g.dtype == torch.float32 and g.is_contiguous() for g in grids):
            from ..autograd import TvLossSumFn
            return TvLossSumFn.apply(reg.TVLoss_weight, *grids)      # one accumulation on the device over the six grids
        total = 0
        for i in range(self.nplane):
            total = total + reg(self.sdf_plane[i]) + reg(self.sdf_line[i])
        return total

    def grid_gaussian_loss(self):
        """fields.py:301-309: squared distance of every plane / line to its Gaussian-blurred self (borders excluded)."""
        total, k = 0.0, self.kernel_size // 2
        for i in range(self.nplane):
            pg = self.gaussian2d(self.sdf_plane[i].permute(1, 0, 2, 3)).permute(1, 0, 2, 3)
            lg = self.gaussian1d(self.sdf_line[i].permute(1, 0, 2, 3).squeeze(-1)).unsqueeze(-1).permute(1, 0, 2, 3)
            total = total + torch.sum((self.sdf_plane[i][..., k:-k, k:-k] - pg[..., k:-k, k:-k]).square())
            total = total + torch.sum((self.sdf_line[i][..., k:-k, :] - lg[..., k:-k, :]).square())
        return total

    def forward(self, xyz_sampled, level_vol):
        """fields.py:262-299 -> [N, 1+app_dim]."""
        _check_no_grad(self, "TensoSDF.forward")
        sdf, feat = ops.sdf_forward(self._field(), *self._w(), xyz_sampled.reshape(-1, 3), level_vol, self.aabb)
        return torch.cat([sdf[:, None], feat], -1)

    def sdf(self, xyz_sampled, level_vol=None):
        _check_no_grad(self, "TensoSDF.sdf")
        return ops.sdf_forward(self._field(), *self._w(), xyz_sampled.reshape(-1, 3), level_vol, self.aabb, want_feat=False)[0][:, None]

    def sdf_hidden_appearance(self, xyz_sampled, level_vol):
        return self.forward(xyz_sampled, level_vol)[..., 1:]

    def gradient(self, x, level_vol, training=False, sdf=None):
        """fields.py:227-260 -> (gradients [N,3], normal_hessian [N] or None); one fused launch instead of 6 forwards."""
        _check_no_grad(self, "TensoSDF.gradient")
        if x.shape[0] == 0:
            return (torch.zeros(0, 3, device=x.device), torch.zeros(0, 3, device=x.device) if training else None)
        z = torch.zeros(x.shape[0], device=x.device)
        _, grad, _, _, nh = ops.sdf_alpha(self._field(), *self._w(), x, level_vol, z, torch.zeros_like(x), self.aabb,
                                          [float(u) for u in self.units], 1.0, 0.0, want_feat=False, want_hess=training)
        return grad, (nh if training else None)


class MCShadingNetwork(nn.Module):
    """Eval-mode material-stage shader with the reference's parameter names (fields.py:668-760).
    `ray_tracer` is the (vertices, triangles) pair the reference hands to raytracing.RayTracer (materialRenderer.py:147-149)."""
    # the reference class's defaults (fields.py:617-667), 'outer_light_version': 'direction' included (every shipped yaml sets the key)
    default_cfg = {"diffuse_sample_num": 512, "specular_sample_num": 256, "outer_light_version": "direction", "light_exp_max": 5.0,
                   "inner_light_exp_max": 5.0, "human_lights": False, "gridSize": [512, 512, 512], "nis_diffuse_sample_num": 64,
                   "nis_specular_sample_num": 32, "light_reso": 128, "mat_grid": 512, "reg_min_max": True,
                   "nis_start_iter_diffuse": 1000, "nis_start_iter_specular": 1000, "nis_update_interval_diffuse": 1000,
                   "nis_update_interval_specular": 1000, "nis_loss_iter_diffuse": 500, "nis_loss_iter_specular": 500,
                   "nis_sample_num": 64, "nis_start_iter": 1000, "nis_loss_iter": 500, "nis_update_interval": 1000,
                   "light_upsample_interval": 1000,
                   "geometry_type": "schlick", "random_azimuth": True, "shade_fn": "shade_mixed", "use_nis_all": False,
                   "use_nis_diffuse": True, "use_nis_specular": True, "flow": "pwquad", "flow_diffuse": "pwquad", "flow_specular": "pwquad",
                   "use_half_all": True, "use_half_diffuse": True, "use_half_specular": True, "disable_tensorial": False,
                   "disable_reflected": False}
    # Every switch of the reference's cfg (fields.py:617-667) selects code this build holds -- the default branches fused, the others (built
    # in round 6, each against a reference-run golden) as mode bits of the kernels or as device-resident compositions:
    #   use_half_diffuse / use_half_specular = False (:1117-1134, :1190-1203)   tf_shade_dirs_whole                     shading_whole
    #   disable_tensorial / disable_reflected (flow.py:807-812)                 TensoFlow._condition, MCShader.shade    shading_ablate
    #   geometry_type = 'ggx_smith' (:1000-1008)                                mode bit 4 of the direction kernels     shading_smith
    #   flow_diffuse / flow_specular = 'pwlinear' (flow.py:174-312)             TensoFlow's composed transforms         shading_pwlinear
    #   use_nis_diffuse / use_nis_specular = False, one copy active (:1081)     forward_train_fixed(flow_lobes=)        shading_nonis_*, shading_mixed
    #   shade_fn = 'shade_mixed_all' (+ use_nis_all, :1337-1451)                forward_all                             shading_all*
    #   flow* = 'realnvp' (flow.py:645)                                         TensoFlow's composed transforms         tensoflow_realnvp, shading_realnvp
    #   human_lights with outer_light_version = 'envlight' (:929-930, :962-968) the composed passes                     shading_envhuman
    # An unknown flow / shade_fn / geometry_type / outer_light_version raises at construction (as the reference does when it gets there).

    def __init__(self, cfg, ray_tracer, aabb, unit_size):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        if self.cfg["outer_light_version"] not in ("envlight", "direction", "sphere_direction"):
            raise NotImplementedError(f"outer_light_version {self.cfg['outer_light_version']!r}")
        if self.cfg["shade_fn"] not in ("shade_mixed", "shade_mixed_all"):     # fields.py:1458-1463
            raise NotImplementedError(f"shade_fn {self.cfg['shade_fn']!r}: 'shade_mixed' or 'shade_mixed_all'")
        for key in ("flow", "flow_diffuse", "flow_specular"):              # TensoFlow.flow_kwargs (flow.py:644-648)
            if self.cfg[key] not in ("pwquad", "pwlinear", "realnvp"):
                raise NotImplementedError(f"MCShadingNetwork cfg {key}={self.cfg[key]!r}: 'pwquad', 'pwlinear' or 'realnvp'")
        if self.cfg["geometry_type"] not in ("schlick", "ggx_smith"):      # fields.py:1026-1033: anything else raises there too
            raise NotImplementedError(f"geometry_type {self.cfg['geometry_type']!r}: 'schlick' or 'ggx_smith'")
        self.aabb, self.unit_size, self.ray_tracer = aabb, float(unit_size), ray_tracer
        R, C = self.cfg["mat_grid"], 36
        self.mat_plane = nn.ParameterList([nn.Parameter(1e-4 * (2 * torch.rand(1, C, R, R) - 1)) for _ in range(3)]).cuda()
        self.mat_line = nn.ParameterList([nn.Parameter(torch.ones(1, C, R, 1) / (C * 3)) for _ in range(3)]).cuda()
        wn = nn.utils.parametrizations.weight_norm
        mk2 = lambda o: nn.Sequential(wn(nn.Linear(108, 128)), nn.ReLU(), wn(nn.Linear(128, o)), nn.Sigmoid()).cuda()
        self.metallic_predictor, self.roughness_predictor, self.albedo_predictor = mk2(1), mk2(1), mk2(3)
        self.inner_light = nn.Sequential(wn(nn.Linear(123, 256)), nn.ReLU(), wn(nn.Linear(256, 256)), nn.ReLU(), wn(nn.Linear(256, 256)),
                                         nn.ReLU(), wn(nn.Linear(256, 3)), nn.Identity()).cuda()
        nn.init.constant_(self.inner_light[-2].bias, np.log(0.5))
        if self.cfg["outer_light_version"] == "envlight":
            self.outer_light = EnvLight(trainable=True, max_res=self.cfg["light_reso"])
        else:
            # fields.py:716-721: make_predictor_4layer(72 | 72 * 2, 3, activation='exp', exp_max=light_exp_max) on the IDE of the ray
            # direction ('sphere_direction': and of the point where the ray leaves the unit sphere)
            n_in = 72 if self.cfg["outer_light_version"] == "direction" else 144
            self.outer_light = nn.Sequential(wn(nn.Linear(n_in, 256)), nn.ReLU(), wn(nn.Linear(256, 256)), nn.ReLU(), wn(nn.Linear(256, 256)),
                                             nn.ReLU(), wn(nn.Linear(256, 3)), nn.Identity()).cuda()
            nn.init.constant_(self.outer_light[-2].bias, np.log(0.5))
        if self.cfg["human_lights"]:
            # fields.py:727-729: make_predictor_4layer(2 * 2 * 6, 4, activation='exp') -- exp(min(x, 0)) -- on the capturer's plane
            self.human_light = nn.Sequential(wn(nn.Linear(24, 256)), nn.ReLU(), wn(nn.Linear(256, 256)), nn.ReLU(), wn(nn.Linear(256, 256)),
                                             nn.ReLU(), wn(nn.Linear(256, 4)), nn.Identity()).cuda()
            nn.init.constant_(self.human_light[-2].bias, np.log(0.02))
        self._composed_lights = self.cfg["outer_light_version"] == "sphere_direction" or self.cfg["human_lights"]
        # fields.py:755-760: one transform per lobe (cfg flow_diffuse / flow_specular; TensoFlow refuses 'realnvp')
        mkflow = lambda kind: TensoFlow(d=2, aabb=aabb, gridSize=self.cfg["gridSize"], device="cuda", flow=kind,
                                        disable_tensorial=bool(self.cfg["disable_tensorial"]), disable_reflected=bool(self.cfg["disable_reflected"]))
        self.use_flow_copy = False
        if self.cfg["use_nis_all"]:                 # fields.py:751-753: shade_mixed_all's single flow over both lobes
            self.flow, self.flow_copy = mkflow(self.cfg["flow"]), mkflow(self.cfg["flow"])
        if self.cfg["use_nis_diffuse"]:             # (a lobe without its flow holds no flow modules: the reference's state_dict keys)
            self.flow_diffuse, self.flow_diffuse_copy = mkflow(self.cfg["flow_diffuse"]), mkflow(self.cfg["flow_diffuse"])
        if self.cfg["use_nis_specular"]:
            self.flow_specular, self.flow_specular_copy = mkflow(self.cfg["flow_specular"]), mkflow(self.cfg["flow_specular"])
        # the fused inference pass (MCShader) instantiates the default transform; any other runs the compositions (see _forward_eval)
        self._fused_flows = all(getattr(self, n)._fused for n in ("flow_diffuse_copy", "flow_specular_copy") if hasattr(self, n))
        self._shader, self._shader_version = None, None
        self.use_flow_diffuse_copy = self.use_flow_specular_copy = False      # fields.py:752-760: set by update_step at nis_start_iter

    def _param_version(self):
        """Every in-place update of a parameter (optimizer.step, load_state_dict, copy_) bumps its _version counter."""
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def shader(self):
        """The device-side evaluator of the CURRENT parameters (packed pyramids, weight-norm folded, fragment-ordered decoders; the
        BVH is uploaded once and kept).  Cached on the parameters' version counters, like TensoSDF._field: an optimizer step,
        load_state_dict or a flow-copy refresh between two evaluations rebuilds it -- a validation render never sees stale weights."""
        ver = self._param_version()
        if self._shader is not None and ver == self._shader_version:
            return self._shader
        v, f = self.ray_tracer
        sd = {k: t.detach() for k, t in self.state_dict().items()}
        old = self._shader
        self._shader = MCShader(sd, v, f, self.aabb, self.unit_size, device="cuda", n_fixed_diffuse=self.cfg["diffuse_sample_num"],
                                exp_max=self.cfg["inner_light_exp_max"], n_fixed_specular=self.cfg["specular_sample_num"],
                                bvh=old.bvh if old is not None else None, light_exp_max=self.cfg["light_exp_max"],
                                # not a reference key: "f16x2" opts in to the narrower inner-light operands (MCShader.__init__); default f16x3
                                inner_precision={"f16x3": ops.PREC_F16X3, "f16x2": ops.PREC_F16X2}[self.cfg.get("inner_light_operands", "f16x3")],
                                use_half=(bool(self.cfg["use_half_diffuse"]), bool(self.cfg["use_half_specular"])),
                                flow_ablate=(bool(self.cfg["disable_tensorial"]), bool(self.cfg["disable_reflected"])),
                                geometry_type=self.cfg["geometry_type"])
        if self._composed_lights:
            self._shader.overlap_dirs = False      # (the composed miss branch allocates between the streams' kernels: keep one stream)
        self._shader_version = ver
        return self._shader

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001, lr_init_envlight=0.001):
        """fields.py:1580-1595."""
        from ..trainer import material_param_groups
        return material_param_groups(self, lr_init_spatialxyz, lr_init_network, lr_init_envlight)

    def flow_copies(self):
        """The frozen sampling copies this cfg holds (fields.py:755-760)."""
        return [getattr(self, n) for n in ("flow_copy", "flow_diffuse_copy", "flow_specular_copy") if hasattr(self, n)]

    def update_step(self, step):
        """fields.py:1050-1065: every nis_update_interval steps from nis_start_iter on, the frozen sampling copies take the weights of
        the trained flows."""
        done = []
        for name, sfx in (("all", ""), ("diffuse", "_diffuse"), ("specular", "_specular")):
            start, every = self.cfg[f"nis_start_iter{sfx}"], self.cfg[f"nis_update_interval{sfx}"]
            if self.cfg[f"use_nis_{name}"] and (step + 1) >= start and (step + 1 - start) % every == 0:
                src, dst = getattr(self, f"flow{sfx}"), getattr(self, f"flow{sfx}_copy")
                dst.load_state_dict(src.state_dict())
                for p in dst.parameters():
                    p.requires_grad = False
                setattr(self, f"use_flow{sfx}_copy", True)        # load_state_dict bumped the copies' parameter versions: shader() re-packs
                done.append(name)
        # fields.py:1067-1068: the cube map's `level` bookkeeping (EnvLight.upsample; no lookup reads it -- kept for state fidelity.  The
        # reference calls it on the net of 'direction' / 'sphere_direction' too, where it does not exist)
        every = self.cfg.get("light_upsample_interval", 1000)
        if self.cfg["outer_light_version"] == "envlight" and every and (step + 1) % every == 0:
            self.outer_light.upsample()
        return done

    def TV_loss(self):
        """fields.py:1525-1530."""
        reg = TVLoss()
        total = 0
        for i in range(3):
            total = total + reg(self.mat_plane[i]) + reg(self.mat_line[i])
        return total

    def material_regularization(self, pts, normals, metallic, roughness, albedo, step):
        """fields.py:1547-1578 (the active terms): 0.1 TV of the material grids + the range hinge of the first 2000 steps."""
        reg = self.TV_loss() * 0.1
        if self.cfg["reg_min_max"] and step is not None and step < 2000:
            reg = reg + torch.sum(torch.clamp(roughness - 0.9 ** 2, min=0)) + torch.sum(torch.clamp(0.1 ** 2 - roughness, min=0))
            reg = reg + torch.sum(torch.clamp(metallic - 0.98, min=0)) + torch.sum(torch.clamp(0.02 - metallic, min=0))
        return reg.reshape(1)

    def _outer_mlp(self, dirs, origins=None):
        """predict_outer_lights('direction' | 'sphere_direction') (fields.py:913-928) as a differentiable composition: IDE of the rows
        as they are (and of the point where the ray leaves the unit sphere), the net's products on tf_linear_fwd / tf_linear_bwd."""
        from ..encodings import ide5
        zero = torch.zeros(dirs.shape[0], 1, device=dirs.device)
        enc = ide5(dirs, zero, wide=True)
        if self.cfg["outer_light_version"] == "sphere_direction":
            o = origins if origins is not None else dirs                    # predict_outer_lights_pts feeds the direction twice (:1516)
            if origins is not None:
                far = (o.norm(dim=-1) > 0.999)[:, None]
                o = torch.where(far, o * 0.999, o)
                dtx = (o * dirs).sum(-1, keepdim=True)
                o = o + dirs * (-dtx + torch.sqrt(dtx ** 2 - (o ** 2).sum(-1, keepdim=True) + 1 + 1e-6))
            enc = torch.cat([enc, ide5(o, zero, wide=True)], -1)
        return torch.exp(torch.clamp(_mlp(self.outer_light, enc), max=self.cfg["light_exp_max"]))

    def _miss_lights(self, origins, dirs, poses):
        """Outer light of the rays that missed, blended with the capturer's reflection when human_lights is on (fields.py:962-968)
        -> (lights [n,3], human_lights * human_weights [n,3] or None)."""
        # (the cube map with human lights, fields.py:929-930 + :962-968 -- no shipped config combines them: composed passes only)
        outer = self.outer_light.direct_light(dirs) if self.cfg["outer_light_version"] == "envlight" else self._outer_mlp(dirs, origins)
        if not self.cfg["human_lights"]:
            return outer, None
        R, t = poses[:, :, :3], poses[:, :, 3]
        p_ = torch.einsum("nij,nj->ni", R, origins) + t
        d_ = torch.einsum("nij,nj->ni", R, dirs)
        hits = d_[:, 2].abs() > 1e-4
        dz = torch.where(hits, d_[:, 2], torch.full_like(d_[:, 2], 1e-4))
        dist = -p_[:, 2] / dz
        mean = (p_[:, :2] + dist[:, None] * d_[:, :2]) * 0.3
        hits = (hits & (mean.norm(dim=-1) < 1.5) & (dist > 0)).float()[:, None]
        mean = mean * hits
        scaled = (mean[:, None, :] * (2.0 ** torch.arange(6, device=mean.device))[:, None]).reshape(-1, 12)         # IPE(mean, 0, 0, 6)
        pe = torch.sin(torch.cat([scaled, scaled + 0.5 * math.pi], -1))
        h = torch.exp(torch.clamp(_mlp(self.human_light, pe), max=0.0)) * hits
        hl, hw = h[:, :3], h[:, 3:].clamp(0.0, 1.0)
        return outer * (1 - hw) + hl * hw, hl * hw

    def predict_outer_lights_pts(self, pts):
        """fields.py:1512-1520."""
        if self.cfg["outer_light_version"] != "envlight":          # 'sphere_direction' feeds the direction's IDE twice (:1515-1516)
            return self._outer_mlp(pts)
        return self.outer_light.direct_light(pts)

    def env_light(self, h, w, gamma=True, no_grad=True):
        """fields.py:1475-1510: lat-long image [h,w,3] of the learned environment light."""
        from ..encodings import linear_to_srgb
        dev = self.mat_line[0].device
        azs = torch.linspace(1.0, 0.0, w, device=dev) * np.pi * 2 - np.pi / 2
        els = torch.linspace(1.0, -1.0, h, device=dev) * np.pi / 2
        els, azs = torch.meshgrid(els, azs, indexing="ij")
        xyz = torch.stack([torch.cos(els) * torch.cos(azs), torch.cos(els) * torch.sin(azs), torch.sin(els)], -1).reshape(h * w, 3)
        with torch.set_grad_enabled(not no_grad):
            lights = self.predict_outer_lights_pts(xyz.contiguous())
        return (linear_to_srgb(lights) if gamma else lights).reshape(h, w, 3)

    def predict_materials(self, pts):
        """fields.py:1010-1017 -> metallic [pn,1], roughness [pn,1] (squared-roughness convention, remapped), albedo [pn,3]."""
        feat = VmGatherFn.apply(pts.contiguous(), None, self.aabb, 3, *self.mat_plane, *self.mat_line)
        return (_mlp(self.metallic_predictor, feat), _mlp(self.roughness_predictor, feat) * (1.0 - 0.04 ** 2) + 0.04 ** 2,
                _mlp(self.albedo_predictor, feat))

    def _linear_to_srgb(self, lin, clamp01=False):
        from ..autograd import linear_to_srgb       # one launch each way on the device (was nine element-wise launches per call)
        return linear_to_srgb(lin, clamp01)

    def _lights_of(self, origins, dirs, poses=None):
        """get_lights (fields.py:951-975) as a differentiable composition: visibility by tf_bvh_trace (no gradient, like the reference's
        ray tracer), the outer light of the rays that miss (cube map or net; human lights blended in), the inner-light net on the hits;
        differentiable wrt the map / nets and wrt `dirs`.  -> (lights [n,3], hit [n], human_lights * human_weights of the misses or None)."""
        from ..encodings import ide5, posenc
        dev = dirs.device
        with torch.no_grad():
            inters, nrm, depth, hit = self._bvh.trace(origins.contiguous(), dirs.detach().contiguous(), 1e-5, 2 * self.unit_size)
        lights = torch.zeros_like(dirs)
        miss = ~hit
        hl = None
        if bool(miss.any()):
            if self.cfg["outer_light_version"] == "envlight" and not self.cfg["human_lights"]:
                outer = self.outer_light.direct_light(dirs[miss])
            else:
                outer, hl = self._miss_lights(origins[miss], dirs[miss], poses[miss] if poses is not None else None)
            lights = lights.index_put((miss,), outer)
        if bool(hit.any()):
            vd = F.normalize(-dirs[hit], dim=-1)
            nh = F.normalize(nrm[hit], dim=-1)
            refl = (vd * nh).sum(-1, keepdim=True) * nh * 2 - vd
            enc = torch.cat([posenc(inters[hit], 8), ide5(refl, torch.zeros(refl.shape[0], 1, device=dev))], -1)
            lights = lights.index_put((hit,), torch.exp(torch.clamp(_mlp(self.inner_light, enc), max=self.cfg["inner_light_exp_max"])))
        return lights * (depth > 1e-5).float()[:, None], hit, hl

    def forward_train(self, pts, view_dirs, normals, step=None, is_train=True, human_poses=None):
        """Differentiable forward of shade_mixed with the flow samplers (fields.py:1075-1335): every per-sample stage runs in
        the HIP kernels through autograd Functions whose backward is HIP as well (VM gather, BRDF weights, cube map, flow
        log-density, the inner-light / outer-light nets: LightsFn) or the dense-layer kernels tf_linear_fwd / tf_linear_bwd (the
        per-point material MLPs through autograd.mlp_apply) -- no library GEMM in a training step."""
        dev = pts.device
        pn = pts.shape[0]
        sd, ss = self.cfg["nis_diffuse_sample_num"], self.cfg["nis_specular_sample_num"]
        feat = VmGatherFn.apply(pts.contiguous(), None, self.aabb, 3, *self.mat_plane, *self.mat_line)
        metallic = _mlp(self.metallic_predictor, feat)
        roughness = _mlp(self.roughness_predictor, feat) * (1.0 - 0.04 ** 2) + 0.04 ** 2
        albedo = _mlp(self.albedo_predictor, feat)
        if not hasattr(self, "_bvh"):
            self._bvh = ops.Bvh(self.ray_tracer[0], self.ray_tracer[1], dev)
            self._fixed = fibonacci_samples(self.cfg["diffuse_sample_num"]).to(dev)
        with torch.no_grad():
            va = ops.view_angles(normals, view_dirs)
            jit = (lambda n: torch.rand(pn, n, device=dev)) if (is_train and self.training) else (lambda n: None)
            ang_d, lq_d = self.flow_diffuse_copy._sample_nograd(pts, va, sd, jit(sd))
            ang_s, lq_s = self.flow_specular_copy._sample_nograd(pts, va, ss, jit(ss))
            az_jit = torch.rand(pn, device=dev) if (is_train and self.training and self.cfg["random_azimuth"]) else None      # fields.py:837
        wgt, dirs, smask, live, logjac = ShadeWeightsFn.apply(metallic, roughness, albedo, normals.contiguous(), view_dirs.contiguous(),
                                                              ang_d, lq_d, self._fixed, ang_s, lq_s, az_jit,
                                                              (not self.cfg["use_half_diffuse"], not self.cfg["use_half_specular"]),
                                                              self.cfg["geometry_type"] == "ggx_smith")
        T = dirs.shape[1]
        nd = sd + self._fixed.shape[0]
        pts_rep = pts.contiguous()                       # T rays per origin row (tf_bvh_trace rays_per_origin)
        inner_wb = []
        for i in (0, 2, 4, 6):
            inner_wb += [self.inner_light[i].weight, self.inner_light[i].bias]     # weight = g*v/|v| (parametrization, autograd)
        hl_all = None
        # zero-weight rays (diffuse samples under the horizon) are TRACED like any other: their light enters diffuse_light, which the
        # diffuse-light regulariser differentiates (materialRenderer.py:562-563), and the unweighted maps.  cfg cull_zero_weight_rays=True
        # skips them (their colour contribution is exactly zero) at the price of those maps.
        if not self.cfg.get("cull_zero_weight_rays", False):
            live = torch.ones_like(live)
        if self._composed_lights:
            # 'sphere_direction' / human lights (configs/mat/custom): the composed get_lights of the fixed-sampler pass
            poses_rep = human_poses[:, None].expand(pn, T, 3, 4).reshape(-1, 3, 4) if human_poses is not None else None
            lights, hit, hl_miss = self._lights_of(pts_rep[:, None].expand(pn, T, 3).reshape(-1, 3), dirs.reshape(-1, 3), poses_rep)
            if hl_miss is not None:                      # human_lights * human_weights of the rays that missed, back on the ray grid
                hl_all = torch.zeros(pn * T, 3, device=dev).index_put((~hit,), hl_miss.detach())
        else:
            if self.cfg["outer_light_version"] == "direction":
                env_base = None
                for i in (0, 2, 4, 6):
                    inner_wb += [self.outer_light[i].weight, self.outer_light[i].bias]
            else:
                env_base = self.outer_light.base
            lights, hit = LightsFn.apply(env_base, pts_rep, dirs.reshape(-1, 3), live.reshape(-1), self._bvh, self.unit_size,
                                         self.cfg["inner_light_exp_max"], self.cfg.get("precision", ops.PREC_F16X3), self.cfg["light_exp_max"],
                                         *inner_wb)
        lights = lights.view(pn, T, 3)
        contrib = wgt * lights
        diffuse_lin, specular_lin = contrib[:, :nd].sum(1), contrib[:, nd:].sum(1)
        colors = self._linear_to_srgb(diffuse_lin + specular_lin)
        from ..shading import LazyOutputs, aux_from_stats
        outputs = LazyOutputs({"albedo": albedo, "roughness": roughness, "metallic": metallic, "normal": (F.normalize(normals, dim=-1) + 1) / 2,
                               "specular_mask": smask})
        def aux_maps():
            # the rest of the reference's dict (fields.py:1232-1256, :1288-1291; `variance` is what trainer_inv.py:299 prints): one
            # statistics launch on the detached arrays.  (The reference's maps carry gradients nobody uses; only diffuse_light feeds a
            # loss.)  Built on first access of any of the keys (round 5): no loss term of the material stage reads them
            # (network/loss.py; trainer.MaterialTrainer), and building them was ~70 of a training step's 223 forward launches.
            with torch.no_grad():
                hit_u8 = hit.reshape(-1).view(torch.uint8) if hit.dtype == torch.bool else hit.reshape(-1).to(torch.uint8)
                _, _, _, stats = ops.shade_reduce_aux(wgt.detach(), smask, nd, ss, lights=lights.detach(), hit_u8=hit_u8, want_colors=False)
                d = aux_from_stats(stats, nd, ss, metallic.detach(), specular_lin.detach(), diffuse_lin.detach(), self.cfg["diffuse_sample_num"])
            d.pop("diffuse_light")                       # the differentiable one below is the dict's
            return d
        outputs.set_lazy_group(("specular_light", "diffuse_color", "specular_color", "approximate_light", "visibility", "indirect_light",
                                "variance", "variance_diffuse_vis", "variance_specular_vis"), aux_maps)
        outputs["diffuse_light"] = self._linear_to_srgb(lights[:, :nd].mean(1), clamp01=True)        # differentiable (diffuse-light regulariser)
        spec_sel = lambda: smask.bool()
        outputs.set_lazy("human_lights", lambda: (lambda sel: (hl_all.view(pn, T, 3)[:, nd:][sel] if hl_all is not None else
                                                               torch.zeros(int(sel.sum()), 3, device=dev)))(spec_sel() & ~hit.view(pn, T)[:, nd:].bool()))
        outputs.set_lazy("inter", lambda: self._bvh.trace(pts_rep[:, None].expand(pn, ss, 3)[spec_sel()].contiguous(),
                                                           dirs.detach()[:, nd:][spec_sel()].contiguous(), 1e-5, 2 * self.unit_size)[0])
        zero = torch.zeros((), device=dev)
        outputs["loss_nis_diffuse"] = outputs["loss_nis_specular"] = zero
        if step is not None and step >= self.cfg.get("nis_loss_iter_diffuse", 500):
            # loss = -mean(fx * logq / p) with fx / p = wgt * count * lights on the flow slots (fields.py:1271-1284)
            x = ang_d.clamp(1e-6, 1 - 1e-6)
            _, logq = self.flow_diffuse(pts, va, roughness.detach(), x, return_jacobian=True)
            logqx = logq[..., 0] - logjac[:, :sd]
            outputs["loss_nis_diffuse"] = -(contrib[:, :sd] * float(nd) * logqx[..., None]).mean()
        if step is not None and step >= self.cfg.get("nis_loss_iter_specular", 500):
            # ONE compaction of the specular mask (a boolean index is a nonzero -- eight launches and a host sync -- each time)
            sel = torch.nonzero(smask.reshape(-1))[:, 0]
            rid = sel // ss
            x = ang_s.reshape(-1, ang_s.shape[-1]).index_select(0, sel).clamp(1e-6, 1 - 1e-6).contiguous()
            _, logq = self.flow_specular(pts, va, roughness.detach(), x, return_jacobian=True, rays_id=rid)
            logqx = logq[:, 0] - logjac[:, sd:].reshape(-1).index_select(0, sel)
            outputs["loss_nis_specular"] = -(contrib[:, nd:].reshape(-1, 3).index_select(0, sel) * float(ss) * logqx[:, None]).mean()
        outputs["loss_nis"] = outputs["loss_nis_diffuse"] + outputs["loss_nis_specular"]
        return colors, outputs

    # ---------------------------------------------------------------- training before the flow copies take over the sampling
    def forward_train_fixed(self, pts, view_dirs, normals, step=None, is_train=True, human_poses=None, flow_lobes=(False, False)):
        """flow_lobes = (diffuse, specular): that lobe draws from its frozen flow copy instead (the mixed states of shade_mixed: one
        copy active and the other not yet, or cfg use_nis_diffuse / use_nis_specular = False; both True is forward_train's fused form).
        A flow-sampled lobe's directions carry no gradient (fields.py:1084-1134, :1163-1203); its NIS loss is fitted on the sampled
        angles (:1257-1284, :1294-1330).  Default:
        shade_mixed with BOTH fixed samplers (fields.py:1075-1335 with the `else` branches: the material stage's first
        nis_start_iter = 1000 steps, update_step :1050-1065): 512 cosine directions for the diffuse lobe and the roughness-warped
        GGX set for the specular lobe (sample_diffuse_directions / sample_specular_directions, :824-903).  Unlike the flow pass, the
        specular DIRECTIONS depend on the predicted roughness, so the colour gradient reaches the material grids through the
        directions as well: through the BRDF terms, through the environment lookup (dr.texture is differentiable in its
        coordinates: tf_cube_lookup_bwd_dirs) and through the inner-light net's reflected-direction encoding.  The per-direction
        algebra is composed from differentiable device ops; visibility (tf_bvh_trace), the cube map, the VM gather and every MLP
        product (tf_linear_fwd / tf_linear_bwd) are the HIP kernels.  From nis_loss_iter on, the NIS losses fit the trainable
        flows on these fixed samples (:1253-1330)."""
        from ..encodings import ide5, posenc
        EPS, PI = 1e-6, math.pi
        dev = pts.device
        pn = pts.shape[0]
        cfg = self.cfg
        view_dirs, normals = F.normalize(view_dirs, dim=-1), F.normalize(normals, dim=-1)
        metallic, roughness, albedo = self.predict_materials(pts)
        sat = lambda a, b: torch.clamp((a * b).sum(-1, keepdim=True), 0.0, 1.0)
        if not hasattr(self, "_bvh"):
            self._bvh = ops.Bvh(self.ray_tracer[0], self.ray_tracer[1], dev)
            self._fixed = fibonacci_samples(cfg["diffuse_sample_num"]).to(dev)
        if not hasattr(self, "_fixed_s"):
            self._fixed_s = fibonacci_samples(cfg["specular_sample_num"]).to(dev)
        # tangent frame (get_orthogonal_directions, :812-822)
        z = normals
        x0 = torch.stack([z[:, 1], -z[:, 0], torch.zeros_like(z[:, 0])], -1)
        x1 = torch.stack([-z[:, 2], torch.zeros_like(z[:, 0]), z[:, 0]], -1)
        x = F.normalize(torch.where((x0.norm(dim=-1) > x1.norm(dim=-1))[:, None], x0, x1), dim=-1)
        y = torch.cross(z, x, dim=-1)
        X, Y, Z, V = x[:, None], y[:, None], z[:, None], view_dirs[:, None]
        jitter = is_train and self.training and cfg.get("random_azimuth", True)

        def half_angles(H):
            cz = (Z * H).sum(-1, keepdim=True).clamp(-1 + EPS, 1 - EPS)
            phi = (torch.atan2((Y * H).sum(-1, keepdim=True), (X * H).sum(-1, keepdim=True)) + 2 * PI) % (2 * PI)
            return phi, torch.acos(cz)

        def lights_of(origins, dirs, poses=None):
            l, h, _ = self._lights_of(origins, dirs, poses)
            return l, h

        # ---- diffuse lobe: fixed cosine set (no parameter dependence in the directions)
        az, el = self._fixed[:, 0][None, :, None] * (2 * PI), self._fixed[:, 1][None, :, None]
        if jitter:
            az = (az + torch.rand(pn, 1, 1, device=dev) * (2 * PI)) % (2 * PI)
        el_sqrt = torch.sqrt(el + 1e-7)
        d_dirs = (el_sqrt * torch.cos(az)) * X + (el_sqrt * torch.sin(az)) * Y + torch.sqrt(1 - el + 1e-7) * Z
        d_pdf = sat(d_dirs, Z) / PI * (torch.cos((1 - el) * PI / 2) * PI / 2)
        fd, fs = bool(flow_lobes[0]), bool(flow_lobes[1])
        va = ops.view_angles(normals, view_dirs)
        flow_jit = (lambda n: torch.rand(pn, n, device=dev)) if (is_train and self.training) else (lambda n: None)      # flow.py:86-87

        def flow_dirs(flow_copy, n, half):
            """A frozen copy's samples as directions (:1084-1134 / :1163-1203): -> dirs [pn,n,3], pdf [pn,n,1], x [pn,n,2], jac [pn,n,1]."""
            with torch.no_grad():
                ang, lq = flow_copy._sample_nograd(pts, va, n, flow_jit(n))
                ph_, th_ = ang[..., :1] * (2 * PI), ang[..., 1:2] * (0.5 * PI)
                Hf = (torch.sin(th_) * torch.cos(ph_)) * X + (torch.sin(th_) * torch.sin(ph_)) * Y + torch.cos(th_) * Z
                if half:
                    HoV_f = sat(V, Hf)
                    dirs_f, jac_f = HoV_f * Hf * 2 - V, 4 * PI ** 2 * HoV_f * torch.sin(th_)
                else:
                    dirs_f, jac_f = Hf, PI ** 2 * torch.sin(th_)
                pdf_f = torch.exp(-lq.clamp(-8, 8)) / jac_f.clamp_min(EPS)
            return dirs_f, pdf_f, ang, jac_f

        if fd:          # the flow's samples first, then the fixed set (:1136-1139)
            fdirs, fpdf, d_x, d_jac = flow_dirs(self.flow_diffuse_copy, cfg["nis_diffuse_sample_num"], cfg["use_half_diffuse"])
            d_dirs, d_pdf = torch.cat([fdirs, d_dirs.expand(pn, -1, 3)], 1), torch.cat([fpdf, d_pdf.expand(pn, -1, 1)], 1)
        nd = d_dirs.shape[1]
        d_lights, _ = lights_of(pts[:, None].expand(pn, nd, 3).reshape(-1, 3), d_dirs.reshape(-1, 3),
                                human_poses[:, None].expand(pn, nd, 3, 4).reshape(-1, 3, 4) if human_poses is not None else None)
        d_lights = d_lights.view(pn, nd, 3)
        kd = 1 - metallic[:, None]
        d_w = albedo[:, None] * kd * (sat(d_dirs, Z) / PI)
        diffuse = torch.mean(d_w * d_lights / d_pdf.clamp_min(EPS), 1)
        # ---- specular lobe: GGX half vectors warped by the predicted (squared) roughness
        azs, els = self._fixed_s[:, 0][None, :, None], self._fixed_s[:, 1][None, :, None]
        a = roughness[:, None]
        cos_t = ((1.0 - els) / (1.0 + (a ** 2 - 1.0) * els).clamp_min(EPS)).clamp_min(EPS).sqrt()
        sin_t = (1 - cos_t ** 2).clamp_min(EPS).sqrt()
        phi = azs * (2 * PI)
        if jitter:
            phi = (phi + torch.rand(pn, 1, 1, device=dev) * (2 * PI)) % (2 * PI)
        Hs = (torch.cos(phi) * sin_t) * X + (torch.sin(phi) * sin_t) * Y + cos_t * Z
        VoH = sat(V, Hs)
        s_dirs = VoH * Hs * 2 - V
        NoH_s = cos_t.clamp_min(0.0)
        ggx = lambda noh, r: r ** 2 / (PI * (noh ** 2 * (r ** 2 - 1.0) + 1.0) ** 2).clamp_min(EPS)
        s_pdf = ggx(NoH_s, a) * NoH_s / (4 * VoH).clamp_min(EPS) * (torch.cos((1 - els) * PI / 2) * PI / 2)
        angles_H = torch.cat([phi.expand(pn, -1, -1), torch.arcsin(sin_t).expand(pn, -1, -1)], -1)
        if fs:          # the specular set IS the flow's samples (:1163-1203)
            s_dirs, s_pdf, s_x, s_jac = flow_dirs(self.flow_specular_copy, cfg["nis_specular_sample_num"], cfg["use_half_specular"])
        ns = s_dirs.shape[1]
        smask = (s_dirs * Z).sum(-1) > 0
        rid = torch.arange(pn, device=dev)[:, None].expand(pn, ns)[smask]
        sd_, sp_ = s_dirs[smask], s_pdf[smask]
        sah = None if fs else angles_H[smask]
        F0 = 0.04 * (1 - metallic) + metallic * albedo
        Hh = F.normalize(view_dirs[rid] + sd_, dim=-1)
        HoV_s = torch.clamp((Hh * view_dirs[rid]).sum(-1, keepdim=True), 0.0, 1.0)
        fres = F0[rid] + (1.0 - F0[rid]) * torch.clamp(1.0 - HoV_s, 0.0, 1.0) ** 5.0
        NoV = sat(normals, view_dirs)[rid]
        NoL = sat(normals[rid], sd_)
        g1 = lambda c, r: c / (c * (1 - r / 2) + r / 2 + 1e-5)
        if cfg["geometry_type"] == "ggx_smith":              # geometry_ggx_smith_correlated (fields.py:1000-1008)
            lam = lambda a2, c: 0.5 * torch.sqrt(1 + a2 * (1 - c ** 2) / (c ** 2 + 1e-7)) - 0.5
            geo = 1.0 / (1.0 + lam(roughness[rid] ** 2, NoV) + lam(roughness[rid] ** 2, NoL))
        else:
            geo = g1(NoV, roughness[rid]) * g1(NoL, roughness[rid])
        dist = ggx(sat(normals[rid], Hh), roughness[rid])
        s_lights, s_hit, s_hl = self._lights_of(pts[rid], sd_, human_poses[rid] if human_poses is not None else None)
        s_w = dist * fres * geo / (4 * NoV).clamp_min(EPS)
        specular = torch.zeros(pn, 3, device=dev).index_add(0, rid, s_w * s_lights / sp_.clamp_min(EPS)) / ns
        colors = self._linear_to_srgb(diffuse + specular)
        from ..shading import LazyOutputs
        outputs = LazyOutputs({"albedo": albedo, "roughness": roughness, "metallic": metallic, "normal": (normals + 1) / 2, "specular_mask": smask,
                               "diffuse_light": self._linear_to_srgb(d_lights.mean(1), clamp01=True)})
        with torch.no_grad():                                   # the rest of the dict (fields.py:1232-1256, :1288-1291), as written there
            c01 = lambda t: self._linear_to_srgb(t, clamp01=True)
            seg = lambda v: torch.zeros(pn, v.shape[-1], device=dev).index_add(0, rid, v)
            sh_f = s_hit.float()[:, None]
            outputs["specular_light"] = c01(seg(s_lights) / ns)
            outputs["diffuse_color"], outputs["specular_color"] = c01(diffuse), c01(specular)
            outputs["approximate_light"] = c01(torch.mean(kd * d_lights, 1) + outputs["specular_color"])
            outputs["visibility"] = 1 - seg(sh_f) / ns
            outputs["indirect_light"] = seg(s_lights * sh_f) / ns
            gd_ = (d_w * d_lights).mean(-1, keepdim=True) / d_pdf.clamp_min(EPS)
            outputs["variance_diffuse_vis"] = torch.var(gd_, dim=1, unbiased=True) / cfg["diffuse_sample_num"]
            gs_ = (s_w * s_lights).mean(-1, keepdim=True) / sp_.clamp_min(EPS)
            outputs["variance"] = torch.var(gs_)
            outputs["variance_specular_vis"] = (seg(gs_ ** 2) / ns - (seg(gs_) / ns) ** 2) / ns
        outputs.set_lazy("human_lights", lambda: s_hl if s_hl is not None else torch.zeros(int((~s_hit).sum()), 3, device=dev))
        outputs.set_lazy("inter", lambda: self._bvh.trace(pts[rid].contiguous(), sd_.detach().contiguous(), 1e-5, 2 * self.unit_size)[0])
        zero = torch.zeros((), device=dev)
        outputs["loss_nis_diffuse"] = outputs["loss_nis_specular"] = zero
        if cfg["use_nis_diffuse"] and step is not None and step >= cfg.get("nis_loss_iter_diffuse", 500):
            sdn = cfg["nis_diffuse_sample_num"]
            if fd:      # fitted on the copy's own samples (:1271-1279): x = the sampled angles, Jacobian of their direction map
                ph, th, jac = None, None, d_jac
            elif cfg["use_half_diffuse"]:
                Hd = F.normalize(V + d_dirs[:, :sdn], dim=-1)
                HoV_d = torch.clamp((Hd * V).sum(-1, keepdim=True), 0.0, 1.0)
                ph, th = half_angles(Hd)
                jac = 4 * PI ** 2 * HoV_d * torch.sin(th)
            else:      # :1276-1279: the flow is fitted on the directions' own angles (az, arcsin(sqrt(el))) of sample_diffuse_directions
                ph, th = az.expand(pn, nd, 1)[:, :sdn], torch.arcsin(el_sqrt).expand(pn, nd, 1)[:, :sdn]
                jac = PI ** 2 * torch.sin(th)
            xq = d_x.clamp(EPS, 1 - EPS) if fd else torch.cat([ph / (2 * PI), th / (0.5 * PI)], -1).clamp(EPS, 1 - EPS)
            _, logq = self.flow_diffuse(pts, va, roughness.detach(), xq.detach().contiguous(), return_jacobian=True)
            logqx = logq - jac.clamp_min(EPS).log()
            outputs["loss_nis_diffuse"] = -((d_w * d_lights)[:, :sdn] * logqx / d_pdf.expand(pn, nd, 1)[:, :sdn].clamp_min(EPS)).mean()
        if cfg["use_nis_specular"] and step is not None and step >= cfg.get("nis_loss_iter_specular", 500):
            if fs:      # (:1294-1317)
                ph, th, jac = None, None, s_jac[smask]
            elif cfg["use_half_specular"]:
                ph, th = sah[:, :1], sah[:, 1:2]
                jac = 4 * PI ** 2 * HoV_s * torch.sin(th)
            else:      # :1314-1317: ... on the angles of the reflected directions themselves (sample_specular_directions' `angles`)
                cz = (Z.expand(pn, ns, 3)[smask] * sd_).sum(-1, keepdim=True).clamp(-1 + EPS, 1 - EPS)
                ph = (torch.atan2((Y.expand(pn, ns, 3)[smask] * sd_).sum(-1, keepdim=True), (X.expand(pn, ns, 3)[smask] * sd_).sum(-1, keepdim=True))
                      + 2 * PI) % (2 * PI)
                th = torch.acos(cz)
                jac = PI ** 2 * torch.sin(th)
            xq = s_x[smask].clamp(EPS, 1 - EPS) if fs else torch.cat([ph / (2 * PI), th / (0.5 * PI)], -1).clamp(EPS, 1 - EPS)
            _, logq = self.flow_specular(pts, va, roughness.detach(), xq.contiguous(), return_jacobian=True, rays_id=rid)
            logqx = logq - jac.clamp_min(EPS).log()
            outputs["loss_nis_specular"] = -(s_w * s_lights * logqx / sp_.clamp_min(EPS)).mean()
        outputs["loss_nis"] = outputs["loss_nis_diffuse"] + outputs["loss_nis_specular"]
        return colors, outputs

    def forward_all(self, pts, view_dirs, normals, step=None, is_train=True, human_poses=None, use_flow=None):
        """shade_mixed_all (fields.py:1337-1451; cfg shade_fn = 'shade_mixed_all'): ONE direction set per point -- the samples of the single
        flow's frozen copy (cfg use_nis_all, once update_step has made the copy; nis_sample_num of them) or the fixed cosine set -- on which
        the diffuse AND the specular weight are evaluated, nothing masked; from nis_loss_iter on the flow is fitted on the set in use.
        A composed pass like forward_train_fixed: per-direction algebra in device-resident differentiable ops, visibility, cube map, VM
        gather, flows and every MLP product on the HIP kernels.  use_flow: None = the training rule (copy active), True / False = the
        two passes of an inference call (nis_sample, :1467-1468)."""
        EPS, PI = 1e-6, math.pi
        dev, pn, cfg = pts.device, pts.shape[0], self.cfg
        view_dirs, normals = F.normalize(view_dirs, dim=-1), F.normalize(normals, dim=-1)
        metallic, roughness, albedo = self.predict_materials(pts)
        sat = lambda a, b: torch.clamp((a * b).sum(-1, keepdim=True), 0.0, 1.0)
        if not hasattr(self, "_bvh"):
            self._bvh = ops.Bvh(self.ray_tracer[0], self.ray_tracer[1], dev)
            self._fixed = fibonacci_samples(cfg["diffuse_sample_num"]).to(dev)
        z = normals
        x0 = torch.stack([z[:, 1], -z[:, 0], torch.zeros_like(z[:, 0])], -1)
        x1 = torch.stack([-z[:, 2], torch.zeros_like(z[:, 0]), z[:, 0]], -1)
        x = F.normalize(torch.where((x0.norm(dim=-1) > x1.norm(dim=-1))[:, None], x0, x1), dim=-1)
        y = torch.cross(z, x, dim=-1)
        X, Y, Z, V = x[:, None], y[:, None], z[:, None], view_dirs[:, None]
        va = ops.view_angles(normals, view_dirs)
        if use_flow is None:
            use_flow = self.use_flow_copy
        use_flow = bool(use_flow) and bool(cfg["use_nis_all"])
        training = is_train and self.training
        if use_flow:
            with torch.no_grad():
                sn = cfg["nis_sample_num"]
                ang, lq = self.flow_copy._sample_nograd(pts, va, sn, torch.rand(pn, sn, device=dev) if training else None)
                ph_s, th_s = ang[..., :1] * (2 * PI), ang[..., 1:2] * (0.5 * PI)
                Hf = (torch.sin(th_s) * torch.cos(ph_s)) * X + (torch.sin(th_s) * torch.sin(ph_s)) * Y + torch.cos(th_s) * Z
                if cfg["use_half_all"]:
                    HoV_f = sat(V, Hf)
                    dirs, jac_s = HoV_f * Hf * 2 - V, 4 * PI ** 2 * HoV_f * torch.sin(th_s)
                else:
                    dirs, jac_s = Hf, PI ** 2 * torch.sin(th_s)
                pdf = torch.exp(-lq.clamp(-8, 8)) / jac_s.clamp_min(EPS)
        else:           # sample_diffuse_directions (:824-856)
            az, el = self._fixed[:, 0][None, :, None] * (2 * PI), self._fixed[:, 1][None, :, None]
            if training and cfg["random_azimuth"]:
                az = (az + torch.rand(pn, 1, 1, device=dev) * (2 * PI)) % (2 * PI)
            el_sqrt = torch.sqrt(el + 1e-7)
            dirs = (el_sqrt * torch.cos(az)) * X + (el_sqrt * torch.sin(az)) * Y + torch.sqrt(1 - el + 1e-7) * Z
            pdf = sat(dirs, Z) / PI * (torch.cos((1 - el) * PI / 2) * PI / 2)
        sn = dirs.shape[1]
        poses = human_poses[:, None].expand(pn, sn, 3, 4).reshape(-1, 3, 4) if human_poses is not None else None
        lights, hit, hl = self._lights_of(pts[:, None].expand(pn, sn, 3).reshape(-1, 3), dirs.reshape(-1, 3), poses)
        lights, hit = lights.view(pn, sn, 3), hit.view(pn, sn)
        kd = 1 - metallic[:, None]
        d_w = albedo[:, None] * kd * (sat(dirs, Z) / PI)
        diffuse = torch.mean(d_w * lights / pdf.clamp_min(EPS), 1)
        F0 = (0.04 * (1 - metallic) + metallic * albedo)[:, None]
        Hh = F.normalize(V + dirs, dim=-1)
        HoV = torch.clamp((Hh * V).sum(-1, keepdim=True), 0.0, 1.0)
        fres = F0 + (1.0 - F0) * torch.clamp(1.0 - HoV, 0.0, 1.0) ** 5.0
        NoV, NoL, a = sat(normals, view_dirs)[:, None], sat(Z, dirs), roughness[:, None]
        if cfg["geometry_type"] == "ggx_smith":
            lam = lambda a2, c: 0.5 * torch.sqrt(1 + a2 * (1 - c ** 2) / (c ** 2 + 1e-7)) - 0.5
            geo = 1.0 / (1.0 + lam(a ** 2, NoV) + lam(a ** 2, NoL))
        else:
            g1 = lambda c, r: c / (c * (1 - r / 2) + r / 2 + 1e-5)
            geo = g1(NoV, a) * g1(NoL, a)
        dist = a ** 2 / (PI * (sat(Z, Hh) ** 2 * (a ** 2 - 1.0) + 1.0) ** 2).clamp_min(EPS)
        s_w = dist * fres * geo / (4 * NoV).clamp_min(EPS)
        specular = torch.mean(s_w * lights / pdf.clamp_min(EPS), 1)
        colors = self._linear_to_srgb(diffuse + specular)
        c01 = lambda t: self._linear_to_srgb(t, clamp01=True)
        hit_f = hit.float()
        fx = (d_w + s_w) * lights
        outputs = {"albedo": albedo, "normal": (normals + 1) / 2, "roughness": roughness, "metallic": metallic,
                   "human_lights": hl if hl is not None else torch.zeros(int((~hit).sum()), 3, device=dev),
                   "diffuse_light": c01(lights.mean(1)), "specular_light": c01(lights.mean(1)), "diffuse_color": c01(diffuse),
                   "specular_color": c01(specular), "approximate_light": c01(torch.mean(kd * lights, 1) + c01(specular)),
                   "visibility": 1 - hit_f.mean(1, keepdim=True), "indirect_light": torch.mean(lights * hit_f[..., None], 1),
                   "variance": torch.var(fx.mean(-1, keepdim=True) / pdf.clamp_min(EPS))}
        outputs["loss_nis"] = torch.zeros((), device=dev)
        if cfg["use_nis_all"] and step is not None and step >= cfg["nis_loss_iter"]:
            if use_flow:
                xq, jac = ang.clamp(EPS, 1 - EPS), jac_s
            elif cfg["use_half_all"]:          # the sampler's half angles (:850-856) with the BRDF's HoV (:1435)
                cz = (Z * Hh).sum(-1, keepdim=True).clamp(-1 + EPS, 1 - EPS)
                ph = (torch.atan2((Y * Hh).sum(-1, keepdim=True), (X * Hh).sum(-1, keepdim=True)) + 2 * PI) % (2 * PI)
                th = torch.acos(cz)
                xq, jac = torch.cat([ph / (2 * PI), th / (0.5 * PI)], -1).clamp(EPS, 1 - EPS), 4 * PI ** 2 * HoV * torch.sin(th)
            else:                              # the directions' own angles (:843)
                ph, th = az.expand(pn, sn, 1), torch.arcsin(el_sqrt).expand(pn, sn, 1)
                xq, jac = torch.cat([ph / (2 * PI), th / (0.5 * PI)], -1).clamp(EPS, 1 - EPS), PI ** 2 * torch.sin(th)
            _, logq = self.flow(pts, va, roughness.detach(), xq.detach().contiguous(), return_jacobian=True)
            logqx = logq - jac.clamp_min(EPS).log()
            outputs["loss_nis"] = -(fx * logqx / pdf.clamp_min(EPS)).mean()
        return colors, outputs

    def forward(self, pts, view_dirs, normals, human_poses=None, step=None, is_train=False):
        """fields.py:1453-1473 with the flow samplers active: -> (colors [pn,3], outputs dict).
        With autograd enabled (training) the differentiable composition is used; otherwise the fused inference path."""
        if self.cfg["human_lights"] and human_poses is None:
            raise ValueError("human_lights=True: forward() needs the per-point human_poses [pn,3,4]")
        if self.cfg["shade_fn"] == "shade_mixed_all":
            grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
            any_nis = self.cfg["use_nis_all"] or self.cfg["use_nis_diffuse"] or self.cfg["use_nis_specular"]          # self.use_nis, :670
            if step is not None or grad or not any_nis:
                with torch.set_grad_enabled(grad):
                    return self.forward_all(pts, view_dirs, normals, step=step, is_train=is_train, human_poses=human_poses)
            with torch.no_grad():             # :1467-1471: nis_sample False, then nis_sample True with the `_nis` suffix
                colors, outputs = self.forward_all(pts, view_dirs, normals, None, is_train, human_poses, use_flow=False)
                c_nis, o_nis = self.forward_all(pts, view_dirs, normals, None, is_train, human_poses, use_flow=True)
            o_nis["rgb_pr"] = c_nis
            outputs.update({k + "_nis": v for k, v in o_nis.items()})
            return colors, outputs
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # fields.py:1081,1160: a lobe draws from its flow copy once the copy exists (update_step at nis_start_iter) -- and never with
            # cfg use_nis_diffuse / use_nis_specular = False; until then from its fixed sampler
            fd = bool(self.cfg["use_nis_diffuse"]) and getattr(self, "use_flow_diffuse_copy", False)
            fs = bool(self.cfg["use_nis_specular"]) and getattr(self, "use_flow_specular_copy", False)
            if step is None:                     # (nis_sample True, :1467-1473: every lobe that has a flow samples it)
                fd, fs = bool(self.cfg["use_nis_diffuse"]), bool(self.cfg["use_nis_specular"])
            if fd and fs:
                return self.forward_train(pts, view_dirs, normals, step=step, is_train=is_train, human_poses=human_poses)
            # both fixed samplers, or the mixed states (one copy active, or a lobe without its flow): the composed pass
            return self.forward_train_fixed(pts, view_dirs, normals, step=step, is_train=is_train, human_poses=human_poses, flow_lobes=(fd, fs))
        return self._forward_eval(pts, view_dirs, normals, human_poses)

    @torch.no_grad()
    def _forward_eval(self, pts, view_dirs, normals, human_poses=None):
        """fields.py:1467-1473 with step=None: the fixed-sampler pass gives `colors` and the un-suffixed outputs, the flow-sampler
        pass the `*_nis` outputs (`rgb_pr_nis` = its colours)."""
        from ..shading import LazyOutputs, aux_outputs
        env_human = self.cfg["human_lights"] and self.cfg["outer_light_version"] == "envlight"       # (MCShader holds the net outer lights with humans)
        if env_human or not (self._fused_flows and self.cfg["use_nis_diffuse"] and self.cfg["use_nis_specular"]):
            return self._forward_eval_composed(pts, view_dirs, normals, human_poses)
        sh = self.shader()
        nrm = (F.normalize(normals, dim=-1) + 1) / 2
        # the unweighted light maps (diffuse_light, visibility ...) average over EVERY ray, incl. those whose BRDF weight is zero:
        # aux=True switches the zero-weight culling of the throughput path off and takes the statistics from tf_shade_reduce_aux
        hp = human_poses.float().contiguous() if (human_poses is not None and self.cfg["human_lights"]) else None
        dsn = self.cfg["diffuse_sample_num"]
        fx = sh.shade_fixed(pts, view_dirs, normals, human_poses=hp, aux=True)
        outputs = LazyOutputs({"albedo": fx["albedo"], "roughness": fx["roughness"], "metallic": fx["metallic"], "normal": nrm,
                               **aux_outputs(fx, dsn)})
        nis = sh.shade(pts, view_dirs, normals, self.cfg["nis_diffuse_sample_num"], self.cfg["nis_specular_sample_num"], human_poses=hp, aux=True)
        outputs.update({k + "_nis": v for k, v in {"albedo": nis["albedo"], "roughness": nis["roughness"], "metallic": nis["metallic"],
                                                  "normal": nrm, "rgb_pr": nis["colors"], **aux_outputs(nis, dsn)}.items()})
        zero = torch.zeros((), device=pts.device)
        for sfx in ("", "_nis"):                               # step is None: no NIS loss (fields.py:1285,1329-1330)
            outputs["loss_nis_diffuse" + sfx] = outputs["loss_nis_specular" + sfx] = outputs["loss_nis" + sfx] = zero
        # data-dependent lengths (one host sync each): built when somebody reads them -- nothing outside shade_mixed does
        for sfx, o in (("", fx), ("_nis", nis)):
            outputs.set_lazy("inter" + sfx, lambda o=o: o["inter"])
            outputs.set_lazy("human_lights" + sfx, lambda o=o: o["human_lights"])
        outputs.set_lazy("specular_rays_id_nis", lambda: nis["specular_rays_id"])
        return fx["colors"], outputs

    @torch.no_grad()
    def _forward_eval_composed(self, pts, view_dirs, normals, human_poses=None):
        """_forward_eval for what the fused pass does not instantiate (cfg flow_diffuse / flow_specular != 'pwquad', a lobe without its
        flow: use_nis_diffuse / use_nis_specular = False): the fixed-sampler pass
        and the flow pass are the two training compositions (forward_train_fixed / forward_train: every stage a HIP kernel or a
        device-resident torch op) evaluated without autograd, no jitter -- correct, not fast.  Same dict as the fused pass."""
        from ..shading import LazyOutputs
        hp = human_poses.float().contiguous() if (human_poses is not None and self.cfg["human_lights"]) else None
        c_fix, o_fix = self.forward_train_fixed(pts, view_dirs, normals, step=None, is_train=False, human_poses=hp)
        fd, fs = bool(self.cfg["use_nis_diffuse"]), bool(self.cfg["use_nis_specular"])
        if fd and fs:
            c_nis, o_nis = self.forward_train(pts, view_dirs, normals, step=None, is_train=False, human_poses=hp)
        else:            # a lobe without its flow keeps its fixed sampler in the flow pass too (fields.py:1081, :1160)
            c_nis, o_nis = self.forward_train_fixed(pts, view_dirs, normals, step=None, is_train=False, human_poses=hp, flow_lobes=(fd, fs))
        outputs = LazyOutputs()
        for src, sfx in ((o_fix, ""), (o_nis, "_nis")):
            keys = set(dict.keys(src)) | set(getattr(src, "_lazy", {}))
            for k in keys:
                if k.startswith("loss_nis") or k == "specular_mask":
                    continue
                if dict.__contains__(src, k):
                    outputs[k + sfx] = src[k]
                else:
                    outputs.set_lazy(k + sfx, lambda src=src, k=k: src[k])
        outputs["rgb_pr_nis"] = c_nis
        outputs.set_lazy("specular_rays_id_nis", lambda: torch.nonzero(o_nis["specular_mask"])[:, 0])
        zero = torch.zeros((), device=pts.device)
        for sfx in ("", "_nis"):
            outputs["loss_nis_diffuse" + sfx] = outputs["loss_nis_specular" + sfx] = outputs["loss_nis" + sfx] = zero
        return c_fix, outputs


def _mlp(seq, x):
    """A predictor's nn.Sequential on the device: every Linear + activation pair is one tf_linear_fwd launch, differentiated by
    tf_linear_bwd (autograd.mlp_apply) -- no library GEMM in a training step.  (CPU tensors -- module construction tests -- take the
    plain module call.)"""
    if x.is_cuda:
        from ..autograd import mlp_apply
        return mlp_apply(seq, x.contiguous())
    return seq(x)


def _predictor3(feats_dim, out_dim, final, run_dim=128):
    """make_predictor_3layer (network/other_field.py:50-84): weight-normed Linear-ReLU-Linear-ReLU-Linear + final activation."""
    wn = nn.utils.parametrizations.weight_norm
    return nn.Sequential(wn(nn.Linear(feats_dim, run_dim)), nn.ReLU(), wn(nn.Linear(run_dim, run_dim)), nn.ReLU(),
                         wn(nn.Linear(run_dim, out_dim)), final)


class _Exp(nn.Module):
    """ExpActivation (other_field.py:12-18)."""

    def __init__(self, max_light):
        super().__init__()
        self.max_light = max_light

    def forward(self, x):
        return torch.exp(torch.clamp(x, max=self.max_light))


class SingleVarianceNetwork(nn.Module):
    """other_field.py:193-207: one scalar parameter; inv_s = exp(10 v) (cfg std_act = 'exp', every shipped config), 10 v ('linear') or
    (10 v)^2 ('square')."""

    def __init__(self, init_val, activation="exp"):
        super().__init__()
        if activation not in ("exp", "linear", "square"):
            raise NotImplementedError(f"std_act {activation!r}: 'exp', 'linear' or 'square' (other_field.py:200-207)")
        self.act = activation
        self.register_parameter("variance", nn.Parameter(torch.tensor(float(init_val))))

    def inv_s(self):
        if self.act == "exp":
            return torch.exp(self.variance * 10.0)
        return self.variance * 10.0 if self.act == "linear" else (self.variance * 10.0) ** 2

    def forward(self, x):
        return torch.ones([*x.shape[:-1], 1], device=x.device) * self.inv_s()


class ShapeShadingNetwork(nn.Module):
    """Split-sum shading of the shape stage (reference: network/fields.py:319-575) with the reference's parameter names.

    Without autograd the whole forward is ONE launch of tf_shape_shade_fwd (three 128-wide MLPs, encodings, cube taps, FG
    LUT, sRGB).  With autograd the same arithmetic is composed from differentiable device ops: the MLP products on tf_linear_fwd /
    tf_linear_bwd (autograd.mlp_apply: no library GEMM), the cube lookups on the HIP autograd ops of EnvLight, so that gradients reach the MLPs, the environment map and -- through the
    normals / reflective directions / roughness -- the SDF.  `rad_mlp` (has_radiance_field: 161-128-128-3 on [feat, xyz,
    embed4(view), normal], fields.py:407-417,476-483) is a torch module in both modes; human lights are not built."""
    default_cfg = {"human_light": False, "sphere_direction": False, "light_pos_freq": 8, "inner_init": -0.95, "light_exp_max": 0.0,
                   "app_feats_dim": 128, "has_radiance_field": False, "radiance_field_step": 0, "mat_pos_multires": -1,
                   "fg_lut_path": "assets/bsdf_256_256.bin"}

    def __init__(self, cfg, device="cuda"):
        super().__init__()
        import os
        from ..synth import synthetic_fg_lut
        self.cfg = {**self.default_cfg, **cfg}
        # Variants no shipped configs/shape file sets (human_light / sphere_direction / mat_pos_multires >= 0, fields.py:344,354-357,
        # 367-370,394-404): built as the reference builds them and evaluated by the differentiable composition below in every mode --
        # the one-launch inference kernel instantiates the default (mat_mlp on the 128 features alone, no capturer reflection).
        mp = int(self.cfg["mat_pos_multires"])
        self._mat_pos_dim = 0 if mp < 0 else 3 if mp == 0 else 3 + 6 * mp
        self._composed_only = bool(self.cfg["human_light"]) or self._mat_pos_dim > 0
        fd, em = self.cfg["app_feats_dim"], self.cfg["light_exp_max"]
        if self.cfg["has_radiance_field"]:
            self.rad_mlp = _predictor3(fd + 3 + 27 + 3, 3, nn.Sigmoid())      # pos_multires=0 (raw xyz), dir_multires=4
        self.mat_mlp = _predictor3(fd + self._mat_pos_dim, 5, nn.Sigmoid())
        path = self.cfg["fg_lut_path"]
        if path and os.path.exists(path):
            lut = torch.from_numpy(np.fromfile(path, dtype=np.float32).reshape(1, 256, 256, 2))
        else:   # analytic stand-in until load_state_dict brings the table of a reference checkpoint (buffer 'FG_LUT')
            lut = synthetic_fg_lut()
        self.register_buffer("FG_LUT", lut)
        pos_dim = 3 + 6 * self.cfg["light_pos_freq"]
        # present in reference checkpoints, unused by forward (fields.py:428); 'sphere_direction' doubles its input (:354-357)
        self.outer_light = _predictor3(144 if self.cfg["sphere_direction"] else 72, 3, _Exp(em))
        nn.init.constant_(self.outer_light[-2].bias, np.log(0.5))
        self.envlight = EnvLight(trainable=True, max_res=128, device=device)
        self.inner_light = _predictor3(pos_dim + 72, 3, _Exp(em))
        nn.init.constant_(self.inner_light[-2].bias, np.log(0.5))
        self.inner_weight = _predictor3(pos_dim + 39, 1, nn.Identity())
        nn.init.constant_(self.inner_weight[-2].bias, self.cfg["inner_init"])
        if self.cfg["human_light"]:      # the light reflected from the photo capturer (fields.py:367-370): IPE(2 x 2 x 6) -> 4, exp(min(., 0))
            self.human_light_predictor = _predictor3(24, 4, _Exp(0.0))
            nn.init.constant_(self.human_light_predictor[-2].bias, np.log(0.01))
        self.to(device)
        self._op, self._op_version = None, None

    def get_optparam_groups(self, lr_init_network, lr_init_envlight):
        return [{"params": self.envlight.parameters(), "lr": lr_init_envlight},
                {"params": [p for n, p in self.named_parameters() if "envlight" not in n], "lr": lr_init_network}]

    # ------------------------------------------------------------------ fused inference path
    def _fused(self):
        env = self.envlight
        if not hasattr(env, "specular"):
            env.build_mips()
        ver = tuple(p._version for p in self.parameters()) + tuple(id(s) for s in env.specular) + (id(env.diffuse), self.FG_LUT._version)
        if self._op is None or ver != self._op_version:
            nets = {}
            for name in ("mat_mlp", "inner_light", "inner_weight"):
                seq = getattr(self, name)
                nets[name] = [(seq[i].weight.detach().contiguous(), seq[i].bias.detach().contiguous()) for i in (0, 2, 4)]
            self._op = ops.ShapeShade(nets, [s.detach() for s in env.specular], env.diffuse.detach(), self.FG_LUT,
                                      env.min_roughness, env.max_roughness, self.cfg["light_exp_max"])
            self._op_version = ver
        return self._op

    def _wants_radiance(self, step):
        return bool(self.cfg["has_radiance_field"]) and step is not None and step > self.cfg["radiance_field_step"]

    def _radiance(self, points, normals, view_dirs, feat):
        """fields.py:476-483 on normalised / patched normals and normalised view directions."""
        from ..encodings import posenc
        return _mlp(self.rad_mlp, torch.cat([feat, points, posenc(view_dirs, 4), normals], -1))

    @staticmethod
    def _unit_inputs(normals, view_dirs):
        normals = F.normalize(normals, dim=-1)
        bad = (normals[:, :2].sum(-1) == 0.0)[:, None]
        normals = torch.where(bad, _const3(0.0, 1e-6, 1.0, normals.device), normals)
        return normals, F.normalize(view_dirs, dim=-1)

    # ------------------------------------------------------------------ differentiable composition
    def _mat_input(self, points, feat):
        """fields.py:488-494: the material net's input -- the features, with mat_pos_multires >= 0 the (embedded) position behind them."""
        if self._mat_pos_dim == 0:
            return feat
        from ..encodings import posenc
        mp = int(self.cfg["mat_pos_multires"])
        return torch.cat([feat, points if mp == 0 else posenc(points, mp)], -1)

    def predict_human_light(self, points, reflective, human_poses, roughness):
        """fields.py:377-392 with get_camera_plane_intersection (utils/network_utils.py:69-88) and IPE (:56-61: E[sin] of a Gaussian
        with mean 2^k m and variance 4^k v) -> (human_lights [N,3], human_weights [N,1])."""
        R, t = human_poses[:, :, :3], human_poses[:, :, 3]
        p_ = torch.einsum("nij,nj->ni", R, points) + t
        d_ = torch.einsum("nij,nj->ni", R, reflective)
        hits = d_[:, 2].abs() > 1e-4
        dz = torch.where(hits, d_[:, 2], torch.full_like(d_[:, 2], 1e-4))     # (the reference writes 1e-4 through a view of dirs_)
        d_ = torch.cat([d_[:, :2], dz[:, None]], -1)
        dist = -p_[:, 2] / dz
        mean = (p_ + dist[:, None] * d_)[:, :2] * 0.3
        var = roughness * (dist[:, None] * 0.3) ** 2
        hits = (hits & (mean.norm(dim=-1) < 1.5) & (dist > 0)).float()[:, None]
        mean, var = mean * hits, (var * hits).expand(-1, 2)
        sc = 2.0 ** torch.arange(6, device=mean.device)
        sm = (mean[:, None, :] * sc[:, None]).reshape(-1, 12)
        sv = (var[:, None, :] * (sc ** 2)[:, None]).reshape(-1, 12)
        pe = torch.exp(-0.5 * torch.cat([sv, sv], -1)) * torch.sin(torch.cat([sm, sm + 0.5 * math.pi], -1))
        h = _mlp(self.human_light_predictor, pe) * hits
        return h[:, :3], h[:, 3:].clamp(0.0, 1.0)

    def _composed(self, points, normals, view_dirs, feat, inter_results, want_rad=False, human_poses=None):
        from ..encodings import ide5, linear_to_srgb, posenc
        env = self.envlight
        if not hasattr(env, "specular"):
            env.build_mips()
        if (points.is_cuda and not inter_results and not self.cfg["human_light"] and not view_dirs.requires_grad and not points.requires_grad
                and self.FG_LUT.is_cuda and self.FG_LUT.is_contiguous()):
            # the default training form: the element-wise algebra as two differentiable launches (tf_shape_glue_*: round 6; ~150
            # element-wise launches of this composition and autograd's mirror image before), nets / lookups / encodings in between
            from ..autograd import ShapeGluePostFn, ShapeGluePreFn
            mat = _mlp(self.mat_mlp, self._mat_input(points, feat))
            normals, view_dirs, NoV, reflective, roughness, mip = ShapeGluePreFn.apply(
                normals.contiguous(), view_dirs.contiguous(), mat, (env.min_roughness, env.max_roughness, len(env.specular)))
            pts = posenc(points, self.cfg["light_pos_freq"])
            indirect_light = _mlp(self.inner_light, torch.cat([pts, ide5(reflective, roughness)], -1))
            occ_raw = _mlp(self.inner_weight, torch.cat([pts.detach(), posenc(reflective.detach(), 6)], -1))
            color, occ_prob = ShapeGluePostFn.apply(mat, NoV, env(normals), env(reflective, roughness, mip=mip), indirect_light, occ_raw, self.FG_LUT)
            occ_info = {"reflective": reflective, "occ_prob": occ_prob, "roughness": roughness}
            return color, (self._radiance(points, normals, view_dirs, feat) if want_rad else None), occ_info
        normals, view_dirs = self._unit_inputs(normals, view_dirs)
        NoV = (normals * view_dirs).sum(-1, keepdim=True)
        reflective = NoV * normals * 2 - view_dirs
        mat = _mlp(self.mat_mlp, self._mat_input(points, feat))
        albedo, roughness, metallic = mat[..., :3] * 0.77 + 0.03, mat[..., 3:4] * 0.9 + 0.09, mat[..., 4:]
        diffuse_albedo = (1 - metallic) * albedo
        diffuse_light = env(normals)
        diffuse_color = diffuse_albedo * diffuse_light
        specular_albedo = 0.04 * (1 - metallic) + metallic * albedo
        direct_light = env(reflective, roughness)
        pts = posenc(points, self.cfg["light_pos_freq"])
        indirect_light = _mlp(self.inner_light, torch.cat([pts, ide5(reflective, roughness)], -1))
        # (the occlusion net's inputs are detached in the reference, fields.py:434: the embedding runs on the detached direction, as
        # one launch of the encoding kernel instead of twelve differentiable sin / cos and a concatenation)
        occ_prob = _mlp(self.inner_weight, torch.cat([pts.detach(), posenc(reflective.detach(), 6)], -1)) * 0.5 + 0.5
        occ = occ_prob.clamp(0, 1)
        if self.cfg["human_light"]:
            if human_poses is None:
                raise ValueError("ShapeShadingNetwork(human_light=True): forward() needs the per-sample human_poses [N,3,4]")
            hl, hw = self.predict_human_light(points, reflective, human_poses, roughness)
            human = hl * hw
            specular_light = indirect_light * occ + (human + direct_light * (1 - hw)) * (1 - occ)
        else:
            human = None
            specular_light = indirect_light * occ + direct_light * (1 - occ)
        # FG LUT: dr.texture(filter_mode='linear', boundary_mode='clamp'), texel centres at (i + .5) / n
        uv = torch.cat([NoV.clamp(0, 1), roughness.clamp(0, 1)], -1)
        lut = self.FG_LUT.reshape(self.FG_LUT.shape[-3], self.FG_LUT.shape[-2], 2).permute(2, 0, 1)[None]
        fg = F.grid_sample(lut, (uv * 2 - 1)[None, :, None, :], mode="bilinear", padding_mode="border", align_corners=False)[0, :, :, 0].t()
        specular_ref = specular_albedo * fg[:, 0:1] + fg[:, 1:2]
        specular_color = specular_ref * specular_light
        color = linear_to_srgb(diffuse_color + specular_color).clamp(0.0, 1.0)
        occ_info = {"reflective": reflective, "occ_prob": occ_prob, "roughness": roughness}
        if not inter_results:
            return color, (self._radiance(points, normals, view_dirs, feat) if want_rad else None), occ_info
        c01 = lambda t: t.clamp(0.0, 1.0)
        inter = {"specular_albedo": specular_albedo, "specular_ref": c01(specular_ref), "specular_direct_light": direct_light,
                 "specular_light": c01(linear_to_srgb(specular_light)), "specular_color": c01(linear_to_srgb(specular_color)),
                 "diffuse_albedo": diffuse_albedo, "diffuse_light": c01(linear_to_srgb(diffuse_light)),
                 "diffuse_color": c01(linear_to_srgb(diffuse_color)), "metallic": metallic, "roughness": roughness, "albedo": albedo,
                 "occ_prob": c01(occ_prob), "indirect_light": indirect_light * occ}
        if human is not None:
            inter["human_light"] = linear_to_srgb(human)
        return color, occ_info, inter

    def forward(self, points, normals, view_dirs, feature_vectors, human_poses=None, inter_results=False, step=None):
        """fields.py:448-567 -> (color [N,3], None, occ_info) or, with inter_results, (color, occ_info, intermediate dict)."""
        want_rad = self._wants_radiance(step) and not inter_results
        if points.shape[0] == 0:
            z = lambda c: torch.zeros(0, c, device=points.device)
            occ_info = {"reflective": z(1), "occ_prob": z(1), "roughness": z(1)}
            return (z(3), occ_info, {}) if inter_results else (z(3), z(3) if want_rad else None, occ_info)
        needs_graph = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or normals.requires_grad
                                                   or feature_vectors.requires_grad)
        if needs_graph or inter_results or self._composed_only:
            return self._composed(points, normals, view_dirs, feature_vectors, inter_results, want_rad, human_poses)
        color, occ_prob, roughness, reflective = self._fused()(points, normals, view_dirs, feature_vectors)
        rad = self._radiance(points, *self._unit_inputs(normals, view_dirs), feature_vectors) if want_rad else None
        return color, rad, {"reflective": reflective, "occ_prob": occ_prob, "roughness": roughness}

    def predict_materials(self, points, feature_vectors):
        """fields.py:569-575: raw mat_mlp outputs (no albedo / roughness remapping, as in the reference)."""
        mat = _mlp(self.mat_mlp, self._mat_input(points, feature_vectors))
        return mat[..., 4:], mat[..., 3:4], mat[..., :3]
