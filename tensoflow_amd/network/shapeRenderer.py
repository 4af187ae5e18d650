"""ShapeRenderer (reference: network/shapeRenderer.py:100-1310) on the HIP kernels -- the drop-in module of the shape stage.

Same constructor (`ShapeRenderer(cfg, training)`), sub-module names (`sdf_network`, `deviation_network`, `color_network`),
`state_dict` keys and call surface as the reference for everything on the hot path:

    render_core / render / sample_ray / compute_sdf_alpha / compute_alpha / near_far_from_sphere / compute_ball_radii
    updateAlphaMask / upsample_sdf_grid / get_train_opt_params / ckpt_to_save / load_ckpt / get_anneal_val / nvs

Without autograd every field evaluation, the split-sum shading and the compositing scan are fused HIP launches
(march.render_core).  With autograd (training) the same quantities come from the autograd ops of tensoflow_amd.autograd
(SdfAlphaFn, CompositeFn) and the differentiable ShapeShadingNetwork, so `loss.backward()` reaches every parameter the
reference trains.  The dataset side of the reference class (`_init_dataset`, `train_step`, `test_step`: image / pose tables,
ray shuffling) is outside the hot path: construct with training=False and feed ray batches to `render`.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import march, ops
from ..autograd import CompositeFn, SdfAlphaFn
from ..surface import _neus_weights
from .fields import ShapeShadingNetwork, SingleVarianceNetwork, TensoSDF, TVLoss, _const3


class AlphaGridMask(nn.Module):
    """shapeRenderer.py:79-97: binary occupancy volume [D,H,W] over `aabb`; sample_alpha = trilinear fetch (align_corners)."""

    def __init__(self, device, aabb, alpha_volume):
        super().__init__()
        self.device = device
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).to(device)
        self.alpha_volume = alpha_volume.view(1, 1, *alpha_volume.shape[-3:]).float().to(device)
        self._mask = march.AlphaMask(self.aabb.cpu(), self.alpha_volume[0, 0])

    def sample_alpha(self, xyz_sampled):
        """-> [N] float; only `> 0` is ever read (shapeRenderer.py:1120, :301), which tf_alpha_mask_sample evaluates bit-exactly."""
        return self._mask.alive(xyz_sampled.reshape(-1, 3).contiguous()).float()


class ShapeRenderer(nn.Module):
    default_cfg = {
        "std_act": "exp", "inv_s_init": 0.3, "freeze_inv_s_step": None, "val_geometry": False, "shader_config": {},
        "n_samples": 64, "n_importance": 64, "up_sample_steps": 4, "perturb": 1.0, "anneal_end": 50000,
        "train_ray_num": 1024, "test_ray_num": 2048, "clip_sample_variance": True,
        "apply_occ_loss": True, "apply_tv_loss": True, "apply_sparse_loss": True, "apply_hessian_loss": True,
        "apply_gaussian_loss": False, "occ_loss_step": 20000, "occ_loss_max_pn": 2048, "occ_sdf_thresh": 0.01, "gaussianLoss_step": 20000,
        "device": "cuda", "gridSize": [512, 512, 512], "aabb": [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], "step_ratio": 0.5,
        "alphaMask_thres": 0.0001, "marched_weights_thres": 0.0001, "sdf_n_comp": 16, "app_n_comp": 36, "sdf_dim": 128,
        "app_dim": 128, "sdf_multires": 0, "max_levels": 1, "has_radiance_field": False, "radiance_field_step": 0,
        "predict_BG": True, "isBGWhite": True, "nerfDataType": False, "mul_length": 10, "use_occ_grid": False,
        "occ_grid_reso": 128, "blend_ratio": 0,
        # dataset side (training=True)
        "database_name": "", "dataset_dir": "", "split_manul": False, "apply_mask_loss": False, "rgb_loss": "charbonier",
        "test_downsample_ratio": True, "downsample_ratio": 0.25,
    }

    def __init__(self, cfg, training=True):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        if training and not (self.cfg["nerfDataType"] and str(self.cfg.get("database_name", "")).split("/")[0] in ("tensoSDF", "tensoIR", "nerf", "orb")):
            raise NotImplementedError("the dataset side of ShapeRenderer (_init_dataset / train_step) reads the Blender-convention "
                                      "layouts (database_name 'tensoSDF/', 'tensoIR/', 'nerf/' or 'orb/<scene>', nerfDataType=True): for anything else "
                                      "construct with training=False and pass ray batches to render()")
        if self.cfg["predict_BG"]:
            raise NotImplementedError("predict_BG (NeRF++ background) raises in the reference's render_core as well (:1109); "
                                      "set predict_BG=False as configs/shape/* do")
        self.device = self.cfg["device"]
        aabb = self.cfg["aabb"]
        self.aabb = torch.tensor(aabb.cpu().tolist() if isinstance(aabb, torch.Tensor) else aabb, dtype=torch.float32, device=self.device)
        self.center = self.aabb.mean(0).float().view(1, 1, 3)
        self.radius = (self.aabb[1] - self.center).mean().float()
        # host copies of the scalars the kernels take by value: read once here, not by a device sync in every step
        self._aabb_cpu, self._radius_f = self.aabb.cpu(), float(self.radius)
        self.alphaMask = None
        # use_occ_grid (configs/shape/syn/compressor_occ.yaml:21; shapeRenderer.py:213-216): the build's own occupancy grid in the
        # role of nerfacc.OccGridEstimator -- EMA occupancy state on the device, sampling by tf_march_uniform (march.OccGrid)
        self.occ_grid = march.OccGrid(self.aabb.reshape(-1).cpu(), self.cfg["occ_grid_reso"], self.device) if self.cfg["use_occ_grid"] else None
        self.step_ratio = self.cfg["step_ratio"]
        self.alphaMask_thres = self.cfg["alphaMask_thres"]
        self.marched_weights_thres = self.cfg["marched_weights_thres"]
        self.sdf_n_comp, self.app_n_comp = self.cfg["sdf_n_comp"], self.cfg["app_n_comp"]
        self.sdf_dim, self.app_dim = self.cfg["sdf_dim"], self.cfg["app_dim"]
        self.update_stepSize(torch.tensor(self.cfg["gridSize"]), self.cfg["max_levels"])
        self.sdf_network = TensoSDF(self.gridSize.cpu(), self.aabb.cpu(), device=self.device, init_n_levels=self.max_levels,
                                    sdf_n_comp=self.sdf_n_comp, sdf_dim=self.sdf_dim, app_dim=self.app_dim,
                                    sdf_multires=self.cfg["sdf_multires"])
        self.deviation_network = SingleVarianceNetwork(self.cfg["inv_s_init"], self.cfg["std_act"]).to(self.device)
        self.cfg["shader_config"] = {**self.cfg["shader_config"], "occ_loss_step": self.cfg["occ_loss_step"],
                                     "has_radiance_field": self.cfg["has_radiance_field"],
                                     "radiance_field_step": self.cfg["radiance_field_step"], "app_feats_dim": self.cfg["app_dim"]}
        self.color_network = ShapeShadingNetwork(self.cfg["shader_config"], device=self.device)
        self.sdf_inter_fun = lambda x: self.sdf_network.sdf(x, None)
        self.tv_reg = TVLoss()
        if training:
            self._init_dataset()

    # ------------------------------------------------------------------------------ bookkeeping
    def update_stepSize(self, gridSize, max_levels):
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invaabbSize = 2.0 / self.aabbSize
        self.gridSize = torch.tensor(torch.as_tensor(gridSize).cpu().tolist(), dtype=torch.int32).to(self.device)
        self._grid_size_cpu = self.gridSize.float().cpu()        # read back once per resolution change, not once per step (a blocking copy)
        self.max_levels = max_levels
        self.units = self.aabbSize / (self.gridSize - 1)
        self.stepSize = torch.mean(self.units) * self.step_ratio
        self.base_radii = self.aabbSize[0] / 2.0 / self.gridSize[0]
        self._units_f, self._base_radii_f = [float(u) for u in self.units], float(self.base_radii)
        self.nSamples = self.cfg["n_samples"] + self.cfg["n_importance"]

    def get_kwargs(self):
        return {"aabb": self.aabb, "gridSize": self.gridSize.tolist(), "sdf_n_comp": self.sdf_n_comp,
                "appearance_n_comp": self.app_n_comp, "sdf_dim": self.sdf_dim, "app_dim": self.app_dim,
                "sdf_multires": self.cfg["sdf_multires"], "alphaMask_thres": self.alphaMask_thres,
                "marched_weights_thres": self.marched_weights_thres, "step_ratio": self.step_ratio, "max_levels": self.max_levels}

    def ckpt_to_save(self):
        """shapeRenderer.py:343-353: kwargs + state_dict (+ bit-packed alpha mask)."""
        ckpt = {"kwargs": self.get_kwargs(), "network_state_dict": self.state_dict()}
        if self.alphaMask is not None:
            vol = self.alphaMask.alpha_volume.bool().cpu().numpy()
            ckpt.update({"alphaMask.shape": vol.shape, "alphaMask.mask": np.packbits(vol.reshape(-1)),
                         "alphaMask.aabb": self.alphaMask.aabb.cpu()})
        if self.occ_grid is not None:
            ckpt["occ_grid_state_dict"] = self.occ_grid.state_dict()
        return ckpt

    def update_occ_grid(self, step):
        """shapeRenderer.py:1286-1290: every 100 steps the occupancy grid takes the EMA of compute_alpha at jittered cell positions
        (all cells during the first 10 000 steps)."""
        if self.occ_grid is not None:
            return self.occ_grid.update_every_n_steps(step=step, occ_eval_fn=lambda x: self.compute_alpha(x), n=100, warmup_steps=10000)
        return False

    def load_ckpt(self, ckpt):
        """shapeRenderer.py:355-362 (strict: every key of a reference checkpoint has a home, incl. the Gaussian-blur buffers)."""
        if "alphaMask.aabb" in ckpt:
            length = int(np.prod(ckpt["alphaMask.shape"]))
            vol = torch.from_numpy(np.unpackbits(ckpt["alphaMask.mask"])[:length].reshape(ckpt["alphaMask.shape"]))
            self.alphaMask = AlphaGridMask(self.device, ckpt["alphaMask.aabb"], vol.float())
        if self.occ_grid is not None and "occ_grid_state_dict" in ckpt:
            self.occ_grid.load_state_dict(ckpt["occ_grid_state_dict"])
        self.load_state_dict(ckpt["network_state_dict"])

    def upsample_sdf_grid(self, res_target):
        new_res, max_levels = self.sdf_network.upsample_volume_grid(torch.as_tensor(res_target))
        self.update_stepSize(new_res, max_levels)

    def get_train_opt_params(self, learning_rate_xyz, learning_rate_net, learning_rate_env):
        grad_vars = self.sdf_network.get_optparam_groups(learning_rate_xyz, learning_rate_net)
        grad_vars += [{"params": self.deviation_network.parameters(), "lr": learning_rate_net}]
        grad_vars += self.color_network.get_optparam_groups(learning_rate_net, learning_rate_env)
        return grad_vars

    def get_anneal_val(self, step):
        return 1.0 if self.cfg["anneal_end"] < 0 else float(np.min([1.0, step / self.cfg["anneal_end"]]))

    # ------------------------------------------------------------------------------ field views
    def _field(self):
        """Eval-side view of sdf_network for the march composition (shares the packed pyramid of the module)."""
        net = self.sdf_network
        f = march.SdfField.__new__(march.SdfField)
        f.planes, f.lines = [p.detach() for p in net.sdf_plane], [p.detach() for p in net.sdf_line]
        f.W = [w.detach() for w in net._w()]
        f.aabb = self._aabb_cpu
        f.aabb_dev = self.aabb
        f.grid_size = self._grid_size_cpu
        f.units = list(self._units_f)
        f.n_levels = self.max_levels
        f.device = self.device
        f.packed = net._field()
        return f

    def _inv_s(self):
        return self.deviation_network.inv_s().clip(1e-6, 1e6)

    def _inv_s_host(self):
        """float(inv_s) for the kernels' scalar argument, read back when the parameter has changed (its version counter) instead of on
        every call: a read-back is a host synchronisation in the middle of the step's forward."""
        v = self.deviation_network.variance
        key = (v.data_ptr(), v._version)
        if getattr(self, "_inv_s_key", None) != key:
            self._inv_s_key, self._inv_s_val = key, float(self._inv_s().detach())
        return self._inv_s_val

    def near_far_from_sphere(self, rays_o, dirs):
        return march.near_far_from_sphere(rays_o, dirs, self._radius_f)

    @staticmethod
    def compute_ball_radii(distance, radiis, cos):
        return march.ball_radii(distance, radiis, cos)

    # ------------------------------------------------------------------------------ occupancy
    @torch.no_grad()
    def updateAlphaMask(self, gridSize=(128, 128, 128)):
        """shapeRenderer.py:257-283 -> new_aabb [2,3]; the lattice evaluation runs in tf_sdf_forward."""
        prev = None if self.alphaMask is None else self.alphaMask._mask
        mask, new_aabb = march.update_alpha_mask(self._field(), float(self._inv_s()), grid=tuple(gridSize), thres=self.alphaMask_thres,
                                                 mul_length=self.cfg["mul_length"], prev=prev)
        self.alphaMask = AlphaGridMask(self.device, self.aabb, mask.volume.float())
        return new_aabb

    @torch.no_grad()
    def compute_alpha(self, points):
        """shapeRenderer.py:972-993: NeuS opacity of one march step at `points`."""
        if points.shape[0] == 0:
            return torch.zeros(0, device=self.device)
        sdf = self.sdf_network.sdf(points)[:, 0]
        inv_s = self._inv_s()
        pc = torch.sigmoid((sdf + self.stepSize * 0.5) * inv_s)
        nc = torch.sigmoid((sdf - self.stepSize * 0.5) * inv_s)
        return ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)

    # ------------------------------------------------------------------------------ sampling + rendering
    @torch.no_grad()
    def sample_ray(self, rays_o, dirs, near, far, perturb, radiis=None, rays_cos=None):
        """shapeRenderer.py:871-932 -> packed t_starts, t_ends, ray_indices (int64; bit-exact against the reference for
        perturb = 0).  perturb > 0: one uniform offset per ray of +-1/n_samples (:888-890, torch.rand on the device);
        clip_sample_variance: the up-sampling sharpness 64 * 2^i is capped by the learned inv_s (:905-907)."""
        t_rand = (torch.rand(rays_o.shape[0], 1, device=rays_o.device) - 0.5) if perturb > 0 else None
        cap = self._inv_s_host() if self.cfg["clip_sample_variance"] else None
        return march.sample_ray(self._field(), rays_o, dirs, near, far, radiis, rays_cos, self._base_radii_f,
                                n_samples=self.cfg["n_samples"], n_importance=self.cfg["n_importance"],
                                up_steps=self.cfg["up_sample_steps"], t_rand=t_rand, inv_s_cap=cap)

    def compute_sdf_alpha(self, points, level, dists, dirs, cos_anneal_ratio, step, is_train):
        """shapeRenderer.py:995-1025 -> alpha, gradients, feature_vector, inv_s [N], sdf, hessian (None when not training)."""
        N = points.shape[0]
        net = self.sdf_network
        if N == 0:
            z = lambda *s: torch.zeros(*s, device=self.device)
            return z(0), z(0, 3), z(0, net.app_dim), z(0), z(0), z(0, 3)
        inv_s = self._inv_s()
        if self.cfg["freeze_inv_s_step"] is not None and step is not None and step < self.cfg["freeze_inv_s_step"]:
            inv_s = inv_s.detach()
        lv = None if level is None else level.reshape(-1).contiguous()
        units = list(self._units_f)
        inv_s._tf_host = self._inv_s_host()               # SdfAlphaFn takes the scalar from here
        if torch.is_grad_enabled() and any(p.requires_grad for p in list(net.parameters()) + [self.deviation_network.variance]):
            alpha, grad, feat, sdf, nh = SdfAlphaFn.apply(points.contiguous(), lv, dists.contiguous(), dirs.contiguous(), inv_s,
                                                          float(cos_anneal_ratio), self._aabb_cpu, units, self.max_levels,
                                                          *net.sdf_plane, *net.sdf_line, *net._w())
        else:
            alpha, grad, feat, sdf, nh = ops.sdf_alpha(net._field(), *[w.detach() for w in net._w()], points, lv, dists, dirs,
                                                       self._aabb_cpu, units, inv_s._tf_host, float(cos_anneal_ratio), want_hess=is_train)
        return alpha, grad, feat, inv_s.expand(N), sdf, (nh if is_train else None)

    def render(self, ray_batch, near, far, human_poses=None, perturb_overwrite=-1, cos_anneal_ratio=0.0, is_train=True, step=None):
        """shapeRenderer.py:934-963."""
        perturb = self.cfg["perturb"] if perturb_overwrite < 0 else perturb_overwrite
        o, d, dirs, radiis, cos = ray_batch["rays_o"], ray_batch["rays_d"], ray_batch["dirs"], ray_batch["radiis"], ray_batch["rays_cos"]
        if self.occ_grid is not None:                     # shapeRenderer.py:950-959
            ridx, t0, t1 = self.occ_grid.sampling(o, dirs, near_plane=float(near.min()), far_plane=float(far.max()),
                                                  render_step_size=float(self.stepSize), stratified=is_train)
        else:
            t0, t1, ridx = self.sample_ray(o, dirs, near, far, perturb, radiis=radiis, rays_cos=cos)
        return self.render_core(o, d, dirs, radiis, cos, t0, t1, ridx, human_poses, cos_anneal_ratio=cos_anneal_ratio, step=step,
                                is_train=is_train)

    def render_core(self, rays_o, rays_d, viewdirs, radiis, rays_cos, t_starts, t_ends, ray_indices, human_poses=None,
                    cos_anneal_ratio=0.0, step=None, is_train=True):
        """shapeRenderer.py:1105-1277 (white background).  Output keys as the reference's: ray_rgb, gradient_error, acc, sample_num,
        normal, std, loss_sparse, loss_hessian (+ the validation keys when is_train is False)."""
        rn = rays_o.shape[0]
        if self.alphaMask is not None:
            mid = (t_starts + t_ends) * 0.5
            dists = t_ends - t_starts
            pts = rays_o[ray_indices] + viewdirs[ray_indices] * mid[:, None]
            keep = self.alphaMask.sample_alpha(pts) > 0
            ray_indices, mid, dists = ray_indices[keep], mid[keep], dists[keep]
        N = ray_indices.shape[0]
        if rays_o.is_cuda and self.alphaMask is None and N > 0 and not (rays_o.requires_grad or viewdirs.requires_grad):
            # points, view directions and mip levels of the packed samples in one launch (round 5; was ~20 element-wise launches)
            mid, dists, viewdir, points, levels = ops.sample_points(rays_o, viewdirs, radiis, rays_cos, ray_indices, t_starts, t_ends,
                                                                    self._base_radii_f)
        else:
            if self.alphaMask is None:
                mid, dists = (t_starts + t_ends) * 0.5, t_ends - t_starts
            viewdir = viewdirs[ray_indices]
            points = rays_o[ray_indices] + viewdir * mid[:, None]
            levels = torch.log2(self.compute_ball_radii(mid[:, None], radiis[ray_indices], rays_cos[ray_indices]) / self.base_radii)
        alpha, gradients, feat, inv_s, sdf, hessian = self.compute_sdf_alpha(points, levels, dists, viewdir, cos_anneal_ratio, step, is_train)
        fused_norm = gradients.is_cuda and gradients.dtype == torch.float32
        if fused_norm:      # F.normalize + the eikonal residual (:1137, :1145) in one launch each way (round 6: 6 + 26 element-wise launches before)
            from ..autograd import Normalize3Fn
            normals, gradient_error = Normalize3Fn.apply(gradients.contiguous(), None, None, True)
        else:
            normals = F.normalize(gradients, dim=-1)
            gradient_error = (torch.linalg.norm(gradients, ord=2, dim=-1) - 1.0) ** 2
        # (per-sample capturer poses, shapeRenderer.py:1139: read by the human_light variant of the shading only)
        poses_pt = human_poses[ray_indices] if (human_poses is not None and self.color_network.cfg["human_light"]) else None
        color, radiance, occ_info = self.color_network(points, normals, -viewdir, feat, poses_pt, step=step)
        zero = torch.zeros(1, device=rays_o.device)
        vals = [color, gradients] + ([radiance, occ_info["roughness"]] if radiance is not None else [])
        vals = torch.cat(vals, -1).contiguous()
        if torch.is_grad_enabled() and (alpha.requires_grad or vals.requires_grad):
            weights, acc, out = CompositeFn.apply(alpha, vals, ray_indices, rn)
        else:                                               # the kernel composites up to 8 value channels per launch
            outs = []
            for c0 in range(0, vals.shape[1], 8):
                weights, acc, o = ops.composite(alpha, ray_indices, vals[:, c0:c0 + 8].contiguous(), rn)
                outs.append(o)
            out = outs[0] if len(outs) == 1 else torch.cat(outs, -1)
        acc = acc[:, None]
        rgb = out[:, :3]
        if self.cfg["isBGWhite"]:
            rgb = rgb + (1 - acc)
        if fused_norm:      # the composited ray normal (:1207-1208): blend with (0, 0, 1) by the opacity + F.normalize, one launch each way
            normal = Normalize3Fn.apply(out[:, 3:6].contiguous(), acc, (0.0, 0.0, 1.0), False)[0]
        else:
            normal = F.normalize(out[:, 3:6] * acc + (1.0 - acc) * _const3(0.0, 0.0, 1.0, rays_o.device), dim=-1)
        outputs = {"ray_rgb": rgb, "gradient_error": gradient_error, "acc": acc, "sample_num": N / max(rn, 1), "normal": normal,
                   "std": torch.mean(1 / inv_s) if N > 0 else zero}
        if radiance is not None:                                   # has_radiance_field and step > radiance_field_step (:1195-1206)
            outputs["radiance"] = out[:, 6:9] + (1 - acc) if self.cfg["isBGWhite"] else out[:, 6:9]
            outputs["roughness_weights"] = out[:, 9].clone().detach()
        if self.cfg["apply_occ_loss"]:
            outputs["loss_occ"] = self.compute_occ_loss(occ_info, points, sdf, normals, viewdir, step) if N > 0 else zero
        if self.cfg["apply_gaussian_loss"] and step is not None and step > self.cfg["gaussianLoss_step"]:
            outputs["loss_gaussian"] = self.sdf_network.grid_gaussian_loss() if N > 0 else zero
        if self.cfg["apply_tv_loss"]:
            outputs["loss_tv_sdf"] = self.sdf_network.TV_loss_sdf(self.tv_reg)
        if self.cfg["apply_sparse_loss"]:
            outputs["loss_sparse"] = torch.exp(-20.0 * sdf.abs()).mean() if N > 0 else zero
        if self.cfg["apply_hessian_loss"]:
            outputs["loss_hessian"] = hessian.abs().mean() if (hessian is not None and N > 0) else zero
        if step is not None and step < 1000:
            outputs["sdf_pts"], outputs["sdf_vals"] = (points, sdf) if N > 0 else (zero, zero)
        if not is_train:
            outputs.update(self._validation_outputs(rays_o, viewdirs, radiis, rays_cos, ray_indices, mid, weights, acc, normal, step, human_poses))
        return outputs

    @torch.no_grad()
    def _traced_occlusion(self, points, dirs, sn0, sn1):
        """Sum of the NeuS weights along (points, dirs) up to the unit sphere: get_intersection (utils/network_utils.py:172-202),
        sn0 uniform + sn1 importance field evaluations per ray in tf_sdf_forward -> [n,1] (0 for points outside r = 0.999)."""
        occ = torch.zeros(points.shape[0], 1, device=points.device)
        inside = points.norm(dim=-1) < 0.999
        if bool(inside.any()):
            p, d = points[inside].contiguous(), dirs[inside].contiguous()
            dtx, xtx = (p * d).sum(-1, keepdim=True), (p ** 2).sum(-1, keepdim=True)
            max_dist = -dtx + torch.sqrt((dtx ** 2 - xtx + 1).clamp(min=0) + 1e-6)
            field, inv_s = self._field(), float(self._inv_s())
            z = max_dist * torch.linspace(0, 1, sn0, device=p.device)[None]
            w = _neus_weights(field, inv_s, z, p, d)
            z_new = march._sample_pdf_det(z, w, sn1)
            occ[inside] = _neus_weights(field, inv_s, z_new, p, d).sum(-1, keepdim=True)
        return occ

    def compute_occ_loss(self, occ_info, points, sdf, gradients, dirs, step):
        """shapeRenderer.py:1027-1103: L1 between the predicted occlusion probability (inner_weight net) and the traced one on
        up to occ_loss_max_pn near-surface, front-facing samples."""
        zero = torch.zeros(1, device=points.device)
        if step is None or step < self.cfg["occ_loss_step"]:
            return zero
        inner = ~((self.aabb[0] > points) | (points > self.aabb[1])).any(-1)
        mask = inner & ((gradients * dirs).sum(-1) < 0) & (sdf.abs() < self.cfg["occ_sdf_thresh"])
        idx = torch.nonzero(mask)[:, 0]
        if idx.numel() > self.cfg["occ_loss_max_pn"]:
            idx = idx[torch.randperm(idx.numel(), device=idx.device)[:self.cfg["occ_loss_max_pn"]]].sort().values
        if idx.numel() == 0:
            return zero
        gt = self._traced_occlusion(points[idx].detach(), occ_info["reflective"][idx].detach(), 64, 16)
        return F.l1_loss(occ_info["occ_prob"].index_select(0, idx), gt)      # (index_select: its backward is an atomic index_add, not the sort of x[idx])

    @torch.no_grad()
    def _validation_outputs(self, rays_o, viewdirs, radiis, rays_cos, ray_indices, mid, weights, acc, normal, step, human_poses=None):
        """Eval branch of render_core (shapeRenderer.py:1239-1275): expected depth -> surface point -> materials / lights there,
        traced occlusion probability along the reflected direction (get_intersection, utils/network_utils.py:172-202)."""
        rn = rays_o.shape[0]
        t_depth = torch.zeros(rn, device=rays_o.device).index_add_(0, ray_indices, weights.detach() * mid)[:, None]
        points = t_depth * viewdirs + rays_o
        level = torch.log2(self.compute_ball_radii(t_depth, radiis, rays_cos) / self.base_radii)
        gradients, _ = self.sdf_network.gradient(points.contiguous(), level.reshape(-1).contiguous(), training=False)
        normals = F.normalize(gradients, dim=-1)
        inner = ~((self.aabb[0] > points) | (points > self.aabb[1])).any(-1)[:, None]
        out = {"normal_vis": ((normal + 1.0) * 0.5) * acc + (1.0 - acc), "depth": t_depth * rays_cos}
        if not self.cfg["nerfDataType"]:
            out["normal_vis"] = ((normals + 1.0) * 0.5) * inner
        feat = self.sdf_network(points.contiguous(), level.reshape(-1).contiguous())[..., 1:]
        poses = human_poses if (human_poses is not None and self.color_network.cfg["human_light"]) else None
        _, occ_info, inter = self.color_network(points, normals, -viewdirs, feat.contiguous(), poses, inter_results=True, step=step)
        # traced occlusion along the reflected ray: 128 uniform + 9 importance field evaluations per pixel
        occ_gt = self._traced_occlusion(points, occ_info["reflective"], 128, 9)
        out["occ_prob_gt"] = occ_gt
        out.update({k: v * inner for k, v in inter.items()})
        return out

    # ------------------------------------------------------------------------------ novel view
    @torch.no_grad()
    def nvs(self, pose, K, h, w, rank=None, world=None):
        """shapeRenderer.py:569-668 (nerfDataType ray construction) -> dict of [h,w,C] numpy arrays.  The reference renders 2048
        rays per pass (launch-bound); here `test_ray_num` rays per pass, default raised by the caller as HBM allows.  With world > 1
        (default: torch.distributed's rank / world when a process group is up) this rank marches rows dist.shard_range(h * w, rank,
        world) of the frame and one all-gather assembles the maps on every rank."""
        if not self.cfg["nerfDataType"]:
            raise NotImplementedError("the reference's non-NeRF ray construction raises as well (:577)")
        dev = self.device
        K = torch.from_numpy(np.asarray(K, np.float32)).to(dev)
        pose = torch.from_numpy(np.asarray(pose, np.float32)).to(dev)
        i, j = torch.meshgrid(torch.linspace(0, w - 1, w, device=dev), torch.linspace(0, h - 1, h, device=dev), indexing="ij")
        i, j = i.t(), j.t()
        rays_d = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
        dx = (rays_d[:, :-1, :] - rays_d[:, 1:, :]).norm(dim=-1, keepdim=True)
        dx = torch.cat([dx, dx[:, -2:-1, :]], 1)
        dy = (rays_d[:-1, :, :] - rays_d[1:, :, :]).norm(dim=-1, keepdim=True)
        dy = torch.cat([dy, dy[-2:-1, :, :]], 0)
        radiis = torch.sqrt(dx * dy / torch.pi).reshape(-1, 1)
        rays_d = rays_d.reshape(-1, 3)
        rays_cos = 1 / rays_d.norm(dim=-1, keepdim=True)
        rays_o = pose[:3, -1].expand(h * w, 3)
        rays_d = (rays_d[:, None, :] * pose[:3, :3]).sum(-1)
        dirs = F.normalize(rays_d, dim=-1)
        keys = {"color": "ray_rgb", "albedo": "albedo", "roughness": "roughness", "normal": "normal", "normal_vis": "normal_vis",
                "occ_predict": "occ_prob", "occ_trace": "occ_prob_gt", "diff_color": "diffuse_color", "spec_color": "specular_color",
                "diff_light": "diffuse_light", "spec_light": "specular_light", "indirect_light": "indirect_light"}
        from .. import dist as tdist
        rank, world = tdist.rank_world(rank, world)
        lo, hi = tdist.shard_range(h * w, rank, world)
        output = {k: [] for k in keys}
        trn = self.cfg["test_ray_num"]
        for ri in range(lo, hi, trn):
            sl = slice(ri, min(ri + trn, hi))
            batch = {"rays_o": rays_o[sl].contiguous(), "rays_d": rays_d[sl].contiguous(), "dirs": dirs[sl].contiguous(),
                     "radiis": radiis[sl].contiguous(), "rays_cos": rays_cos[sl].contiguous()}
            near, far = self.near_far_from_sphere(batch["rays_o"], batch["rays_d"])
            cur = self.render(batch, near, far, None, is_train=False, step=300000)
            for k, src in keys.items():
                output[k].append(cur[src].detach().reshape(sl.stop - sl.start, -1))
        local = {k: (torch.cat(v, 0) if v else torch.zeros(0, 1, device=dev)) for k, v in output.items()}
        if world > 1 and lo == hi:        # (more ranks than rows: this rank contributes nothing, its maps still need the right widths)
            raise RuntimeError("ShapeRenderer.nvs: fewer pixels than ranks")
        full = tdist.gather_maps(local, h * w, rank, world)
        return {k: v.reshape(h, w, -1).cpu().numpy() for k, v in full.items()}

    # ------------------------------------------------------------------------------ dataset side (TensoSDF synthetic scenes)
    def _init_dataset(self):
        """shapeRenderer.py:383-409: database, manual split (first 100 images train, the rest thinned for validation,
        dataset/database.py:824-833), CPU-resident ray table of every training pixel, shuffled.  cfg['rank'] / cfg['world'] select
        this process's stride of each batch."""
        from ..dataset import RayTable, construct_ray_batch_nerf, parse_database_name
        self.database = parse_database_name(self.cfg["database_name"], self.cfg["dataset_dir"], white_bg=self.cfg["isBGWhite"])
        ids = self.database.get_img_ids()
        if self.cfg.get("split_manul", False):
            border = self.cfg.get("split_borderline", 100)
            self.train_ids, test = ids[:border], ids[border:]
            self.test_ids = test[::50] if len(test) > 10 else test[::4]
        else:                                            # get_database_split(split_type='validation'), dataset/database.py:834-839
            import random
            ids = list(ids)
            random.Random(6033).shuffle(ids)             # random.seed(6033); random.shuffle(img_ids)
            self.test_ids, self.train_ids = ids[:1], ids[1:]
        self.train_imgs_info = self.database.imgs_info(self.train_ids)
        self.test_imgs_info = self.database.imgs_info(self.test_ids)
        self.train_num, self.test_num = len(self.train_ids), len(self.test_ids)
        batch, self.tbn, _, _ = construct_ray_batch_nerf(self.train_imgs_info)
        batch, self.tbn = self.filtering_train_rays(batch)
        self.train_table = RayTable(batch, rank=self.cfg.get("rank", 0), world=self.cfg.get("world", 1),
                                    seed=self.cfg.get("random_seed", 6033), device=self.device)

    @torch.no_grad()
    def filtering_train_rays(self, batch, chunk=1 << 22):
        """shapeRenderer.py:539-566: rays whose slab interval with the aabb is empty never enter the training table."""
        o, d = batch["rays_o"], batch["dirs"]
        aabb = self.aabb.cpu()
        keep = []
        for i in range(0, o.shape[0], chunk):
            oo, dd = o[i:i + chunk], d[i:i + chunk]
            vec = torch.where(dd == 0, torch.full_like(dd, 1e-6), dd)
            ra, rb = (aabb[1] - oo) / vec, (aabb[0] - oo) / vec
            keep.append(torch.maximum(ra, rb).amin(-1) > torch.minimum(ra, rb).amax(-1))
        keep = torch.cat(keep) if keep else torch.zeros(0, dtype=torch.bool)
        return {k: v[keep] for k, v in batch.items()}, int(keep.sum())

    def compute_rgb_loss(self, rgb_pr, rgb_gt):
        from ..trainer import rgb_loss
        return rgb_loss(self.cfg["rgb_loss"], rgb_pr, rgb_gt)

    def train_step(self, step):
        """shapeRenderer.py:777-794."""
        batch = self.train_table.next_batch(self.cfg["train_ray_num"])
        near, far = self.near_far_from_sphere(batch["rays_o"], batch["dirs"])
        outputs = self.render(batch, near, far, batch["human_poses"], -1, self.get_anneal_val(step), is_train=True, step=step)
        outputs["loss_rgb"] = self.compute_rgb_loss(outputs["ray_rgb"], batch["rgbs"])
        outputs["psnr"] = 20 * torch.log10(1.0 / torch.sqrt(F.mse_loss(outputs["ray_rgb"], batch["rgbs"])))
        if "radiance" in outputs:
            outputs["loss_radiance"] = self.compute_rgb_loss(outputs["radiance"], batch["rgbs"]) * outputs["roughness_weights"]
            outputs["loss_rgb"] = outputs["loss_rgb"] * (1.0 - outputs["roughness_weights"])
        if self.cfg["apply_mask_loss"]:
            outputs["loss_mask"] = F.binary_cross_entropy(outputs["acc"].clip(1e-3, 1.0 - 1e-3), (batch["masks"] > 0.5).float())
        return outputs

    @torch.no_grad()
    def test_step(self, index, step):
        """shapeRenderer.py:720-775: one validation image, `test_ray_num` rays per pass.  `test_downsample_ratio`: box average by the
        integer factor 1 / downsample_ratio (what cv2.INTER_AREA computes for integer factors); intrinsics scaled like the image."""
        from ..dataset import construct_ray_batch_nerf
        info = {k: v[index:index + 1] for k, v in self.test_imgs_info.items()}
        if self.cfg.get("test_downsample_ratio", False) and self.cfg.get("downsample_ratio", 1) != 1:
            f = round(1.0 / self.cfg["downsample_ratio"])
            if abs(f * self.cfg["downsample_ratio"] - 1.0) > 1e-6 or info["imgs"].shape[-1] % f or info["imgs"].shape[-2] % f:
                raise NotImplementedError("downsample_ratio must be 1 / integer dividing the image size")
            Ks = info["Ks"].clone()
            Ks[:, :2] = Ks[:, :2] / f
            info = {"imgs": F.avg_pool2d(info["imgs"], f), "masks": F.avg_pool2d(info["masks"], f), "Ks": Ks, "poses": info["poses"]}
        batch, rn, h, w = construct_ray_batch_nerf(info, device=self.device, is_train=False)
        keys = ["ray_rgb", "gradient_error", "depth", "acc", "normal_vis", "diffuse_albedo", "diffuse_light", "diffuse_color",
                "specular_albedo", "specular_light", "specular_color", "specular_ref", "specular_direct_light", "metallic",
                "roughness", "occ_prob", "indirect_light", "occ_prob_gt", "radiance", "roughness_weights", "human_light"]
        outputs = {}
        trn = self.cfg["test_ray_num"]
        for ri in range(0, rn, trn):
            cur = {k: v[ri:ri + trn].contiguous() for k, v in batch.items()}
            near, far = self.near_far_from_sphere(cur["rays_o"], cur["dirs"])
            res = self.render(cur, near, far, cur["human_poses"], 0, 0, is_train=False, step=step)
            for k in keys:
                if k in res and torch.is_tensor(res[k]) and res[k].dim() >= 1 and res[k].shape[0] == cur["rays_o"].shape[0]:
                    outputs.setdefault(k, []).append(res[k].detach())
        outputs = {k: torch.cat(v, 0) for k, v in outputs.items()}
        outputs["loss_rgb"] = self.compute_rgb_loss(outputs["ray_rgb"], batch["rgbs"])
        outputs["gt_rgb"] = batch["rgbs"].reshape(h, w, 3)
        outputs["ray_rgb"] = outputs["ray_rgb"].reshape(h, w, 3)
        if "radiance" in outputs:
            outputs["loss_radiance"] = self.compute_rgb_loss(outputs["radiance"], batch["rgbs"]) * outputs["roughness_weights"]
            outputs["loss_rgb"] = outputs["loss_rgb"] * (1.0 - outputs["roughness_weights"])
            outputs["radiance"] = outputs["radiance"].reshape(h, w, 3)
        outputs["gt_mask"] = (F.avg_pool2d(self.test_imgs_info["masks"][index:index + 1], self.test_imgs_info["masks"].shape[-1] // w)[0, 0] > 0).int()[..., None]
        return outputs

    def forward(self, data):
        """shapeRenderer.py:1279-1310: a training iteration (`data['step']`) or one validation image (`data['eval']`, `data['index']`).
        Needs the dataset side (training=True).  cfg val_geometry (:1298-1302): the first validation image also carries the 512^3
        marching-cubes surface of the SDF (`vertices`, `triangles`: tensoflow_amd.mesh.extract_mesh)."""
        if not hasattr(self, "train_table"):
            raise NotImplementedError("ShapeRenderer.forward drives the dataset tables: construct with training=True on a TensoSDF "
                                      "synthetic scene, or call render() with a ray batch / nvs(pose, K, h, w)")
        step = data["step"]
        if "eval" not in data:
            self.update_occ_grid(step)
            self.color_network.envlight.build_mips()
            return self.train_step(step)
        outputs = self.test_step(data["index"], step=step)
        if data["index"] == 0 and self.cfg.get("val_geometry", False):
            from ..mesh import extract_mesh
            outputs["vertices"], outputs["triangles"] = extract_mesh(self, 512)
        return outputs
