"""Training harnesses for the material stage and the shape stage: the optimisation loop of train/trainer_inv.py
(:88-153 optimizer set-up, :181-305 step loop, :339-343 cosine decay, :355-369 checkpoint) around the drop-in
MCShadingNetwork, with the gradient all-reduce of the data-parallel path (tensoflow_amd/dist.py) in place of the reference's
single-GPU step.

What is mirrored:
  * Adam(betas=(0.9, 0.99)) over the reference's parameter groups and their three learning rates
    (MCShadingNetwork.get_optparam_groups, network/fields.py:1580-1595): VM lines / planes at lr_xyz, the environment map at
    lr_env, predictors + inner-light net at lr_net, the two trainable flows' grids at lr_xyz and nets at lr_net;
  * the per-step multiplicative cosine decay  lr *= f(step) / f(step-1),
    f = (cos(pi step / lr_decay_iters) + 1) / 2 * (1 - target) + target   (trainer_inv.py:247-252, :339-343);
  * the periodic refresh of the frozen flow copies the samplers draw from (MCShadingNetwork.update_step, fields.py:1050-1065);
  * the checkpoint dictionary layout of TrainerInv._save_model (step, best_para, lr_factor, pre_lr_factor, lr_xyz, lr_net,
    optimizer_state_dict, N_voxel_list, network_state_dict, kwargs) so that files move between the two code bases.
  * shape stage (ShapeTrainer): the log-spaced N_voxel_list and N_to_reso (:124-127, :349-353), the weighted loss terms of
    network/loss.py named in cfg['loss'], alpha-mask refresh at update_AlphaMask_lst (:277-278), grid upsampling at
    upsample_list followed by a fresh optimizer (:286-295), EnvLight.build_mips every step (shapeRenderer.py:1291).
What is not: datasets, validation / test rendering, logging (out of scope).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import dist as tdist

DEFAULT_CFG = {
    "lr_xyz_init": 1e-2, "lr_net_init": 1e-3, "lr_env_init": 1e-2,            # trainer_inv.py:30-32
    "lr_decay_target_ratio": 5e-2, "lr_decay_iters": -1, "total_step": 40000,   # :33-34
    "nis_start_iter_diffuse": 1000, "nis_start_iter_specular": 1000,           # fields.py:649-650
    "nis_update_interval_diffuse": 1000, "nis_update_interval_specular": 1000,  # :655-656
    "nis_loss_iter": 500,
    # material objective (configs/mat/syn/*.yaml + MaterialRenderer.default_cfg + NISLoss.default_cfg)
    "rgb_loss": "charbonier", "reg_mat": True, "reg_diffuse_light": True, "reg_diffuse_light_lambda": 0.1, "nis_loss_weight": 1e-4,
}


def cosine_lr_factor(step, lr_decay_iters, target_ratio):
    """f(step) of trainer_inv.py:339-341."""
    progress = step / lr_decay_iters
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1.0 - target_ratio) + target_ratio


def n_to_reso(n_voxels, bbox):
    """TrainerInv.N_to_reso (trainer_inv.py:349-353): grid resolution with ~n_voxels cells inside bbox."""
    bbox = torch.as_tensor(bbox, dtype=torch.float32)
    ext = bbox[1] - bbox[0]
    voxel = (ext.prod() / n_voxels).pow(1.0 / 3.0)
    return (ext / voxel).long().tolist()


def material_param_groups(net, lr_xyz, lr_net, lr_env):
    """MCShadingNetwork.get_optparam_groups (fields.py:1580-1595) for the drop-in module."""
    groups = [
        {"params": list(net.mat_line), "lr": lr_xyz},
        {"params": list(net.mat_plane), "lr": lr_xyz},
        # fields.py:1584: the cube map takes the environment-light rate, the direction-encoded net the network rate
        {"params": list(net.outer_light.parameters()), "lr": lr_env if net.cfg["outer_light_version"] == "envlight" else lr_net},
        {"params": list(net.albedo_predictor.parameters()) + list(net.metallic_predictor.parameters())
                   + list(net.roughness_predictor.parameters()) + list(net.inner_light.parameters()), "lr": lr_net},
        # (the reference's groups do not list `human_light`: with human_lights=True that net keeps its initial weights, fields.py:1580-1586)
    ]
    for name in ("flow", "flow_diffuse", "flow_specular"):          # fields.py:1589-1594: the flows this cfg holds (use_nis_all / _diffuse / _specular)
        if hasattr(net, name):
            groups += getattr(net, name).get_optparam_groups(lr_xyz, lr_net)
    return groups


def material_loss_terms(cfg, colors, outputs, target_rgb, mat_reg, step):
    """The `loss_*` terms TrainerInv sums for the material stage (configs/mat/syn/*.yaml: loss = ['nerf_render', 'mat_reg', 'nis']):
    MaterialRenderer.train_step (materialRenderer.py:555-565) + NeRFRenderLoss / MaterialRegLoss / NISLoss (network/loss.py).
    -> {name: tensor}; the caller adds up their means (trainer_inv.py:196-207)."""
    terms = {"loss_rgb": rgb_loss(cfg["rgb_loss"], colors, target_rgb)}                      # compute_rgb_loss, default charbonier
    if cfg["reg_mat"] and mat_reg is not None:
        terms["loss_mat_reg"] = mat_reg                                                       # material_regularization (fields.py:1547-1578)
    if cfg["reg_diffuse_light"] and "diffuse_light" in outputs:
        dl = outputs["diffuse_light"]
        terms["loss_diffuse_light"] = torch.sum(torch.abs(dl - dl.mean(-1, keepdim=True)), -1) * cfg["reg_diffuse_light_lambda"]
    if "loss_nis" in outputs:
        terms["loss_nis"] = outputs["loss_nis"].reshape(1) * cfg["nis_loss_weight"]          # NISLoss: weight 1e-4
    return terms


class MaterialTrainer:
    """One process per GPU; `world` > 1 adds the RCCL gradient averaging before every optimizer step.
    `net`: the drop-in MCShadingNetwork, or a MaterialRenderer (its shader_network is trained; its sdf_network / deviation_net ride
    along in the checkpoint, as in the reference, whose material checkpoints are MaterialRenderer.state_dict())."""

    def __init__(self, net, cfg=None, world=1):
        self.cfg = {**DEFAULT_CFG, **(cfg or {})}
        if self.cfg["lr_decay_iters"] < 0:
            self.cfg["lr_decay_iters"] = self.cfg["total_step"]
        self.renderer = net if hasattr(net, "shader_network") else None
        self.net = net.shader_network if self.renderer is not None else net
        net = self.net
        self.world = world
        for k in ("nis_start_iter_diffuse", "nis_start_iter_specular", "nis_update_interval_diffuse", "nis_update_interval_specular"):
            if cfg and k in cfg:                                         # schedule keys live in the shader cfg (fields.py:649-656)
                net.cfg[k] = cfg[k]
        if cfg and "nis_loss_iter" in cfg:
            net.cfg["nis_loss_iter_diffuse"] = net.cfg["nis_loss_iter_specular"] = cfg["nis_loss_iter"]
        for fl in net.flow_copies():                                     # the samplers' copies are never trained (fields.py:1054-1065)
            for p in fl.parameters():
                p.requires_grad = False
        self.optimizer = torch.optim.Adam(material_param_groups(net, self.cfg["lr_xyz_init"], self.cfg["lr_net_init"],
                                                                self.cfg["lr_env_init"]), betas=(0.9, 0.99))
        self.cur_lr_xyz, self.cur_lr_net = self.cfg["lr_xyz_init"], self.cfg["lr_net_init"]
        self.lr_factor = self.pre_lr_factor = 1.0
        self.step_count, self.best_para = 0, 0.0

    def trainable(self, step=None):
        """Parameters that receive a gradient at `step` -- a decision every rank makes identically from the step alone, so the
        replicas' Adam states stay in lockstep with a single-process run: the two trainable flows are only reached through the
        NIS losses, which start at nis_loss_iter (fields.py:1257,1296); before that their .grad stays None and Adam skips them."""
        ps = [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]
        if step is not None:
            # the thresholds are the NETWORK's (a shader_cfg may set them per lobe; the trainer cfg key only seeds them above): a flow
            # whose loss has not started is left out, one whose loss has started is exchanged -- per flow, not for both at once
            skip = set()
            for name in ("diffuse", "specular"):
                if step < self.net.cfg[f"nis_loss_iter_{name}"]:
                    skip |= {id(p) for p in getattr(self.net, f"flow_{name}").parameters()}
            ps = [p for p in ps if id(p) not in skip]
        return ps

    def _exchange(self):
        """world > 1: the hooked gradient exchange over every parameter the optimizer holds (dist.GradientExchange: persistent flat
        buckets, collectives launched during backward); rebuilt when the optimizer's parameter set changes."""
        force = getattr(self, "force_exchange", False)        # tests: run the exchange at one rank (dist.GradientExchange force_collectives)
        if self.world <= 1 and not force:
            return None
        ps = self.trainable()
        ex = getattr(self, "_ex", None)
        if ex is None or not ex.same_params(ps):
            if ex is not None:
                ex.remove()
            ex = self._ex = tdist.GradientExchange(ps, self.world, force_collectives=force)
        return ex

    def refresh_flow_copies(self, step):
        """MCShadingNetwork.update_step (fields.py:1056-1065)."""
        return self.net.update_step(step)

    def train_step(self, pts, view_dirs, normals, target_rgb, human_poses=None):
        """One iteration of the loop at trainer_inv.py:181-252 on a batch of surface points, with the reference's objective:
        sum of the means of loss_rgb (charbonier), loss_mat_reg, loss_diffuse_light and 1e-4 loss_nis.
        human_poses [pn,3,4]: the capturer's pose per point (shader_cfg.human_lights, configs/mat/custom)."""
        step = self.step_count
        self.net.train()
        ex = self._exchange()
        if ex is not None:
            ex.stats = getattr(self, "comm_stats", None)                  # (before backward: the buckets are sent from inside it)
            ex.zero_grad(expected=self.trainable(step))                   # gradients accumulate into the persistent exchange buckets
        else:
            self.optimizer.zero_grad(set_to_none=True)
        self.refresh_flow_copies(step)                                    # MaterialRenderer.train_step calls update_step first (:549)
        colors, outputs = self.net(pts, view_dirs, normals, human_poses, step, True)
        mat_reg = None
        if self.cfg["reg_mat"]:
            mat_reg = self.net.material_regularization(pts, normals, outputs["metallic"], outputs["roughness"], outputs["albedo"], step)
        terms = material_loss_terms(self.cfg, colors, outputs, target_rgb, mat_reg, step)
        loss = sum(v.mean() for v in terms.values())
        loss.backward()                                                   # every bucket's collective is queued as its last gradient lands
        if ex is not None:
            ex.finish(expected=self.trainable(step))
        self.optimizer.step()
        # learning-rate bookkeeping, in the reference's order (:247-252)
        for g in self.optimizer.param_groups:
            g["lr"] *= self.lr_factor
        self.cur_lr_xyz *= self.lr_factor
        self.cur_lr_net *= self.lr_factor
        cur = cosine_lr_factor(step, self.cfg["lr_decay_iters"], self.cfg["lr_decay_target_ratio"])
        self.lr_factor = cur / self.pre_lr_factor
        self.pre_lr_factor = cur
        self.step_count += 1
        return {"loss": loss.detach(), **{k: v.detach().mean() for k, v in terms.items()}}

    # ---- checkpoint (TrainerInv._save_model, trainer_inv.py:355-369; network part = MaterialRenderer.ckpt_to_save, :230-232)
    def network_state_dict(self):
        """MaterialRenderer.state_dict() layout: 'shader_network.<key>' (+ 'sdf_network.*' / 'deviation_net.*' when a renderer is wrapped)."""
        if self.renderer is not None:
            return self.renderer.state_dict()
        return {"shader_network." + k: v for k, v in self.net.state_dict().items()}

    def state(self):
        return {"step": self.step_count, "best_para": self.best_para, "lr_factor": self.lr_factor,
                "pre_lr_factor": self.pre_lr_factor, "lr_xyz": self.cur_lr_xyz, "lr_net": self.cur_lr_net,
                "optimizer_state_dict": self.optimizer.state_dict(), "N_voxel_list": [],
                "network_state_dict": self.network_state_dict(), "trainer_cfg": {k: v for k, v in self.cfg.items()}}

    def save(self, path):
        torch.save(self.state(), path)

    def load(self, ckpt, load_optimizer=True):
        """Accepts a reference material checkpoint (keys 'shader_network.*', 'sdf_network.*', 'deviation_net.*') or a round-1 file of
        this trainer (bare MCShadingNetwork keys).  Raises if NOTHING in the file matches a parameter (strict=False would hide it)."""
        ckpt = torch.load(ckpt, weights_only=False) if isinstance(ckpt, str) else ckpt
        sd = ckpt["network_state_dict"]
        own = set(self.net.state_dict().keys())
        mapped = {}
        for k, v in sd.items():
            kk = k[len("shader_network."):] if k.startswith("shader_network.") else k
            if kk in own:
                mapped[kk] = v
        if not mapped:
            raise RuntimeError("MaterialTrainer.load: no key of the checkpoint's network_state_dict matches the shading network "
                               f"(first keys: {list(sd)[:3]})")
        self.net.load_state_dict(mapped, strict=False)
        if self.renderer is not None:
            rest = {k: v for k, v in sd.items() if k.startswith(("sdf_network.", "deviation_net."))}
            if rest:
                self.renderer.load_state_dict(rest, strict=False)
        self.loaded_keys = len(mapped)
        if load_optimizer and "optimizer_state_dict" in ckpt:
            self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        self.step_count, self.best_para = ckpt["step"], ckpt["best_para"]
        self.lr_factor, self.pre_lr_factor = ckpt["lr_factor"], ckpt["pre_lr_factor"]
        self.cur_lr_xyz, self.cur_lr_net = ckpt["lr_xyz"], ckpt["lr_net"]
        self.net._shader = None        # the fused evaluator repacks from the loaded parameters on next use


# ------------------------------------------------------------------------------------------------ shape stage
SHAPE_CFG = {
    **{k: DEFAULT_CFG[k] for k in ("lr_xyz_init", "lr_net_init", "lr_env_init", "lr_decay_target_ratio", "lr_decay_iters")},
    "total_step": 200000, "N_voxel_init": 128 ** 3, "N_voxel_final": 400 ** 3,                    # trainer_inv.py:35,47-48
    "aabb": [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]],
    "upsample_list": None, "update_AlphaMask_lst": None, "hessian_update_list": None, "sparse_update_list": None,
    "loss": ["nerf_render", "eikonal", "std", "init_sdf_reg", "occ", "Sparse", "TV", "mask", "Gaussian"],   # configs/shape/syn/*.yaml
    "rgb_loss": "charbonier", "apply_mask_loss": True,
    # weights: defaults of the classes in network/loss.py
    "eikonal_weight": 0.1, "eikonal_weight_anneal_begin": 0, "eikonal_weight_anneal_end": 0,
    "TV_weight_sdf": 0.1, "sparse_weight": 0.02, "sparse_ratio": [1.0, 1.0], "hessian_weight": 5e-4, "hessian_ratio": [1.0, 1.0],
    "gaussian_weight": 5e-4, "mask_loss_weight": 0.01, "apply_std_loss": False, "std_loss_weight": 0.05,
}


def voxel_schedule(n_init, n_final, upsample_list):
    """trainer_inv.py:124-126: voxel budgets log-spaced from N_voxel_init to N_voxel_final, one per grid (first + one per upsample)."""
    n = len(upsample_list) + 1 if upsample_list is not None else 1
    return np.round(np.exp(np.linspace(np.log(n_init), np.log(n_final), n))).astype(np.int32).tolist()


def rgb_loss(kind, pr, gt):
    """ShapeRenderer.compute_rgb_loss (shapeRenderer.py:796-808) -> [rn]."""
    if kind == "l2":
        return ((pr - gt) ** 2).sum(-1)
    if kind == "l1":
        return (pr - gt).abs().sum(-1)
    if kind == "smooth_l1":
        return F.smooth_l1_loss(pr, gt, reduction="none", beta=0.25).sum(-1)
    if kind == "charbonier":
        return torch.sqrt(((gt - pr) ** 2).sum(-1) + 0.001)
    raise NotImplementedError(kind)


def _staged_ratio(step, update_list, ratios):
    """Sparse_Loss / Hessian_Loss (loss.py:95-103, :115-123): the ratio of the last stage whose start step has passed (stage 0 is
    never selected by the reference's loop; before any stage the ratio stays 1)."""
    if update_list:
        for i in range(len(update_list) - 1, 0, -1):
            if step >= update_list[i]:
                return ratios[i]
    return 1.0


def shape_loss_terms(cfg, out, batch, step):
    """The `loss_*` terms TrainerInv sums (trainer_inv.py:196-207) for the shape stage: render outputs -> {name: weighted tensor}.
    Follows network/loss.py class by class; every term is reduced with mean() by the caller, like the reference."""
    names, terms = set(cfg["loss"]), {}
    if "nerf_render" in names:                                                          # train_step :787-791 + NeRFRenderLoss
        terms["loss_rgb"] = rgb_loss(cfg["rgb_loss"], out["ray_rgb"], batch["rgbs"])
        if "radiance" in out:
            terms["loss_radiance"] = rgb_loss(cfg["rgb_loss"], out["radiance"], batch["rgbs"]) * out["roughness_weights"]
            terms["loss_rgb"] = terms["loss_rgb"] * (1.0 - out["roughness_weights"])
    if "eikonal" in names:                                                              # EikonalLoss
        b, e = cfg["eikonal_weight_anneal_begin"], cfg["eikonal_weight_anneal_end"]
        w = 0.0 if step < b else (cfg["eikonal_weight"] * (step - b) / (e - b) if step < e else cfg["eikonal_weight"])
        terms["loss_eikonal"] = out["gradient_error"] * w
    if "std" in names and cfg["apply_std_loss"]:                                        # StdRecorder
        terms["loss_std"] = out["std"] * cfg["std_loss_weight"]
    if "init_sdf_reg" in names and "sdf_vals" in out and step < 1000 and out["sdf_pts"].dim() == 2:   # InitSDFRegLoss
        norm, sdf = out["sdf_pts"].norm(dim=-1), out["sdf_vals"].reshape(-1)
        anneal = (math.cos(step / 1000 * math.pi) + 1) / 2
        zero = torch.zeros(1, device=sdf.device)
        small, large = norm < 0.1, norm > 1.05
        if bool(small.any()):
            sl = torch.clamp(sdf[small] - (norm[small] - 0.1), min=0.0).mean()
            sl = sl / ((sl > 1e-5).float() + 1e-3)
        else:
            sl = zero
        if bool(large.any()):
            ll = torch.clamp((norm[large] - 1.05) - sdf[large], min=0.0)
            ll = ll.sum() / ((ll > 1e-5).sum() + 1e-3)
        else:
            ll = zero
        terms["loss_sdf_large"], terms["loss_sdf_small"] = ll * anneal, sl * anneal
    if "occ" in names and "loss_occ" in out:                                            # OccLoss
        terms["loss_occ"] = out["loss_occ"].mean().reshape(1)
    if "Sparse" in names and "loss_sparse" in out:
        terms["loss_sparse"] = out["loss_sparse"] * cfg["sparse_weight"] * _staged_ratio(step, cfg["sparse_update_list"], cfg["sparse_ratio"])
    if "Hessian" in names and "loss_hessian" in out:
        terms["loss_hessian"] = out["loss_hessian"] * cfg["hessian_weight"] * _staged_ratio(step, cfg["hessian_update_list"], cfg["hessian_ratio"])
    if "TV" in names and "loss_tv_sdf" in out:
        terms["loss_tv_sdf"] = out["loss_tv_sdf"].mean().reshape(1) * cfg["TV_weight_sdf"]
    if "Gaussian" in names and "loss_gaussian" in out:
        terms["loss_gaussian"] = out["loss_gaussian"] * cfg["gaussian_weight"]
    if "mask" in names and cfg["apply_mask_loss"] and "masks" in batch:                 # train_step :792-793 + MaskLoss
        bce = F.binary_cross_entropy(out["acc"].clip(1e-3, 1.0 - 1e-3), (batch["masks"] > 0.5).float().reshape(out["acc"].shape))
        terms["loss_mask"] = bce.reshape(1) * cfg["mask_loss_weight"]
    return terms


class ShapeTrainer:
    """Shape-stage loop around the drop-in ShapeRenderer; `world` > 1 adds the gradient all-reduce before the optimizer step.
    `make_renderer(gridSize)` builds the network for the first grid of the voxel schedule (trainer_inv.py:127-129)."""

    def __init__(self, make_renderer, cfg=None, world=1):
        self.cfg = {**SHAPE_CFG, **(cfg or {})}
        if self.cfg["lr_decay_iters"] < 0:
            self.cfg["lr_decay_iters"] = self.cfg["total_step"]
        self.world = world
        self.N_voxel_list = voxel_schedule(self.cfg["N_voxel_init"], self.cfg["N_voxel_final"], self.cfg["upsample_list"])
        self.net = make_renderer(n_to_reso(self.N_voxel_list.pop(0), self.cfg["aabb"]))
        self._new_optimizer()
        self.cur_lr_xyz, self.cur_lr_net = self.cfg["lr_xyz_init"], self.cfg["lr_net_init"]
        self.lr_factor = self.pre_lr_factor = 1.0
        self.step_count, self.best_para = 0, 0.0

    def _new_optimizer(self):
        c = self.cfg
        self.optimizer = torch.optim.Adam(self.net.get_train_opt_params(c["lr_xyz_init"], c["lr_net_init"], c["lr_env_init"]), betas=(0.9, 0.99))

    def trainable(self):
        return [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]

    def train_step(self, batch):
        """batch: rays_o, rays_d, dirs [rn,3], radiis, rays_cos [rn,1], rgbs [rn,3] (, masks [rn] or [rn,1])."""
        step, net, c = self.step_count, self.net, self.cfg
        net.train()
        ex = MaterialTrainer._exchange(self)                                           # (same rule: world > 1, rebuilt after a grid upsample)
        if ex is not None:
            ex.zero_grad(expected=self.trainable())
        else:
            self.optimizer.zero_grad(set_to_none=True)
        net.color_network.envlight.build_mips()
        near, far = net.near_far_from_sphere(batch["rays_o"], batch["dirs"])
        out = net.render(batch, near, far, batch.get("human_poses"), -1, net.get_anneal_val(step), is_train=True, step=step)
        terms = shape_loss_terms(c, out, batch, step)
        loss = sum(v.mean() for v in terms.values())
        loss.backward()
        if ex is not None:
            ex.finish(expected=self.trainable())
        self.optimizer.step()
        for g in self.optimizer.param_groups:                                           # :247-252
            g["lr"] *= self.lr_factor
        self.cur_lr_xyz *= self.lr_factor
        self.cur_lr_net *= self.lr_factor
        cur = cosine_lr_factor(step, c["lr_decay_iters"], c["lr_decay_target_ratio"])
        self.lr_factor, self.pre_lr_factor = cur / self.pre_lr_factor, cur
        events = []
        if net.occ_grid is None and c["update_AlphaMask_lst"] is not None and step in c["update_AlphaMask_lst"]:      # :277-278
            net.updateAlphaMask()
            events.append("alpha_mask")
        if c["upsample_list"] is not None and step in c["upsample_list"]:               # :286-295
            net.upsample_sdf_grid(n_to_reso(self.N_voxel_list.pop(0), c["aabb"]))
            self._new_optimizer()                                                       # fresh Adam moments at the initial rates
            self.cur_lr_xyz, self.cur_lr_net = c["lr_xyz_init"] * 0.5, c["lr_net_init"]
            events.append("upsample")
        self.step_count += 1
        with torch.no_grad():
            psnr = 20 * torch.log10(1.0 / torch.sqrt(F.mse_loss(out["ray_rgb"], batch["rgbs"])))
        return {"loss": loss.detach(), "psnr": psnr, "sample_num": out["sample_num"], "events": events,
                **{k: v.detach().mean() for k, v in terms.items()}}

    # ---- checkpoint (TrainerInv._save_model :355-369: trainer fields + ShapeRenderer.ckpt_to_save)
    def state(self):
        st = {"step": self.step_count, "best_para": self.best_para, "lr_factor": self.lr_factor, "pre_lr_factor": self.pre_lr_factor,
              "lr_xyz": self.cur_lr_xyz, "lr_net": self.cur_lr_net, "optimizer_state_dict": self.optimizer.state_dict(),
              "N_voxel_list": list(self.N_voxel_list)}
        st.update(self.net.ckpt_to_save())
        return st

    def save(self, path):
        torch.save(self.state(), path)

    @classmethod
    def resume(cls, ckpt, make_renderer, cfg=None, world=1):
        """trainer_inv.py:97-113: the network is rebuilt at the checkpoint's gridSize (kwargs), the optimizer starts fresh (the
        reference leaves its load commented out), the schedule state continues."""
        ckpt = torch.load(ckpt, weights_only=False) if isinstance(ckpt, str) else ckpt
        self = cls.__new__(cls)
        self.cfg = {**SHAPE_CFG, **(cfg or {})}
        if self.cfg["lr_decay_iters"] < 0:
            self.cfg["lr_decay_iters"] = self.cfg["total_step"]
        self.world = world
        self.N_voxel_list = list(ckpt["N_voxel_list"])
        self.net = make_renderer(ckpt["kwargs"]["gridSize"], ckpt["kwargs"]["max_levels"])
        self.net.load_ckpt(ckpt)
        self._new_optimizer()
        self.cur_lr_xyz, self.cur_lr_net = ckpt["lr_xyz"], ckpt["lr_net"]
        self.lr_factor, self.pre_lr_factor = ckpt["lr_factor"], ckpt["pre_lr_factor"]
        self.step_count, self.best_para = ckpt["step"], ckpt["best_para"]
        return self
