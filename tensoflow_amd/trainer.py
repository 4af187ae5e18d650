"""Training harness for the material stage: the optimisation loop of train/trainer_inv.py
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
What is not: datasets, validation / test rendering, logging, the shape-stage grid-upsampling schedule (out of scope).
"""
import math

import torch

from . import dist as tdist

DEFAULT_CFG = {
    "lr_xyz_init": 1e-2, "lr_net_init": 1e-3, "lr_env_init": 1e-2,            # trainer_inv.py:30-32
    "lr_decay_target_ratio": 5e-2, "lr_decay_iters": -1, "total_step": 40000,   # :33-34
    "nis_start_iter_diffuse": 1000, "nis_start_iter_specular": 1000,           # fields.py:649-650
    "nis_update_interval_diffuse": 1000, "nis_update_interval_specular": 1000,  # :655-656
    "nis_loss_iter": 500,
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
        {"params": list(net.outer_light.parameters()), "lr": lr_env},
        {"params": list(net.albedo_predictor.parameters()) + list(net.metallic_predictor.parameters())
                   + list(net.roughness_predictor.parameters()) + list(net.inner_light.parameters()), "lr": lr_net},
    ]
    groups += net.flow_diffuse.get_optparam_groups(lr_xyz, lr_net)
    groups += net.flow_specular.get_optparam_groups(lr_xyz, lr_net)
    return groups


class MaterialTrainer:
    """One process per GPU; `world` > 1 adds the RCCL gradient all-reduce (mean) before every optimizer step."""

    def __init__(self, net, cfg=None, world=1):
        self.cfg = {**DEFAULT_CFG, **(cfg or {})}
        if self.cfg["lr_decay_iters"] < 0:
            self.cfg["lr_decay_iters"] = self.cfg["total_step"]
        self.net, self.world = net, world
        for fl in (net.flow_diffuse_copy, net.flow_specular_copy):       # the samplers' copies are never trained (fields.py:1054-1065)
            for p in fl.parameters():
                p.requires_grad = False
        self.optimizer = torch.optim.Adam(material_param_groups(net, self.cfg["lr_xyz_init"], self.cfg["lr_net_init"],
                                                                self.cfg["lr_env_init"]), betas=(0.9, 0.99))
        self.cur_lr_xyz, self.cur_lr_net = self.cfg["lr_xyz_init"], self.cfg["lr_net_init"]
        self.lr_factor = self.pre_lr_factor = 1.0
        self.step_count, self.best_para = 0, 0.0

    def trainable(self):
        return [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]

    def refresh_flow_copies(self, step):
        """MCShadingNetwork.update_step (fields.py:1056-1065)."""
        c = self.cfg
        done = []
        for name, start, every in (("diffuse", c["nis_start_iter_diffuse"], c["nis_update_interval_diffuse"]),
                                   ("specular", c["nis_start_iter_specular"], c["nis_update_interval_specular"])):
            if (step + 1) >= start and (step + 1 - start) % every == 0:
                src, dst = getattr(self.net, f"flow_{name}"), getattr(self.net, f"flow_{name}_copy")
                dst.load_state_dict(src.state_dict())
                for p in dst.parameters():
                    p.requires_grad = False
                done.append(name)
        return done

    def train_step(self, pts, view_dirs, normals, target_rgb):
        """One iteration of the loop at trainer_inv.py:181-252 on a batch of surface points: loss_rgb (L2 on sRGB) + loss_nis."""
        step = self.step_count
        self.net.train()
        self.optimizer.zero_grad(set_to_none=True)
        colors, outputs = self.net(pts, view_dirs, normals, None, step if step >= self.cfg["nis_loss_iter"] else None, True)
        loss_rgb = ((colors - target_rgb) ** 2).mean()
        loss = loss_rgb + outputs["loss_nis"]
        loss.backward()
        if self.world > 1:
            tdist.allreduce_gradients(self.trainable(), world=self.world)
        self.optimizer.step()
        # learning-rate bookkeeping, in the reference's order (:247-252)
        for g in self.optimizer.param_groups:
            g["lr"] *= self.lr_factor
        self.cur_lr_xyz *= self.lr_factor
        self.cur_lr_net *= self.lr_factor
        cur = cosine_lr_factor(step, self.cfg["lr_decay_iters"], self.cfg["lr_decay_target_ratio"])
        self.lr_factor = cur / self.pre_lr_factor
        self.pre_lr_factor = cur
        self.refresh_flow_copies(step)
        self.step_count += 1
        return {"loss": loss.detach(), "loss_rgb": loss_rgb.detach(), "loss_nis": outputs["loss_nis"].detach()}

    # ---- checkpoint (TrainerInv._save_model, trainer_inv.py:355-369)
    def state(self):
        return {"step": self.step_count, "best_para": self.best_para, "lr_factor": self.lr_factor,
                "pre_lr_factor": self.pre_lr_factor, "lr_xyz": self.cur_lr_xyz, "lr_net": self.cur_lr_net,
                "optimizer_state_dict": self.optimizer.state_dict(), "N_voxel_list": [],
                "network_state_dict": self.net.state_dict(), "kwargs": {k: v for k, v in self.cfg.items()}}

    def save(self, path):
        torch.save(self.state(), path)

    def load(self, ckpt, load_optimizer=True):
        ckpt = torch.load(ckpt, weights_only=False) if isinstance(ckpt, str) else ckpt
        self.net.load_state_dict(ckpt["network_state_dict"], strict=False)
        if load_optimizer and "optimizer_state_dict" in ckpt:
            self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        self.step_count, self.best_para = ckpt["step"], ckpt["best_para"]
        self.lr_factor, self.pre_lr_factor = ckpt["lr_factor"], ckpt["pre_lr_factor"]
        self.cur_lr_xyz, self.cur_lr_net = ckpt["lr_xyz"], ckpt["lr_net"]
        self.net._shader = None        # the fused evaluator repacks from the loaded parameters on next use
