"""MaterialRenderer (reference: network/materialRenderer.py:98-830) on the HIP kernels -- the drop-in module of the material stage.

Same constructor (`MaterialRenderer(cfg, training, nvs)`), sub-module names (`shader_network`, `sdf_network`, `deviation_net`)
and call surface for everything on the hot path:

    trace / trace_in_batch / trace_sdf_with_mesh / near_far_from_sphere / shade / compute_rgb_loss /
    compute_diffuse_light_regularization / get_train_opt_params / ckpt_to_save / load_ckpt / init_sdf / nvs / predict_materials

`cfg['mesh']` is the geometry the reference reads with open3d and hands to raytracing.RayTracer (:147-149): here a
(vertices [V,3] float, triangles [F,3] int) pair, the path of a triangle .ply (tensoflow_amd.mesh.read_ply; the file
extract_mesh.py / tensoflow_amd.mesh.extract_mesh writes) or of an .npz holding `vertices` / `triangles`.  The dataset side (`_init_dataset`, `train_step`, `test_step`: image tables,
ray shuffling) is outside the hot path: construct with nvs=True and feed surface points to `shade`.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, surface
from ..march import SdfField, near_far_from_sphere
from ..trainer import material_param_groups
from .fields import MCShadingNetwork, SingleVarianceNetwork, TensoSDF


class MaterialRenderer(nn.Module):
    default_cfg = {"train_ray_num": 2048, "test_ray_num": 8192, "rgb_loss": "charbonier", "mesh": None, "shader_cfg": {},
                   "reg_mat": True, "reg_diffuse_light": True, "reg_diffuse_light_lambda": 0.1, "nerfDataType": False,
                   "device": "cuda", "direct_sn0": 128, "direct_sn1": 9, "sec_sn0": 64, "sec_sn1": 6, "geo_model_path": "",
                   "std_act": "exp", "inv_s_init": 0.3, "downsample_ratio": 1,
                   "aabb": [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], "gridSize": [512, 512, 512]}

    def __init__(self, cfg, training=True, nvs=False):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        with_data = training or not nvs
        if with_data and not (self.cfg["nerfDataType"] and str(self.cfg.get("database_name", "")).split("/")[0] in ("tensoSDF", "tensoIR", "nerf", "orb")):
            raise NotImplementedError("the dataset side of MaterialRenderer (_init_dataset / train_step) reads the Blender-convention "
                                      "layouts (database_name 'tensoSDF/', 'tensoIR/', 'nerf/' or 'orb/<scene>', nerfDataType=True): for anything else "
                                      "construct with training=False, nvs=True and pass surface points to shade()")
        self.device = self.cfg["device"]
        self._init_geometry()
        # without a geometry checkpoint the reference has no aabb either; the cfg values let the module stand alone
        self.aabb = torch.tensor(self.cfg["aabb"], dtype=torch.float32)
        self.gridSize = torch.tensor(self.cfg["gridSize"])
        self._set_extent()
        self.sdf_network = None
        geo = self.cfg["geo_model_path"]
        if isinstance(geo, dict):
            self.init_sdf(geo)
        elif geo and os.path.exists(geo):
            self.init_sdf(torch.load(geo, weights_only=False))
        self._init_shader()
        if with_data:
            self._init_dataset()

    # ------------------------------------------------------------------------------ dataset side (TensoSDF synthetic scenes)
    def _init_dataset(self):
        """materialRenderer.py:345-382 + filtering_train_rays (:384-417): database, split, per-pixel ray table, every training ray
        traced against the mesh and refined on the SDF on the device (chunks of 512^2 rays, no CPU round trip per chunk); rays that
        miss are dropped; the table of surface points is shuffled.  cfg['rank'] / cfg['world']: this process's stride of a batch."""
        from ..dataset import RayTable, construct_ray_batch_nerf_material, parse_database_name
        self.database = parse_database_name(self.cfg["database_name"], self.cfg["dataset_dir"], white_bg=self.cfg.get("isBGWhite", True))
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
        self.train_num, self.test_num = len(self.train_ids), len(self.test_ids)
        # the material stage's OWN ray table (materialRenderer.py:452-480): unit directions, no half-pixel offset -- the same
        # convention nvs() / test_step render with, so depth, the +-4 unit refinement window and the NeuS cosine are in world units
        batch, n_rays, _, _ = construct_ray_batch_nerf_material(self.database.imgs_info(self.train_ids), fixed_camera=self.cfg.get("fixed_camera", False))
        self.train_batch = self.filtering_train_rays(batch)
        self.tbn = self.train_batch["rays_o"].shape[0]
        self.ray_mask_ratio = self.tbn / max(n_rays, 1)
        self.train_table = RayTable(self.train_batch, rank=self.cfg.get("rank", 0), world=self.cfg.get("world", 1),
                                    seed=self.cfg.get("random_seed", 6033), device=self.device)

    @torch.no_grad()
    def filtering_train_rays(self, batch, chunk=512 ** 2):
        out = {}
        n = batch["rays_o"].shape[0]
        for i in range(0, n, chunk):
            cur = {k: v[i:i + chunk].to(self.device) for k, v in batch.items()}
            inters, normals, depth, hit = self.trace_sdf_with_mesh(cur["rays_o"].contiguous(), cur["rays_d"].contiguous())
            hit = hit.reshape(-1)
            cur = {k: v[hit] for k, v in cur.items()}
            cur.update({"inters": inters.reshape(-1, 3)[hit], "normals": normals.reshape(-1, 3)[hit], "depth": depth.reshape(-1, 1)[hit]})
            for k, v in cur.items():
                out.setdefault(k, []).append(v.cpu())
        return {k: torch.cat(v, 0) for k, v in out.items()}

    def train_step(self, step):
        """materialRenderer.py:539-566."""
        b = self.train_table.next_batch(self.cfg["train_ray_num"])
        pts, view_dirs, normals, rgb_gt = b["inters"], -b["rays_d"], b["normals"], b["rgb"]
        self.shader_network.update_step(step)
        out = self.shade(pts, view_dirs, normals, b["human_poses"], True, step)
        out["rgb_gt"] = rgb_gt
        out["loss_rgb"] = self.compute_rgb_loss(out["rgb_pr"], rgb_gt)
        out["psnr"] = 20 * torch.log10(1.0 / torch.sqrt(F.mse_loss(out["rgb_pr"], rgb_gt)))
        if self.cfg["reg_mat"]:
            out["loss_mat_reg"] = self.shader_network.material_regularization(pts, normals, out["metallic"], out["roughness"], out["albedo"], step)
        if self.cfg["reg_diffuse_light"]:
            out["loss_diffuse_light"] = self.compute_diffuse_light_regularization(out["diffuse_light"])
        return out

    @torch.no_grad()
    def test_step(self, index):
        """materialRenderer.py:568-640, reduced to what the validation metrics read: the rendered and the ground-truth image."""
        i = self.test_ids[index]
        h, w = self.database.get_image(i).shape[:2]       # as the reference (materialRenderer.py:583): TensoIRDatabase keeps H / W as floats
        img = self.nvs(self.database.get_pose(i)[:3], self.database.get_K(i), h, w)
        return {"rgb_pr": torch.from_numpy(img["color"]), "rgb_gt": torch.from_numpy(self.database.get_image(i).astype(np.float32) / 255.0),
                "gt_mask": torch.from_numpy(self.database.get_mask(i) > 0)[..., None], **{k: torch.from_numpy(v) for k, v in img.items() if k != "color"}}

    def _set_extent(self):
        self.center = self.aabb.mean(0).float().view(1, 1, 3)
        self.radius = (self.aabb[1] - self.center).mean().float()
        self.unit_size = torch.mean((self.aabb[1] - self.aabb[0]) / (self.gridSize - 1), dim=-1)

    def _init_geometry(self):
        mesh = self.cfg["mesh"]
        if isinstance(mesh, str):
            if mesh.endswith(".ply"):                    # what extract_mesh.py writes and the reference reads through open3d (:148)
                from ..mesh import read_ply
                mesh = read_ply(mesh)
            elif mesh.endswith(".npz"):
                z = np.load(mesh)
                mesh = (z["vertices"], z["triangles"])
            else:
                raise NotImplementedError("mesh files: .ply (tensoflow_amd.mesh.read_ply) or .npz with `vertices` / `triangles`")
        if mesh is None:
            raise ValueError("cfg['mesh'] = (vertices [V,3], triangles [F,3]) is required")
        v, f = mesh
        self.mesh = (np.ascontiguousarray(np.asarray(v, np.float32)), np.ascontiguousarray(np.asarray(f, np.int32)))
        self.ray_tracer = ops.Bvh(self.mesh[0], self.mesh[1], self.device)

    def init_sdf(self, ckpt):
        """materialRenderer.py:151-179: frozen TensoSDF + variance from a shape-stage checkpoint (ShapeRenderer.ckpt_to_save)."""
        kw = ckpt["kwargs"]
        self.aabb = torch.as_tensor(kw["aabb"], dtype=torch.float32).cpu()
        self.gridSize = torch.tensor(kw["gridSize"])
        self._set_extent()
        self.sdf_network = TensoSDF(self.gridSize, self.aabb, device=self.device, init_n_levels=kw["max_levels"], sdf_n_comp=kw["sdf_n_comp"],
                                    sdf_dim=kw["sdf_dim"], app_dim=kw["app_dim"], sdf_multires=kw.get("sdf_multires", 0))
        self.deviation_net = SingleVarianceNetwork(self.cfg["inv_s_init"], self.cfg["std_act"]).to(self.device)
        sd = ckpt["network_state_dict"]
        own = self.sdf_network.state_dict()                       # materialRenderer.py:166-173: model dict updated by the trained one
        own.update({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("sdf")})
        self.sdf_network.load_state_dict(own)
        self.deviation_net.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("deviation")})
        for p in list(self.sdf_network.parameters()) + list(self.deviation_net.parameters()):
            p.requires_grad = False
        self.sdf_inter_fun = lambda x: self.sdf_network.sdf(x, None)

    def _init_shader(self):
        self.shader_network = MCShadingNetwork(self.cfg["shader_cfg"], self.mesh, self.aabb, float(self.unit_size))

    # ------------------------------------------------------------------------------ bookkeeping
    def get_train_opt_params(self, learning_rate_xyz, learning_rate_net, learning_rate_env):
        return material_param_groups(self.shader_network, learning_rate_xyz, learning_rate_net, learning_rate_env)

    def ckpt_to_save(self):
        return {"network_state_dict": self.state_dict()}

    def load_ckpt(self, ckpt):
        self.load_state_dict(ckpt["network_state_dict"], strict=False)
        self.shader_network._shader = None

    # ------------------------------------------------------------------------------ geometry queries
    def near_far_from_sphere(self, rays_o, rays_d):
        return near_far_from_sphere(rays_o, rays_d, float(self.radius))

    @torch.no_grad()
    def trace(self, rays_o, rays_d):
        """materialRenderer.py:253-263 -> inters [M,3], normals [M,3] (flipped, unit), depth [M,1], hit_mask [M,1] bool."""
        inters, normals, depth, hit = self.ray_tracer.trace(rays_o.contiguous(), rays_d.contiguous())
        return inters, normals, depth[:, None], hit[:, None]

    def trace_in_batch(self, rays_o, rays_d, batch_size=512 ** 2, cpu=False):
        outs = [self.trace(rays_o[i:i + batch_size], rays_d[i:i + batch_size]) for i in range(0, rays_o.shape[0], batch_size)]
        cat = [torch.cat(c, 0) for c in zip(*outs)]
        return tuple(c.cpu() for c in cat) if cpu else tuple(cat)

    def _sdf_field(self):
        net = self.sdf_network
        if net is None:
            raise RuntimeError("trace_sdf_with_mesh needs the shape-stage checkpoint (cfg['geo_model_path'])")
        f = SdfField.__new__(SdfField)
        f.planes, f.lines = [p.detach() for p in net.sdf_plane], [p.detach() for p in net.sdf_line]
        f.W = [w.detach() for w in net._w()]
        f.aabb, f.aabb_dev = self.aabb.cpu(), self.aabb.to(self.device)
        f.grid_size = self.gridSize.float().cpu()
        f.units = [float(u) for u in net.units]
        f.n_levels, f.device, f.packed = net.n_levels, self.device, net._field()
        return f

    @torch.no_grad()
    def trace_sdf_with_mesh(self, rays_o, rays_d, sn0=32, sn1=9):
        """materialRenderer.py:316-343: mesh hit refined on the SDF (sn0 uniform + sn1 importance evaluations, FD normal)."""
        return surface.trace_sdf_with_mesh(self.ray_tracer, self._sdf_field(), rays_o.contiguous(), rays_d.contiguous(),
                                           float(self.deviation_net.inv_s()), float(self.unit_size), sn0=sn0, sn1=sn1)

    def trace_sdf_in_batch(self, rays_o, rays_d, batch_size=10240 * 5, cpu=False):
        outs = [self.trace_sdf_with_mesh(rays_o[i:i + batch_size], rays_d[i:i + batch_size]) for i in range(0, rays_o.shape[0], batch_size)]
        cat = [torch.cat(c, 0) for c in zip(*outs)]
        return tuple(c.cpu() for c in cat) if cpu else tuple(cat)

    # ------------------------------------------------------------------------------ shading
    def shade(self, pts, view_dirs, normals, human_poses=None, is_train=False, step=None):
        """materialRenderer.py:518-521 -> outputs dict with 'rgb_pr'."""
        rgb_pr, outputs = self.shader_network(pts, view_dirs, normals, human_poses, step, is_train)
        outputs["rgb_pr"] = rgb_pr
        return outputs

    def compute_rgb_loss(self, rgb_pr, rgb_gt):
        if self.cfg["rgb_loss"] == "l1":
            return torch.sum(F.l1_loss(rgb_pr, rgb_gt, reduction="none"), -1)
        if self.cfg["rgb_loss"] == "charbonier":
            return torch.sqrt(torch.sum((rgb_gt - rgb_pr) ** 2, dim=-1) + 0.001)
        raise NotImplementedError

    def compute_diffuse_light_regularization(self, diffuse_lights):
        return torch.sum(torch.abs(diffuse_lights - torch.mean(diffuse_lights, dim=-1, keepdim=True)), dim=-1) * self.cfg["reg_diffuse_light_lambda"]

    NVS_KEYS = {"color": 3, "normal": 3, "spec_light": 3, "diff_light": 3, "indirect_light": 3, "spec_color": 3, "diff_color": 3, "albedo": 3,
                "roughness": 1, "metallic": 1, "occ_trace": 1, "variance_diffuse_vis": 1, "variance_specular_vis": 1,
                "variance_diffuse_vis_nis": 1, "variance_specular_vis_nis": 1}

    @staticmethod
    def assemble_frame(local, hit_local, h, w, rank, world):
        """The frame from every rank's rows: `local` {key: [n_local, C]} and `hit_local` [n_local] bool are rows
        dist.shard_range(h * w, rank, world) of the per-pixel maps; one all-gather (dist.gather_maps), then the ONE thing the reference's
        512-ray chunk loop decides on the whole frame: a pixel that misses gets the normal (0,0,1) inside `if sum(hit) > 0`
        (materialRenderer.py:713,725), i.e. only in 512-ray chunks -- counted from pixel 0 of the FRAME, whatever the tiling -- that
        contain a hit.  -> {key: [h,w,C] numpy}."""
        from .. import dist as tdist
        rn = h * w
        full = tdist.gather_maps(dict(local, _hit=hit_local), rn, rank, world)
        hit_all = full.pop("_hit")
        pad = (-rn) % 512
        blk = torch.cat([hit_all, hit_all.new_zeros(pad)]).view(-1, 512).any(1).repeat_interleave(512)[:rn]
        nz = full["normal"][:, 2]
        full["normal"] = torch.cat([full["normal"][:, :2], torch.where(~hit_all & blk, torch.ones_like(nz), nz)[:, None]], 1)
        return {k: v.reshape(h, w, -1).cpu().numpy() for k, v in full.items()}

    @torch.no_grad()
    def nvs(self, pose, K, h, w, chunk=65536, rank=None, world=None):
        """materialRenderer.py:641-752 (nerfDataType rays, :647-672): primary rays -> BVH -> SDF refinement (32 + 9 evaluations) ->
        MCShadingNetwork.forward(step=None) on the pixels that see the object -> dict of 15 [h,w,C] numpy maps under the reference's
        keys (:707): colour (white background, :743), normal, specular / diffuse / indirect light, specular / diffuse colour, albedo,
        roughness (sqrt of the squared prediction, :739), metallic, occ_trace (= visibility) and the four variance maps, which the
        reference leaves at zero (their assignments are commented out, :732-735).  The reference shades 512 rays per pass on one GPU
        (:705-709); `chunk` only sizes the launches here, and with world > 1 (default: torch.distributed's rank / world when a process
        group is up) those pieces are the multi-GPU unit: this rank renders rows dist.shard_range(h * w, rank, world) of the frame and
        `assemble_frame` all-gathers the maps, so every rank returns the whole frame ('tile the image, all-gather')."""
        from .. import dist as tdist
        dev = self.device
        K = torch.from_numpy(np.asarray(K, np.float32)).to(dev)
        pose = torch.from_numpy(np.asarray(pose, np.float32)).to(dev)
        h, w = int(h), int(w)
        rank, world = tdist.rank_world(rank, world)
        lo, hi = tdist.shard_range(h * w, rank, world)
        i, j = torch.meshgrid(torch.linspace(0, w - 1, w, device=dev), torch.linspace(0, h - 1, h, device=dev), indexing="ij")
        i, j = i.t(), j.t()
        dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1).reshape(-1, 3)[lo:hi]
        rays_d = F.normalize(dirs @ pose[:3, :3].t(), dim=-1).contiguous()
        rn = hi - lo
        rays_o = pose[:3, 3].expand(rn, 3).contiguous()
        human = self.shader_network.cfg["human_lights"]
        if human:
            from ..dataset import human_coordinate_poses
            hp_all = human_coordinate_poses(pose[None, :3], self.cfg.get("fixed_camera", False))[0]
        out = {k: torch.zeros(rn, c, device=dev) for k, c in self.NVS_KEYS.items()}
        out["color"][:] = 1.0
        hit_all = torch.zeros(rn, dtype=torch.bool, device=dev)
        src = {"color": "rgb_pr", "spec_light": "specular_light", "diff_light": "diffuse_light", "indirect_light": "indirect_light",
               "occ_trace": "visibility", "spec_color": "specular_color", "diff_color": "diffuse_color", "albedo": "albedo", "metallic": "metallic"}
        for s in range(0, rn, chunk):
            o, d = rays_o[s:s + chunk], rays_d[s:s + chunk]
            inters, nrm, depth, hit = self.trace_sdf_with_mesh(o, d) if self.sdf_network is not None else self.trace(o, d)
            hit_all[s:s + chunk] = hit[:, 0]
            idx = torch.nonzero(hit[:, 0], as_tuple=False)[:, 0]
            if idx.numel() == 0:
                continue
            hp = hp_all[None].expand(idx.numel(), 3, 4).contiguous() if human else None
            sh = self.shade(inters[idx].contiguous(), (-d[idx]).contiguous(), nrm[idx].contiguous(), hp, False)
            for k, sk in src.items():
                out[k][s + idx] = sh[sk]
            out["normal"][s + idx] = nrm[idx]
            out["roughness"][s + idx] = torch.sqrt(sh["roughness"])      # predictions are squared roughness (:739)
        return self.assemble_frame(out, hit_all, h, w, rank, world)

    @torch.no_grad()
    def predict_materials(self, batch_size=8192):
        """materialRenderer.py:770-782: per-vertex metallic / roughness (sqrt of the squared prediction) / albedo as numpy arrays."""
        verts = torch.from_numpy(self.mesh[0]).to(self.device)
        m_, r_, a_ = [], [], []
        for vi in range(0, verts.shape[0], batch_size):
            m, r, a = self.shader_network.predict_materials(verts[vi:vi + batch_size].contiguous())
            m_.append(m.cpu().numpy()); r_.append(torch.sqrt(torch.clamp(r, min=1e-7)).cpu().numpy()); a_.append(a.cpu().numpy())
        return {"metallic": np.concatenate(m_, 0), "roughness": np.concatenate(r_, 0), "albedo": np.concatenate(a_, 0)}

    def extract_materials(self, material_dir, albedo_ratio=None, batch_size=8192):
        """eval_mat.py:114-134: per-vertex metallic / roughness / albedo of the mesh as `metallic.npy`, `roughness.npy`, `albedo.npy`
        under `material_dir`, sRGB-encoded like the reference (its Blender script stores them as vertex colours).  `albedo_ratio`:
        the optional rescale (scalar or [3]) the reference derives from ground-truth albedo images (dataset side, not mirrored)."""
        os.makedirs(material_dir, exist_ok=True)
        mats = self.predict_materials(batch_size)
        if albedo_ratio is not None:
            mats["albedo"] = mats["albedo"] * np.asarray(albedo_ratio, np.float32)
        for name in ("metallic", "roughness", "albedo"):
            lin = mats[name]
            srgb = np.where(lin <= 0.0031308, 323 / 25 * lin, (211 * np.maximum(np.finfo(np.float32).eps, lin) ** (5 / 12) - 11) / 200)
            np.save(os.path.join(material_dir, name + ".npy"), srgb)
        return mats

    def forward(self, data):
        """materialRenderer.py:754-768: a training iteration (`data['step']`) or a validation image (`data['eval']`, `data['index']`)."""
        if not hasattr(self, "train_table"):
            raise NotImplementedError("MaterialRenderer.forward drives the dataset tables: construct with training=True on a TensoSDF "
                                      "synthetic scene, or call shade() with surface points / nvs(pose, K, h, w)")
        if "eval" not in data:
            return self.train_step(data["step"])
        return self.test_step(data["index"])
