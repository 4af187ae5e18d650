"""The data side in front of the shape stage (reader side): the TensoSDF synthetic scene layout
(`transforms_{train,val,test}.json` + RGBA PNGs) -> the per-ray training table ShapeRenderer.train_step slices.

* `TensoSDFSynDatabase`      -- dataset/database.py:479-579: frames of the listed splits; RGB composited over white (or black)
                                with the alpha channel and quantised to uint8 exactly like the reference; mask = alpha;
                                K from `camera_angle_x`; `get_pose` halves the translation (scale_factor 0.5).
                                PNG decoding: PIL (the reference uses skimage.io.imread -- same 8-bit samples).  The test split's
                                extras are read on request: `_normal.png` -> [-1,1] over a (0,0,1) background (:519-527),
                                `_diffColor.exr` -> rgb x alpha through hdr_io.read_exr (:530-535; cv2 in the reference).
* `ORBDatabase`              -- dataset/database.py:723-802 (Stanford-ORB in the Blender layout): one split ('train' or 'test'), RGB(A)
                                PNGs with the object mask in a SEPARATE file (`<split>_mask/...png`), no pose rescaling.
* `TensoIRDatabase`          -- dataset/database.py:376-477 (TensoIR synthetic: the lego / armadillo / horse configs): a directory per frame
                                with `metadata.json` + `rgba_<light>_<rotation>.png`; `NeRFSynDatabase` -- :288-374 (Blender layout, no rescaling).
* `parse_database_name`      -- :804-823 for the database types this build reads ('tensoSDF/', 'tensoIR/', 'nerf/', 'orb/<model>').
* `construct_ray_batch_nerf` -- shapeRenderer.py:471-518: pinhole rays through pixel centres in the OpenGL camera frame, cone radii
                                from neighbouring-ray distances, `rays_cos = 1 / |rays_d|`, colours, masks, per-ray pose rows.
* `RayTable`                 -- `_shuffle_train_batch` / `train_step`'s slicing (:411-415, :778-782) with the data-parallel split of
                                SURVEY.md 8(e): every rank shuffles with the same generator and takes its own stride of each batch.
"""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F


class TensoSDFSynDatabase:
    def __init__(self, root, splits=("train", "val"), white_bg=True, load_normals=False, load_diff_color=False):
        from PIL import Image
        self.root = root
        self.imgs_all, self.masks_all, self.pose_all = [], [], []
        self.normals_all, self.diffColor_all = [], []
        meta = None
        for s in splits:
            with open(os.path.join(root, f"transforms_{s}.json")) as fp:
                meta = json.load(fp)
            for fr in meta["frames"]:
                im = Image.open(os.path.join(root, fr["file_path"] + ".png"))
                if im.mode != "RGBA":
                    raise ValueError(f"{fr['file_path']}.png: RGBA expected (the alpha channel is the object mask), got {im.mode}")
                img = np.asarray(im).astype(np.float32) / 255.0
                mask = img[..., -1:]
                rgb = img[..., :3] * mask + (1 - mask) if white_bg else img[..., :3] * mask
                self.imgs_all.append((rgb * 255.0).astype(np.uint8))
                self.masks_all.append(mask)
                self.pose_all.append(np.array(fr["transform_matrix"], dtype=np.float64))
                if load_normals:
                    nrm = np.asarray(Image.open(os.path.join(root, fr["file_path"] + "_normal.png")))[..., :3] / 255
                    nrm = (nrm - 0.5) * 2.0
                    self.normals_all.append(nrm * mask + (1 - mask) * np.array([0, 0, 1]))
                if load_diff_color:
                    from .hdr_io import read_exr
                    dc, names = read_exr(os.path.join(root, fr["file_path"] + "_diffColor.exr"))
                    if names[:3] != ["R", "G", "B"] or dc.shape[-1] < 4:
                        raise ValueError(f"{fr['file_path']}_diffColor.exr: RGBA expected, got channels {names}")
                    self.diffColor_all.append(dc[..., :3] * dc[..., 3:4])
        if not self.imgs_all:
            raise ValueError(f"{root}: no frames in splits {splits}")
        self.H, self.W = self.imgs_all[0].shape[:2]
        self.focal = 0.5 * self.W / np.tan(0.5 * float(meta["camera_angle_x"]))
        self.K = np.array([[self.focal, 0, 0.5 * self.W], [0, self.focal, 0.5 * self.H], [0, 0, 1]], dtype=np.float32)
        self.scale_factor = 0.5
        self.img_ids = list(range(len(self.imgs_all)))

    def get_normal(self, i):
        return self.normals_all[int(i)]

    def get_albedo(self, i):
        return self.diffColor_all[int(i)]

    def get_image(self, i):
        return self.imgs_all[i]

    def get_mask(self, i):
        return self.masks_all[int(i)][..., -1]

    def get_K(self, i):
        return self.K

    def get_pose(self, i):
        pose = self.pose_all[i].copy()
        pose[:, 3:] *= self.scale_factor            # the whole last column, as the reference does (its bottom 1 becomes 0.5; unused)
        return pose

    def get_img_ids(self):
        return self.img_ids

    def imgs_info(self, ids=None):
        """build_imgs_info + imgs_info_to_torch for this database -> imgs [n,3,h,w] in [0,1], masks [n,1,h,w], Ks [n,3,3], poses [n,4,4]."""
        ids = self.img_ids if ids is None else ids
        imgs = np.stack([self.get_image(i) for i in ids]).astype(np.float32) / 255.0
        return {"imgs": torch.from_numpy(imgs).permute(0, 3, 1, 2).contiguous(),
                "masks": torch.from_numpy(np.stack([self.get_mask(i) for i in ids]).astype(np.float32))[:, None],
                "Ks": torch.from_numpy(np.stack([self.get_K(i) for i in ids])),
                "poses": torch.from_numpy(np.stack([self.get_pose(i) for i in ids]).astype(np.float32))}


class ORBDatabase(TensoSDFSynDatabase):
    """Stanford-ORB scenes in the Blender layout (dataset/database.py:723-802): `transforms_<split>.json`, colour PNGs, the object
    mask in `<file_path with <split> -> <split>_mask>.png` (one channel), poses used as they are (scale_factor 1)."""

    def __init__(self, root, is_test=False, white_bg=True):
        from PIL import Image
        split = "test" if is_test else "train"
        self.root = root
        self.imgs_all, self.masks_all, self.pose_all, self.normals_all, self.diffColor_all = [], [], [], [], []
        with open(os.path.join(root, f"transforms_{split}.json")) as fp:
            meta = json.load(fp)
        for fr in meta["frames"]:
            img = np.asarray(Image.open(os.path.join(root, fr["file_path"] + ".png"))).astype(np.float32) / 255.0
            m = np.asarray(Image.open(os.path.join(root, fr["file_path"].replace(split, f"{split}_mask") + ".png")))
            if m.ndim != 2:
                raise ValueError(f"{fr['file_path']}: the mask PNG must have one channel, got shape {m.shape}")
            mask = m[..., None].astype(np.float32) / 255.0
            rgb = img[..., :3] * mask + (1 - mask) if white_bg else img[..., :3] * mask
            self.imgs_all.append((rgb * 255.0).astype(np.uint8))
            self.masks_all.append(mask)
            self.pose_all.append(np.array(fr["transform_matrix"], dtype=np.float64))
        if not self.imgs_all:
            raise ValueError(f"{root}: no frames in split {split}")
        self.H, self.W = self.imgs_all[0].shape[:2]
        self.focal = 0.5 * self.W / np.tan(0.5 * float(meta["camera_angle_x"]))
        self.K = np.array([[self.focal, 0, 0.5 * self.W], [0, self.focal, 0.5 * self.H], [0, 0, 1]], dtype=np.float32)
        self.scale_factor = 1.0
        self.img_ids = list(range(len(self.imgs_all)))


class NeRFSynDatabase(TensoSDFSynDatabase):
    """NeRF-synthetic (Blender) scenes (dataset/database.py:288-374): splits train + test (test alone when is_test), RGBA PNGs, the test
    split's `_normal.png` on request, poses used as they are (scale_factor 1)."""

    def __init__(self, root, is_test=False, white_bg=True):
        super().__init__(root, splits=("test",) if is_test else ("train", "test"), white_bg=white_bg, load_normals=is_test)
        self.scale_factor = 1.0


class TensoIRDatabase(TensoSDFSynDatabase):
    """TensoIR synthetic scenes (dataset/database.py:376-477; configs/{shape,mat}/syn/{lego,armadillo,...}.yaml): one DIRECTORY per frame
    (`train_000`, `val_012`, `test_003` ...) holding `metadata.json` (`cam_transform_mat`: 16 comma-separated numbers, `cam_angle_x`,
    `imh`, `imw`), `rgba_<light>_<rotation>.png` and -- test split -- `normal.png`, `albedo.png` (both RGBA: value x alpha; the normal
    over a (0,0,1) background).  Frames sorted by directory name; translation halved (scale_factor 0.5)."""

    def __init__(self, root, is_test=False, white_bg=True, light_name="sunset", light_rotation="000"):
        from PIL import Image
        self.root = root
        self.imgs_all, self.masks_all, self.pose_all, self.normals_all, self.diffColor_all = [], [], [], [], []
        meta = None
        for s in (("test",) if is_test else ("train", "val")):
            for item in sorted(d for d in os.listdir(root) if d.startswith(s) and os.path.isdir(os.path.join(root, d))):
                with open(os.path.join(root, item, "metadata.json")) as fp:
                    meta = json.load(fp)
                im = Image.open(os.path.join(root, item, f"rgba_{light_name}_{light_rotation}.png"))
                if im.mode != "RGBA":
                    raise ValueError(f"{item}/rgba_{light_name}_{light_rotation}.png: RGBA expected, got {im.mode}")
                img = np.asarray(im).astype(np.float32) / 255.0
                mask = img[..., -1:]
                rgb = img[..., :3] * mask + (1 - mask) if white_bg else img[..., :3] * mask
                self.imgs_all.append((rgb * 255.0).astype(np.uint8))
                self.masks_all.append(mask)
                self.pose_all.append(np.array([float(v) for v in meta["cam_transform_mat"].split(",")], dtype=np.float64).reshape(4, 4))
                if is_test:
                    nim = np.asarray(Image.open(os.path.join(root, item, "normal.png")))
                    na = nim[..., [-1]] / 255
                    self.normals_all.append((nim[..., :3] / 255 - 0.5) * 2.0 * na + (1 - na) * np.array([0, 0, 1]))
                    aim = np.asarray(Image.open(os.path.join(root, item, "albedo.png")))
                    self.diffColor_all.append(aim[..., :3] / 255 * (aim[..., [-1]] / 255))
        if not self.imgs_all:
            raise ValueError(f"{root}: no frame directories for the requested splits")
        self.H, self.W = float(meta["imh"]), float(meta["imw"])                 # (floats, as the reference keeps them)
        self.focal = 0.5 * self.W / np.tan(0.5 * float(meta["cam_angle_x"]))
        self.K = np.array([[self.focal, 0, 0.5 * self.W], [0, self.focal, 0.5 * self.H], [0, 0, 1]], dtype=np.float32)
        self.scale_factor = 0.5
        self.img_ids = list(range(len(self.imgs_all)))


def parse_database_name(database_name, dataset_dir, is_test=False, white_bg=False):
    """dataset/database.py:804-823 for the layouts read here: '<type>/<model>' under `dataset_dir`."""
    if dataset_dir in (None, "None"):
        raise AssertionError("change your own dataset dir!")
    kind, model = database_name.split("/")
    root = os.path.join(dataset_dir, model)
    if kind == "tensoIR":
        return TensoIRDatabase(root, is_test=is_test, white_bg=white_bg)
    if kind == "nerf":
        return NeRFSynDatabase(root, is_test=is_test, white_bg=white_bg)
    if kind == "tensoSDF":
        return TensoSDFSynDatabase(root, splits=("test",) if is_test else ("train", "val"), white_bg=white_bg,
                                   load_normals=is_test, load_diff_color=is_test)
    if kind == "orb":
        return ORBDatabase(root, is_test=is_test, white_bg=white_bg)
    raise NotImplementedError(f"database type '{kind}' (this build reads tensoSDF/*, tensoIR/*, nerf/* and orb/*; the Glossy and "
                              f"custom-COLMAP layouts are not read)")


def construct_ray_batch_nerf(imgs_info, device="cpu", is_train=True):
    """-> (ray_batch dict of [rn, .] tensors, rn, h, w); keys: dirs, rays_d, rays_o, radiis, rays_cos, rgbs, human_poses (, masks)."""
    imn, _, h, w = imgs_info["imgs"].shape
    K, poses = imgs_info["Ks"][0], imgs_info["poses"]
    i, j = torch.meshgrid(torch.linspace(0, w - 1, w), torch.linspace(0, h - 1, h), indexing="ij")
    i, j = i.t(), j.t()
    cam = torch.stack([(i - K[0][2] + 0.5) / K[0][0], -(j - K[1][2] + 0.5) / K[1][1], -torch.ones_like(i)], -1)       # h,w,3
    dx = (cam[:, :-1] - cam[:, 1:]).norm(dim=-1, keepdim=True)
    dx = torch.cat([dx, dx[:, -2:-1]], 1)
    dy = (cam[:-1] - cam[1:]).norm(dim=-1, keepdim=True)
    dy = torch.cat([dy, dy[-2:-1]], 0)
    radiis = torch.sqrt(dx * dy / torch.pi)[None].expand(imn, h, w, 1).reshape(-1, 1)
    rn = imn * h * w
    idx = torch.arange(imn)[:, None].expand(imn, h * w).reshape(-1)
    rays_o = poses[:, :3, -1][:, None].expand(imn, h * w, 3).reshape(rn, 3)
    rays_d = (cam.reshape(1, h * w, 1, 3) * poses[:, None, :3, :3]).sum(-1).reshape(rn, 3)
    batch = {"dirs": F.normalize(rays_d, dim=-1), "rays_d": rays_d, "rays_o": rays_o, "radiis": radiis,
             "rays_cos": 1 / rays_d.norm(dim=-1, keepdim=True),
             "rgbs": imgs_info["imgs"].permute(0, 2, 3, 1).reshape(rn, 3), "human_poses": poses[idx, :3, :]}
    if is_train:
        batch["masks"] = imgs_info["masks"].reshape(rn, 1).float()
    return {k: v.float().contiguous().to(device) for k, v in batch.items()}, rn, h, w


def human_coordinate_poses(poses, fixed_camera=False):
    """MaterialRenderer.get_human_coordinate_poses (materialRenderer.py:364-380): poses [n,3,4] -> [n,3,4]."""
    pn = poses.shape[0]
    cam_cen = (-poses[:, :, :3].permute(0, 2, 1) @ poses[:, :, 3:])[..., 0]
    if not fixed_camera:
        cam_cen[..., 2] = 0
    Y = torch.zeros(pn, 3, device=poses.device)
    Y[:, 2] = -1.0
    Z = poses[:, 2, :3].clone()
    Z[:, 2] = 0
    Z = F.normalize(Z, dim=-1)
    X = torch.cross(Y, Z, dim=-1)
    R = torch.stack([X, Y, Z], 1)
    t = -R @ cam_cen[:, :, None]
    return torch.cat([R, t], -1)


def construct_ray_batch_nerf_material(imgs_info, device="cpu", fixed_camera=False):
    """MaterialRenderer._construct_ray_batch_nerf (materialRenderer.py:452-480) -- NOT the shape stage's constructor: pixel
    coordinates without the half-pixel offset, rays_d = normalize(R @ dirs), the 'human' poses of get_human_coordinate_poses.
    -> (dict rays_o, rays_d, human_poses [rn,3,4], rgb; rn, h, w)."""
    imn, _, h, w = imgs_info["imgs"].shape
    i, j = torch.meshgrid(torch.linspace(0, w - 1, w), torch.linspace(0, h - 1, h), indexing="ij")
    i, j = i.t(), j.t()
    K = imgs_info["Ks"][0]
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    rays_d = dirs[None].repeat(imn, 1, 1, 1).reshape(imn, h * w, 3)
    poses = imgs_info["poses"][:, :3, :].float()
    R, t = poses[:, :, :3], poses[:, :, 3:]
    rays_d = F.normalize((R @ rays_d.permute(0, 2, 1)).permute(0, 2, 1), dim=-1)
    rays_o = t.permute(0, 2, 1).repeat(1, h * w, 1)
    hp = human_coordinate_poses(poses, fixed_camera).unsqueeze(1).repeat(1, h * w, 1, 1)
    rgb = imgs_info["imgs"].reshape(imn, 3, h * w).permute(0, 2, 1)
    batch = {"rays_o": rays_o.reshape(-1, 3), "rays_d": rays_d.reshape(-1, 3), "human_poses": hp.reshape(-1, 3, 4), "rgb": rgb.reshape(-1, 3)}
    return {k: v.float().contiguous().to(device) for k, v in batch.items()}, imn * h * w, h, w


class RayTable:
    """The shuffled per-ray table of the training loop.  `next_batch(rn)` returns this rank's rows of the next rn-row slice (on
    `device`); the table reshuffles when fewer than 2 rn rows remain, like the reference (:782)."""

    def __init__(self, ray_batch, rank=0, world=1, seed=6033, device="cuda"):
        self.table = ray_batch
        self.tbn = next(iter(ray_batch.values())).shape[0]
        self.rank, self.world, self.device = rank, world, device
        self.gen = torch.Generator().manual_seed(seed)              # same seed on every rank: identical permutations
        self.shuffle()

    def shuffle(self):
        self.i = 0
        perm = torch.randperm(self.tbn, generator=self.gen)
        self.table = {k: v[perm] for k, v in self.table.items()}

    def next_batch(self, rn):
        sl = slice(self.i + self.rank, self.i + rn, self.world)
        out = {k: v[sl].to(self.device, non_blocking=True) for k, v in self.table.items()}
        self.i += rn
        if self.i + rn >= self.tbn:
            self.shuffle()
        return out
