import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from conftest import Golden, AABB
from tensoflow_amd.network.fields import MCShadingNetwork
g = Golden("shading_grad"); dev = "cuda:0"
n_fd, n_fs, sn_d, sn_s = [int(v) for v in g["sn"]]
cfg = dict(gridSize=[32, 32, 32], light_reso=16, mat_grid=32, diffuse_sample_num=n_fd, nis_diffuse_sample_num=sn_d, nis_specular_sample_num=sn_s)
m = MCShadingNetwork(cfg, (g["verts"].numpy(), g["faces"].numpy()), AABB, float(g["unit_size"]))
m.load_state_dict(g.sd, strict=False)
for fl in (m.flow_diffuse_copy, m.flow_specular_copy):
    for p in fl.parameters(): p.requires_grad = False
m.eval()
colors, out = m(g["pts"].to(dev), g["view_in"].to(dev), g["normals_in"].to(dev), None, 600, False)
((colors * g["bwd_w"].to(dev)).sum() + out["loss_nis"]).backward()
for name, p in m.named_parameters():
    if name in g.grad and p.requires_grad:
        scale = float(g.grad[name].abs().max()) + 1e-12
        err = float((p.grad.cpu() - g.grad[name]).abs().max()) / scale
        if err > 1e-4 or "inner" in name or "outer" in name: print(f"{name:60s} max {scale:.3e} rel err {err:.2e}")
