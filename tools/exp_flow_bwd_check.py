import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from conftest import Golden, AABB
from tensoflow_amd.network.flow import TensoFlow
g = Golden("tensoflow_r32"); dev = "cuda:0"
m = TensoFlow(2, AABB, device=dev, gridSize=[32, 32, 32]); m.load_state_dict(g.sd)
c = lambda k: g[k].to(dev)
z, logq = m(c("pts"), c("view_angles"), c("roughness"), c("x_rand"), return_jacobian=True)
(-(c("bwd_w") * logq).mean()).backward()
worst = 0
for name, p in m.named_parameters():
    if name in g.grad:
        scale = float(g.grad[name].abs().max()) + 1e-12
        err = float((p.grad.cpu() - g.grad[name]).abs().max()) / scale
        worst = max(worst, err)
        print(f"{name:28s} max|grad| {scale:.3e}  rel err {err:.2e}")
print("worst", worst)
