"""Dev: d logq / d x of FlowLogqFn (central differences of the HIP forward) against autograd of the fp64 oracle."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from conftest import Golden, AABB
from oracle import flow as ofl
from tensoflow_amd.network.flow import TensoFlow
g = Golden("tensoflow_r32"); dev = "cuda:0"
m = TensoFlow(2, AABB, device=dev, gridSize=[32, 32, 32]); m.load_state_dict(g.sd)
pts, va, rough = g["pts"], g["view_angles"], g["roughness"]
pn, sn = pts.shape[0], 64
gen = torch.Generator().manual_seed(3)
x = torch.rand(pn, sn, 2, generator=gen).clamp(1e-3, 1 - 1e-3)
w = torch.randn(pn, sn, 1, generator=gen)
sd64 = {k: v.double() if v.is_floating_point() else v for k, v in g.sd.items()}
x64 = x.double().requires_grad_(True)
_, lq = ofl.flow_logq(sd64, pts.double(), va.double(), rough.double(), x64, AABB.double() if torch.is_tensor(AABB) else AABB)
(lq * w.double()).sum().backward()
ref = x64.grad
xd = x.to(dev).requires_grad_(True)
_, lq2 = m(pts.to(dev), va.to(dev), rough.to(dev), xd, return_jacobian=True)
(lq2 * w.to(dev)).sum().backward()
got = xd.grad.cpu().double()
err = (got - ref).abs()
scale = ref.abs().max()
print("max |g|", float(scale), "max err", float(err.max()), "median err", float(err.median()), "q99", float(err.flatten().quantile(0.99)), "q999", float(err.flatten().quantile(0.999)))
print("fraction within 1e-2 of max|g|:", float((err < 1e-2 * scale).double().mean()), " within 1e-3:", float((err < 1e-3 * scale).double().mean()))
print("logq fwd err", float((lq2.detach().cpu().double() - lq.detach()).abs().max()))
