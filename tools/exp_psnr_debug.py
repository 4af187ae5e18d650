"""Dev: where does the HIP-vs-oracle colour difference on the bench's reduced scene come from?"""
import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import shading as osh
from tensoflow_amd.shading import MCShader
from tensoflow_amd.synth import random_mc_state, sphere_surface_points, sphere_torus_mesh

dev = torch.device("cuda:0")
sd = random_mc_state(seed=4, R=512, flow_R=512, env_res=128)
verts, faces = sphere_torus_mesh(24, 48, 32, 16)
aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]]); unit = 2.0 / 511
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(4096, seed=77)]
pts, nrm, view = pts[:n], nrm[:n], view[:n]
torch.set_num_threads(32)
tr = osh.MeshTracer(torch.from_numpy(verts)[torch.from_numpy(faces).long()])
with torch.no_grad():
    ref = osh.shade(sd, tr, unit, aabb, pts, view, nrm, 128, 128, n_fixed_diffuse=512, use_flow=True)
sh = MCShader(sd, verts, faces, aabb, unit, device=dev, n_fixed_diffuse=512)
out = sh.shade(pts.to(dev), view.to(dev), nrm.to(dev), 128, 128)
got = out["colors"].cpu()
err = ((got - ref["colors"]).abs() / ref["colors"].abs().clamp_min(1.0)).amax(-1)
print("max rel err", float(err.max()), "points > 1e-4:", int((err > 1e-4).sum()), "of", n)
print("ref keys", [k for k in ref.keys()])
worst = torch.argsort(err, descending=True)[:6]
for k in ("metallic", "roughness", "albedo"):
    print(k, float((out[k].cpu() - ref[k]).abs().max()))
for k_h, k_r in (("diffuse_angles", "diffuse_flow_angles"), ("specular_angles", "specular_flow_angles"), ("diffuse_logq", "diffuse_flow_logq"), ("specular_logq", "specular_flow_logq")):
    if k_r in ref:
        d = (out[k_h].cpu().reshape(ref[k_r].shape) - ref[k_r]).abs()
        print(k_h, "max", float(d.max()), "per-point max at worst pts", [float(d[i].max()) for i in worst])
hit = out["hit"].cpu()
for k in ref:
    if "hit" in k or "mask" in k or "visib" in k:
        print(k, tuple(ref[k].shape), ref[k].dtype)
print("worst points", worst.tolist(), [float(err[i]) for i in worst])
for k in ("diffuse_lin", "specular_lin"):
    for kr in ("diffuse_colors", "specular_colors"):
        if kr in ref and k.split("_")[0] == kr.split("_")[0]:
            d = (out[k].cpu() - ref[kr]).abs().amax(-1)
            print(k, "max abs", float(d.max()), "at worst pts", [float(d[i]) for i in worst], "ref val", [ref[kr][i].tolist() for i in worst[:2]])
