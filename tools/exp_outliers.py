"""Dev: where do the out-of-tolerance pixels of bench.py's psnr leg come from?  Per outlier point: hit-flag mismatches against the
oracle, largest flow-sample displacement, pixel error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from oracle import shading as osh
from oracle.mesh import BvhRayTracer
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(16384, seed=77)]
pts, nrm, view = pts[:n], nrm[:n], view[:n]
torch.set_num_threads(32)
tri = torch.from_numpy(verts)[torch.from_numpy(faces).long()]
tr = osh.MeshTracer(tri, bvh=BvhRayTracer(verts, faces))
S = 128
ref = {}
for c0 in range(0, n, 128):
    with torch.no_grad():
        r = osh.shade(sd, tr, unit, aabb, pts[c0:c0 + 128], view[c0:c0 + 128], nrm[c0:c0 + 128], S, S, n_fixed_diffuse=512, use_flow=True)
    for k in ("colors", "diffuse_hit", "diffuse_flow_angles", "specular_flow_angles", "specular_mask", "diffuse_dirs"):
        ref.setdefault(k, []).append(r[k])
ref = {k: torch.cat(v) for k, v in ref.items()}
sh.cull_dead_rays = False
out = sh.shade(pts.to(dev), view.to(dev), nrm.to(dev), S, S)
got = out["colors"].cpu()
err = (got - ref["colors"]).abs().amax(-1)
nd = S + 512
hit = out["hit"].cpu()
dh = (hit[:, :nd] != ref["diffuse_hit"])
da = (out["diffuse_angles"].cpu() - ref["diffuse_flow_angles"]).abs().amax(-1)
sa = (out["specular_angles"].cpu() - ref["specular_flow_angles"]).abs().amax(-1)
print("points", n, "outliers", int((err > 1e-4).sum()), "max err", float(err.max()))
print("diffuse hit mismatches total", int(dh.sum()), "of", dh.numel())
for i in (err > 1e-4).nonzero()[:, 0].tolist()[:40]:
    print(f"pt {i}: err {float(err[i]):.2e}  hit mismatches {int(dh[i].sum())}  max flow-sample move: diffuse {float(da[i].max()):.1e} specular {float(sa[i].max()):.1e}")
inl = err <= 1e-4
print("inliers: points with a hit mismatch", int((dh.sum(1) > 0)[inl].sum()), " max flow move among inliers", float(da[inl].max()), float(sa[inl].max()))
