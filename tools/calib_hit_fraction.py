"""Dev (round 6, verdict item 9): the MEASURED hit fraction of the secondary rays of the bench scene (sphere r = 0.5 inside a torus
R = 0.75) as a function of the torus' tube radius, points area-uniform over sphere AND torus (SURVEY.md 8(d) config 3 asks for ~0.20).
  python tools/calib_hit_fraction.py [points] [R:r ...]   ->  gpurun_out/hit_fraction_calib.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tensoflow_amd import ops
from tensoflow_amd.shading import MCShader
from tensoflow_amd.synth import random_mc_state, scene_surface_points, sphere_surface_points, sphere_torus_mesh

dev = torch.device("cuda:0")
pn = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
radii = [tuple(float(x) for x in v.split(":")) for v in sys.argv[2:]] or [(0.75, 0.12), (0.65, 0.14), (0.66, 0.15), (0.64, 0.13), (0.67, 0.16)]      # (major R, tube r)
sd = random_mc_state(seed=4, R=512, flow_R=512, env_res=128)
aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
res = []
for R_, r in radii:
    verts, faces = sphere_torus_mesh(224, 448, 256, 128, torus_r=r, torus_R=R_)
    sh = MCShader(sd, verts, faces, aabb, 2.0 / 511, device=dev, n_fixed_diffuse=512)
    row = {"torus_R": R_, "torus_r": r}
    for name, gen in (("scene_points", lambda: scene_surface_points(pn, seed=16, torus_r=r, torus_R=R_)), ("sphere_points", lambda: sphere_surface_points(pn, seed=6))):
        pts, nrm, view = [torch.from_numpy(a).to(dev) for a in gen()]
        out = sh.shade(pts, view, nrm, 128, 128)
        depth = out["_pos_depth"]
        row[name + "_hit_fraction"] = float((depth < ops.MISS_DEPTH).float().mean())
        row[name + "_live_fraction"] = float(out["_pos_live"].float().mean())
    print(row, flush=True)
    res.append(row)
    del sh
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/hit_fraction_calib.json", "w"), indent=1)
