"""Dev: per-stage HIP-event times of one 262 144-point shade() call of the bench scene under a given library (product or a
tools/build_variant.sh build).   python tools/exp_stages.py [lib.so] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L

if len(sys.argv) > 1 and sys.argv[1] != "-":
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch

import bench
from tensoflow_amd.shading import StageTimer
from tensoflow_amd.synth import scene_surface_points

dev = torch.device("cuda:0")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tr, tR = bench.HEADLINE_TORUS[1], bench.HEADLINE_TORUS[0]
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128), torus_r=tr, torus_R=tR)
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in scene_surface_points(262144, seed=6, torus_r=tr, torus_R=tR)]
for _ in range(2):
    c = sh.shade(pts, view, nrm, 128, 128)["colors"]
torch.cuda.synchronize()
sh.timer = StageTimer()
for _ in range(reps):
    c = sh.shade(pts, view, nrm, 128, 128)["colors"]
torch.cuda.synchronize()
s = sh.timer.summary()
print({k: round(v[0] / max(1, v[1]), 3) for k, v in s.items()}, "checksum", f"{float(c.double().sum()):.6f}")
