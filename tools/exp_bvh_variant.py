"""Dev: the traversal alone under a VARIANT build of the library (tools/build_variant.sh: -DBVH_WIDE, -DBVH_LEAF=3 ...) on the bench's
own rays (one shade() call's direction rows): ms per launch and a checksum of the depths (every variant must return the same).
  python tools/exp_bvh_variant.py <lib.so> [points]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L

L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch

import bench
from tensoflow_amd.synth import scene_surface_points

dev = torch.device("cuda:0")
pn = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
R_, r_ = bench.HEADLINE_TORUS
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128), torus_r=r_, torus_R=R_)
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in scene_surface_points(pn, seed=6, torus_r=r_, torus_R=R_)]
out = sh.shade(pts, view, nrm, 128, 128)
dirs, live = out["_pos_dirs"].reshape(-1, 3).contiguous(), out["_pos_live"]


def trace():
    return sh.bvh.trace(pts, dirs, 1e-5, 2 * unit, live=live, hit_rows_only=True, want_hit=False)[2]


for _ in range(2):
    d = trace()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8):
    d = trace()
e1.record()
torch.cuda.synchronize()
res = dict(lib=os.path.basename(sys.argv[1]), rays_issued=int(dirs.shape[0]), rays_traced=int(live.sum()), ms=e0.elapsed_time(e1) / 8,
           pair_records=int(sh.bvh.n_pairs), record_dwords=int(L.load().tf_bvh_record_dwords()),
           hit_fraction=float((d < 10.0).float().mean()), depth_checksum=float(d.double().sum()))
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/bvh_variants.jsonl", "a") as f:
    f.write(json.dumps(res) + "\n")
