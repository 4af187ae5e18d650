"""Dev: the traversal alone on the bench scene's rays, for alternative library builds.  python tools/exp_bvh_variants.py lib.so"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-":
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.shading import StageTimer
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = 262144
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
out = sh.shade(pts, view, nrm, 128, 128)
dirs, live = out["_pos_dirs"].reshape(-1, 3), out["_pos_live"]
def run():
    return sh.bvh.trace(pts, dirs, 1e-5, 2 * sh.unit, live=live, hit_rows_only=True, want_hit=False)
for _ in range(2): r = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): r = run()
e1.record(); torch.cuda.synchronize()
depth = r[2]
print(sys.argv[1] if len(sys.argv) > 1 else "-", f"{e0.elapsed_time(e1) / 5:.3f} ms per {dirs.shape[0]} rays; pairs {sh.bvh.n_pairs}; depth checksum {float(depth.double().sum()):.6f} hits {int((depth < 10).sum())}", flush=True)
