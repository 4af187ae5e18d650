"""Dev: bvh_trace on the bench's rays with and without the hit-point / normal rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = 262144
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
out = sh.shade(pts, view, nrm, 128, 128)
dirs, live = out["_pos_dirs"].reshape(-1, 3), out["_pos_live"]
for kw in (dict(want_pos=True, want_nrm=True), dict(want_pos=True, want_nrm=True, want_hit=False), dict(want_pos=False, want_nrm=False), dict(want_pos=False, want_nrm=False, want_hit=False)):
    for _ in range(2): sh.bvh.trace(pts, dirs, 1e-5, 2 * sh.unit, live=live, hit_rows_only=True, **kw)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = sh.bvh.trace(pts, dirs, 1e-5, 2 * sh.unit, live=live, hit_rows_only=True, **kw)
    e1.record(); torch.cuda.synchronize()
    print(kw, f"{e0.elapsed_time(e1) / 5:.2f} ms", "hit frac", float((r[2] < 10).float().mean()))
