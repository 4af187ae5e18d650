"""Dev: time the inner-light kernel alone on the bench scene's hit rays.  python tools/exp_il2.py [lib.so] [precision codes...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-":
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
precs = [int(a, 0) for a in sys.argv[2:]] or [2, 3, 1]
pn = 65536
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
out = sh.shade(pts, view, nrm, 128, 128)
dirs = out["dirs"].reshape(-1, 3)
inters, nr, depth, hit = sh.bvh.trace(pts, dirs, 1e-5, 2 * sh.unit, live=out["live"], slot_order=sh.slot_order(128, 128), hit_rows_only=True)
idx, count = ops.compact_mask(hit.view(torch.uint8))
n = int(count)
lights = torch.empty_like(dirs)
for p in precs:
    cache = ops.PackCache()
    for _ in range(2):
        ops.inner_light_indexed(sh.inner, inters, dirs, nr, idx, count, depth, lights, precision=p, cache=cache)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.inner_light_indexed(sh.inner, inters, dirs, nr, idx, count, depth, lights, precision=p, cache=cache)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    mf = {1: 1008, 3: 672, 2: 336}.get(p & 0xff, 0)
    print(f"precision {p:#x}: {n} hit rays, {ms:.3f} ms, algorithmic {n * 326656 / ms / 1e9:.0f} TF/s, executed {n / 32 * mf * 32768 / ms / 1e9:.0f} TF/s, checksum {float(lights[idx[:n]].double().sum()):.6f}", flush=True)
