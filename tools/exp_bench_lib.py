"""Dev: run bench.py's headline pass against an alternative library build.  python tools/exp_bench_lib.py lib.so [points]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-":
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch, bench
from tensoflow_amd.shading import StageTimer
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
if os.environ.get("TF_BVH_STATIC"):          # dev experiment: one ray per lane, no refill (lanes of a wave stay in phase)
    _trace = sh.bvh.trace
    sh.bvh.trace = lambda *a, **k: _trace(*a, **{**k, "dynamic": False})
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
for _ in range(2): out = sh.shade(pts, view, nrm, 128, 128)
t = StageTimer(); sh.timer = t
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): out = sh.shade(pts, view, nrm, 128, 128)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(sys.argv[1] if len(sys.argv) > 1 else "-", f"{pn/dt/1e6:.3f} M points/s", {k: round(v[0] / 5, 2) for k, v in t.summary().items()}, "colors checksum", float(out["colors"].double().sum()))
try:
    import ctypes as C
    lib = L.load()
    lib.tf_bvh_stats.argtypes = [C.c_void_p]
    st = (C.c_ulonglong * 8)()
    lib.tf_bvh_stats(st)
    m = pn * 768 * 7
    print(f"bvh stats over {m} rays: inner lane-steps/ray {st[0]/m:.1f} leaf {st[1]/m:.2f} wave inner x64/ray {st[2]/m:.1f} wave leaf x64/ray {st[3]/m:.1f} "
          f"spine entries/ray {st[4]/m:.1f} pushes/ray {st[5]/m:.2f} triangles tested/ray {st[6]/m:.2f} max steps {st[7]}")
except AttributeError:
    pass
