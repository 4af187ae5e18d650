"""Dev: traversal statistics of the bench's own pass (instrumented build -DBVH_STATS) -> profiles/bvh_stats.json.
python tools/exp_bvh_stats_bench.py build_variants/lib_bvhstats.so [points]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch, bench
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
lib = L.load()
lib.tf_bvh_stats.argtypes = [C.c_void_p]
st = (C.c_ulonglong * 8)()
sh.shade(pts, view, nrm, 128, 128)
torch.cuda.synchronize()
lib.tf_bvh_stats(st)                       # reads and clears
out = sh.shade(pts, view, nrm, 128, 128)
torch.cuda.synchronize()
lib.tf_bvh_stats(st)
live = int(out["_pos_live"].sum())
issued = pn * 768
hits = int((out["_pos_depth"] < 10.0).sum())
res = dict(source="instrumented build (-DBVH_STATS) of bvh.hip on one shade() call of the bench scene: " f"{pn} points x 768 rays",
           issued_rays=issued, traced_rays=live, hit_rays=hits,
           pair_steps_per_traced_ray=st[0] / live, leaf_visits_per_traced_ray=st[1] / live, triangle_tests_per_traced_ray=st[6] / live,
           spine_box_tests_per_traced_ray=st[4] / live, spine_pushes_per_traced_ray=st[5] / live,
           inner_simd_efficiency=st[0] / max(st[2], 1), leaf_simd_efficiency=st[1] / max(st[3], 1), max_pair_steps_of_one_ray=int(st[7]),
           pair_records=int(sh.bvh.n_pairs), triangles=int(len(faces)))
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bvh_stats.json", "w"), indent=1)
