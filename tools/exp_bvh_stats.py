import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tensoflow_amd.lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "build_variants/lib_bvhstats.so")
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_torus_mesh, sphere_surface_points
dev = torch.device("cuda:0")
verts, faces = sphere_torus_mesh(224, 448, 256, 128)
bvh = ops.Bvh(verts, faces, dev)
print("pairs", bvh.n_pairs, "tris", len(faces))
pn = 4096
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
T = 768
d = torch.nn.functional.normalize(torch.randn(pn, T, 3, device=dev) + 1.2 * nrm[:, None], dim=-1)
d = torch.where(((d * nrm[:, None]).sum(-1, keepdim=True) < 0), -d, d).reshape(-1, 3).contiguous()
lib = L.load()
lib.tf_bvh_stats.argtypes = [C.c_void_p]
st = (C.c_ulonglong * 8)()
lib.tf_bvh_stats(st)
pos, n, depth, hit = bvh.trace(pts, d, 1e-5, 2 * 2 / 511)
torch.cuda.synchronize()
lib.tf_bvh_stats(st)
m = d.shape[0]
print(f"rays {m}: inner lane-steps/ray {st[0]/m:.1f}  leaf lane-steps/ray {st[1]/m:.2f}  wave inner iters x64 /ray {st[2]/m:.1f}  "
      f"wave leaf iters x64 /ray {st[3]/m:.1f}  inner SIMD eff {st[0]/max(st[2],1):.2f}  leaf SIMD eff {st[1]/max(st[3],1):.2f}  hit frac {hit.float().mean():.3f}")
print(f"max inner steps of one ray {st[7]}  triangles tested/ray {st[6]/m:.2f}")
print(f"spine entries tested/ray {st[4]/m:.1f}  spine pushes/ray {st[5]/m:.2f}")
