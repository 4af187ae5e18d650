"""Dev: first-hit results of a library build against the brute-force tracer.   python tools/exp_bvh_check.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_torus_mesh, sphere_surface_points
from oracle import mesh as om
dev = torch.device("cuda:0")
verts, faces = sphere_torus_mesh(12, 24, 16, 8)
bvh = ops.Bvh(verts, faces, dev)
pts, nrm, view = [torch.from_numpy(a) for a in sphere_surface_points(256, seed=3)]
g = torch.Generator().manual_seed(0)
d = torch.nn.functional.normalize(torch.randn(256, 64, 3, generator=g) + nrm[:, None], dim=-1).reshape(-1, 3)
o = pts[:, None].expand(256, 64, 3).reshape(-1, 3).contiguous()
tr = om.BruteForceRayTracer(verts, faces)
ri, rn, rd = tr.trace(o + d * 1e-5 + 0.004 * d, d)
rh = rd.reshape(-1) < 10
for dyn in (True, False):
    pos, n, depth, hit = bvh.trace(o.to(dev), d.to(dev), 1e-5, 0.004, dynamic=dyn)
    h = hit.cpu().bool()
    both = h & rh
    print(f"dynamic={dyn}: hit mismatch {int((h != rh).sum())} of {h.numel()} (ref hits {int(rh.sum())}, false hits {int((h & ~rh).sum())}, "
          f"missed {int((~h & rh).sum())}), max depth err on common hits {float((depth.cpu() - rd.reshape(-1)).abs()[both].max()) if both.any() else 0:.2e}")
