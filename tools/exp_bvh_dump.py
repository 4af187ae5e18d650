"""Dev: dump first-hit results on the bench mesh for a library build (compare two builds bit for bit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L
out = sys.argv[1]
if len(sys.argv) > 2:
    L.LIB_PATH = os.path.abspath(sys.argv[2])
import torch
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_torus_mesh, sphere_surface_points
dev = torch.device("cuda:0")
verts, faces = sphere_torus_mesh(224, 448, 256, 128)
bvh = ops.Bvh(verts, faces, dev)
pn, T = 2048, 768
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
g = torch.Generator(device="cpu").manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(pn, T, 3, generator=g).to(dev) + nrm[:, None], dim=-1).reshape(-1, 3).contiguous()
pos, n, depth, hit = bvh.trace(pts, d, 1e-5, 2 * 2 / 511)
torch.save({"depth": depth.cpu(), "hit": hit.cpu()}, out)
print(out, "records", bvh.n_pairs, "hit frac", float(hit.float().mean()))
