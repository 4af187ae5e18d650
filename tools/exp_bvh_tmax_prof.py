"""Dev: the launches of tools/exp_bvh_tmax.py that the rocprofv3 --pmc passes look at, from a saved ray set (BVH_TMAX_SAVE): 7 launches
of bvh_trace_kernel without a bound, then 7 with the saved per-ray bound.  python tools/exp_bvh_tmax_prof.py <lib.so> <saved.pt>"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L

L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch

from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_torus_mesh

dev = torch.device("cuda:0")
d = torch.load(sys.argv[2])
pts, dirs, live, tmax = d["pts"].to(dev), d["dirs"].to(dev), d["live"].to(dev), d["tmax"].to(dev).contiguous()
verts, faces = sphere_torus_mesh(224, 448, 256, 128)
bvh = ops.Bvh(verts, faces, dev)
lib = L.load()
lib.tf_bvh_dev_tmax.argtypes = [C.c_void_p]
lib.tf_bvh_dev_tmax.restype = None
for ptr in (None, tmax.data_ptr()):
    lib.tf_bvh_dev_tmax(ptr)
    for _ in range(7):
        bvh.trace(pts, dirs, 1e-5, 2 * d["unit"], live=live, hit_rows_only=True, want_hit=False)
    torch.cuda.synchronize()
lib.tf_bvh_dev_tmax(None)
print("done", d["name"], d["n_live"])
