"""Dev: phase clocks of bvh_trace_kernel (library built with -DBVH_CLOCK).  python tools/exp_bvh_clock.py lib.so [points]"""
import os, sys, ctypes as C
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
lib.tf_bvh_clock.argtypes = [C.c_void_p]
ck = (C.c_ulonglong * 12)()
for _ in range(2): sh.shade(pts, view, nrm, 128, 128)
torch.cuda.synchronize(); lib.tf_bvh_clock(ck)
sh.shade(pts, view, nrm, 128, 128)
torch.cuda.synchronize(); lib.tf_bvh_clock(ck)
refill, inner, leaf, life, waves, e_min, e_max, s_min, retire, claim, startray, _ = [int(x) for x in ck]
us = lambda t: t * 0.01
print(f"{sys.argv[1]}: waves {waves}  kernel span {us(e_max - s_min):.0f} us  earliest wave end {us(e_min - s_min):.0f} us  mean wave life {us(life / waves):.0f} us")
print(f"  per wave: refill {us(refill / waves):.0f} us ({refill / life:.1%})  inner {us(inner / waves):.0f} us ({inner / life:.1%})  leaf {us(leaf / waves):.0f} us ({leaf / life:.1%})  "
      f"other {us((life - refill - inner - leaf) / waves):.0f} us")
print(f"  of the refill: result stores {us(retire / waves):.0f} us  unit claim + spine build {us(claim / waves):.0f} us  ray fetch + spine walk {us(startray / waves):.0f} us")
