"""Dev (round 6): cube-map tap sets -- the integer edge adjacency of cube.h against the floating-point re-projection it replaces
(variant: tools/build_variant.sh cubefloat "-DCUBE_TAPS_FLOAT=1" light.hip shade.hip shape_shade.hip).  Prints a digest of
ops.cube_lookup / cube_lookup_bwd on directions hugging every edge and corner for several resolutions: the two builds must print the
same lines.   python tools/exp_cube_taps.py [lib.so]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L

if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch

from tensoflow_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
n = 2_000_000
d = torch.randn(n, 3, generator=g)
d[: n // 2] = torch.sign(d[: n // 2]) * (1 + 0.01 * torch.randn(n // 2, 3, generator=g))          # corners
k = n // 4
ax = torch.randint(0, 3, (k,), generator=g)
d[n // 2: n // 2 + k] = torch.sign(d[n // 2: n // 2 + k]) * (1 + 0.005 * torch.randn(k, 3, generator=g))
d[n // 2: n // 2 + k][torch.arange(k), ax] = torch.rand(k, generator=g) * 2 - 1                     # edges: one coordinate free
d = d.to(dev)
for R in (4, 16, 128, 512):
    base = torch.randn(6, R, R, 3, generator=g).to(dev)
    out = ops.cube_lookup(base, d, apply_exp=False)
    gb = ops.cube_lookup_bwd(base, d[:200000], torch.ones(200000, 3, device=dev), apply_exp=False)
    print(R, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16], f"{float(out.double().sum()):.6f}", f"{float(gb.double().abs().sum()):.3f}")
