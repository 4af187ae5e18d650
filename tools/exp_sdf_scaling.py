import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tensoflow_amd import ops
from tensoflow_amd.synth import random_sdf_state
dev = torch.device("cuda:0")
R = 300
sd = {k: v.to(dev) for k, v in random_sdf_state(seed=1, R=R).items()}
packed = ops.VmPacked([sd[f"sdf_plane.{i}"] for i in range(3)], [sd[f"sdf_line.{i}"] for i in range(3)], 3)
aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
W = [sd["sdf_mat.0.weight"], sd["sdf_mat.0.bias"], sd["sdf_mat.2.weight"], sd["sdf_mat.2.bias"]]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for n in (32768, 131072, 524288, 1048576, 4194304):
    pts = torch.rand(n, 3, device=dev) * 1.8 - 0.9
    same = pts[:1].expand(n, 3).contiguous()
    coh = (torch.arange(n, device=dev)[:, None] * torch.tensor([1e-6, 2e-6, 3e-6], device=dev) % 1.8 - 0.9).contiguous()
    r = timeit(lambda: ops.sdf_forward(packed, *W, pts, None, aabb, want_feat=False))
    s_ = timeit(lambda: ops.sdf_forward(packed, *W, same, None, aabb, want_feat=False))
    c = timeit(lambda: ops.sdf_forward(packed, *W, coh, None, aabb, want_feat=False))
    print(f"n={n}: random {r:.3f} ms, same-point {s_:.3f} ms, coherent {c:.3f} ms; ideal-mfma {n/32*448*64/1024/2.1e9*1e3:.3f} ms")
