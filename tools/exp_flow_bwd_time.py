"""Dev tool: time tf_flow_logq_bwd on a training step's shape (2048 points x 128 samples).  TF_LIB=<variant .so> to compare builds."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from tensoflow_amd import lib as L  # noqa: E402
if os.environ.get("TF_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["TF_LIB"])
from tensoflow_amd import ops  # noqa: E402
from tensoflow_amd.synth import random_flow_state  # noqa: E402

dev = "cuda"
pn, sn = 2048, 128
sd = {}
random_flow_state(sd, "f.", torch.Generator().manual_seed(1), R=32)
w = [[(sd[f"f.flows.{b}.nn.{l}.weight"].to(dev), sd[f"f.flows.{b}.nn.{l}.bias"].to(dev)) for l in (1, 3, 5, 7)] for b in range(2)]
g = torch.Generator(device=dev).manual_seed(2)
cond = torch.randn(pn, 37, device=dev, generator=g)
cond[:, 30:] = 0
x = torch.rand(pn, sn, 2, device=dev, generator=g).clamp(1e-3, 1 - 1e-3)
gl = torch.randn(pn, sn, device=dev, generator=g) / (pn * sn)
for rid in (None, torch.arange(pn, device=dev)[:, None].expand(pn, sn).reshape(-1)[: pn * sn // 2].contiguous()):
    xx = x if rid is None else x.reshape(-1, 2)[: pn * sn // 2].contiguous()
    gg = gl if rid is None else gl.reshape(-1)[: pn * sn // 2].contiguous()
    for _ in range(3):
        ops.flow_logq_bwd(w, cond, xx, gg, rays_id=rid)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        grads, gc = ops.flow_logq_bwd(w, cond, xx, gg, rays_id=rid)
    torch.cuda.synchronize()
    print("rows", gg.numel(), "rays_id" if rid is not None else "dense", "ms per call", (time.perf_counter() - t0) / 10 * 1e3,
          "checksum", float(sum(float(a.abs().sum()) + float(b.abs().sum()) for net in grads for a, b in net)))
