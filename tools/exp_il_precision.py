"""Dev: per-ray error of the inner-light decoder's operand modes against an fp64 evaluation of the same net, next to the error of the
reference's own arithmetic (fp32 PyTorch on the CPU: oracle.shading.inner_light) against that fp64 evaluation.
python tools/exp_il_precision.py [n_rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import shading as osh
from tensoflow_amd import ops
from tensoflow_amd.shading import wn_weight
from tensoflow_amd.synth import random_mc_state
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
torch.set_num_threads(min(16, os.cpu_count() or 1))
nets = {"bench net (random_mc_state seed 4)": random_mc_state(seed=4, R=32, flow_R=32, env_res=16)}
# (shading_stress.npz holds only what differs from shading_default.npz: the raised gains)
zb = np.load(os.path.join(REPO, "tests", "golden", "shading_default.npz"))
z = np.load(os.path.join(REPO, "tests", "golden", "shading_stress.npz"))
stress = {k[3:]: torch.from_numpy(zb[k]) for k in zb.files if k.startswith("sd/")}
stress.update({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")})
nets["shading_stress net (gains raised: log-radiance spans [-1.8, 0.5])"] = stress
g = torch.Generator().manual_seed(1)
pos = (torch.rand(n, 3, generator=g) * 2 - 1) * 0.8
dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
modes = ((ops.PREC_F32, "exact fp32 MFMA"), (ops.PREC_F16X3, "f16x3"), (ops.PREC_F16X2, "f16x2 (activations rounded once)"), (ops.PREC_F16, "f16"))
def stats(a, ref):
    e = ((a.double() - ref).abs() / ref.abs().clamp_min(1e-3)).amax(-1)
    return f"max {float(e.max()):.2e}  99.9 % {float(e.quantile(0.999)):.2e}  rms {float((e ** 2).mean().sqrt()):.2e}"
for name, sd in nets.items():
    sd = {k: v for k, v in sd.items() if "inner_light" in k and v.is_floating_point()}
    ref64 = osh.inner_light({k: v.double() for k, v in sd.items()}, pos.double(), -dirs.double(), nrm.double())
    ref32 = osh.inner_light({k: v.float() for k, v in sd.items()}, pos, -dirs, nrm)
    print(f"== {name}: {n} rays, radiance range [{float(ref64.min()):.3g}, {float(ref64.max()):.3g}]")
    print(f"  reference arithmetic (fp32 PyTorch, CPU) vs fp64: {stats(ref32, ref64)}")
    sdd = {k: v.to(dev).float() for k, v in sd.items()}
    W = [(wn_weight(sdd, f"inner_light.{i}").contiguous(), sdd[f"inner_light.{i}.bias"].contiguous()) for i in (0, 2, 4, 6)]
    idx = torch.arange(n, device=dev)
    count = torch.tensor([n], dtype=torch.int64, device=dev)
    depth = torch.ones(n, device=dev)
    for p, label in modes:
        lights = torch.zeros(n, 3, device=dev)
        ops.inner_light_indexed(W, pos.to(dev), dirs.to(dev), nrm.to(dev), idx, count, depth, lights, precision=p, cache=ops.PackCache())
        torch.cuda.synchronize()
        print(f"  {label:36s} vs fp64: {stats(lights.cpu(), ref64)}   | vs the fp32 reference: {stats(lights.cpu(), ref32.double())}")
