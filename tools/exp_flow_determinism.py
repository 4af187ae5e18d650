import sys, os, torch
sys.path.insert(0, os.getcwd())
import tensoflow_amd.lib as L
if len(sys.argv) > 1: L.LIB_PATH = os.path.abspath(sys.argv[1])
from tensoflow_amd import ops
from tensoflow_amd.synth import random_mc_state
from tensoflow_amd.shading import FlowParams, sphere_latent
dev = torch.device("cuda:0")
sd = random_mc_state(seed=4, R=64, flow_R=64, env_res=16)
fp = FlowParams(sd, "flow_diffuse_copy.", dev)
pn = 65536
cond = torch.randn(pn, 37, device=dev)
lat = sphere_latent(128).to(dev)
outs = []
for i in range(4):
    a, l = ops.flow_sample(fp.nets, cond, lat, None, precision=ops.PREC_F16X3, cache=fp.cache)
    outs.append((a.clone(), l.clone()))
torch.cuda.synchronize()
for i in range(1, 4):
    d = (outs[i][0] - outs[0][0]).abs()
    print(sys.argv[1] if len(sys.argv) > 1 else "default", "run", i, "angles differ:", int((d > 0).sum()), "max", float(d.max()), "finite", bool(torch.isfinite(outs[i][0]).all()))
