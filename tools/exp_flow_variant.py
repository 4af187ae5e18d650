"""Dev: the flow sampler alone under a VARIANT build of the library (tools/build_variant.sh), at the bench's size (262 144 points x 128
samples, one lobe): ms per launch and checksums of the samples, log-densities and bins (a variant that only re-arranges instructions must
reproduce them bit for bit).   python tools/exp_flow_variant.py <lib.so> [points]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L

L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch

from tensoflow_amd import ops
from tensoflow_amd.shading import FlowParams, sphere_latent
from tensoflow_amd.synth import random_mc_state

dev = torch.device("cuda:0")
pn = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
sd = random_mc_state(seed=4, R=64, flow_R=64, env_res=16)
fp = FlowParams(sd, "flow_diffuse_copy.", dev)
g = torch.Generator().manual_seed(0)
cond = torch.rand(pn, 37, generator=g).to(dev)
lat = sphere_latent(128).to(dev)
for _ in range(2):
    ang, lq = ops.flow_sample(fp.nets, cond, lat, None, precision=ops.PREC_F16X3, cache=fp.cache)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ang, lq = ops.flow_sample(fp.nets, cond, lat, None, precision=ops.PREC_F16X3, cache=fp.cache)
e1.record()
torch.cuda.synchronize()
res = dict(lib=os.path.basename(sys.argv[1]), rows=pn * 128, ms=e0.elapsed_time(e1) / 10, angles_checksum=float(ang.double().sum()),
           logq_checksum=float(lq.double().sum()), angles_xor=int(ang.view(torch.int32).long().sum()))
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/flow_variants.jsonl", "a") as f:
    f.write(json.dumps(res) + "\n")
