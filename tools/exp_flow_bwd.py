"""Dev tool: tf_flow_logq_bwd alone at the material training step's size (2 048 points x 128 samples, twice per step), ms per call,
and its gradients against the first library variant run in the same gpurun call (gpurun_out/flow_bwd_ref.pt).
    TF_LIB=<path to a variant .so> python tools/exp_flow_bwd.py [pn] [sn] [masked]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensoflow_amd import lib as L  # noqa: E402

if os.environ.get("TF_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["TF_LIB"])
from tensoflow_amd import ops  # noqa: E402
from tensoflow_amd.shading import FlowParams  # noqa: E402
from tensoflow_amd.synth import random_mc_state  # noqa: E402


def main():
    pn = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    sn = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    masked = len(sys.argv) > 3 and sys.argv[3] == "masked"
    dev = torch.device("cuda:0")
    sd = random_mc_state(seed=4, R=32, flow_R=32, env_res=8)
    fp = FlowParams(sd, "flow_diffuse_copy.", dev)
    g = torch.Generator().manual_seed(1)
    cond = torch.rand(pn, 37, generator=g).to(dev)
    x = torch.rand(pn, sn, 2, generator=g).clamp(1e-3, 1 - 1e-3)
    w = (torch.randn(pn, sn, 1, generator=g) / (pn * sn))
    rid = None
    if masked:
        keep = torch.rand(pn, sn, generator=g) < 0.6
        rid = torch.arange(pn)[:, None].expand(pn, sn)[keep].to(dev)
        x, w = x[keep], w[keep]
    x, w = x.to(dev), w.to(dev)
    z, lq = ops.flow_logq(fp.nets, cond, x, rays_id=rid, precision=ops.PREC_F32)

    def run():
        return ops.flow_logq_bwd(fp.nets, cond, x, w, rays_id=rid, want_gx=True, z=z if os.environ.get('FB_Z') else None)

    grads, g_cond, g_x = run()
    flat = [t.clone() for k in range(2) for pair in grads[k] for t in pair] + [g_cond.clone(), g_x.clone()]
    ref_path = f"gpurun_out/flow_bwd_ref{'_masked' if masked else ''}.pt"
    if os.path.exists(ref_path):
        ref = torch.load(ref_path)
        errs = []
        for a, b in zip(flat, ref):
            b = b.to(dev)
            errs.append(float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30))
        print("max-err / max|grad| against the first variant, per tensor (16 net tensors, g_cond, g_x): " + " ".join(f"{e:.1e}" for e in errs))
        d = (flat[-1] - ref[-1].to(dev)).abs().reshape(-1)
        print(f"g_x: {int((d > 1e-4 * float(ref[-1].abs().max())).sum())} of {d.numel()} entries differ by more than 1e-4 of the largest")
    else:
        os.makedirs("gpurun_out", exist_ok=True)
        torch.save([t.cpu() for t in flat], ref_path)
    for rep in range(3):
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"{os.environ.get('TF_LIB', 'product')}{' +z' if os.environ.get('FB_Z') else ''}: rows {x.shape[0] * (1 if masked else sn)}: {e0.elapsed_time(e1) / 50:.3f} ms per call (incl. host-side folds)")


if __name__ == "__main__":
    main()
