"""Dev: the inner-light kernels alone on synthetic hit rays (bench network state).  python tools/exp_il3.py [n_rays] [precision codes...]
codes (TfPrecision): 1 = f16x3 (staggered two-team kernel, the library default), 3 = f16x2, 2 = f16, 0 = exact fp32.
IL_SPARSE=<density>: a sorted hit list over a larger ray array (the bench's access pattern); IL_TEAMS=1: one team per workgroup;
TF_LIB=<variant .so> (tools/build_variant.sh, e.g. "-DTF_DEV" for the cycle stamps of profiles/r6_il3_cycle_stamps.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L
if os.environ.get("TF_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["TF_LIB"])
import torch
from tensoflow_amd import ops
from tensoflow_amd.shading import wn_weight
from tensoflow_amd.synth import random_mc_state
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7_424_837
precs = [int(a, 0) for a in sys.argv[2:]] or [1, 3, 2]
sd = random_mc_state(seed=4, R=32, flow_R=32, env_res=16)
sdd = {k: v.to(dev).float() for k, v in sd.items() if v.is_floating_point() and "inner_light" in k}
W = [(wn_weight(sdd, f"inner_light.{i}").contiguous(), sdd[f"inner_light.{i}.bias"].contiguous()) for i in (0, 2, 4, 6)]
if os.environ.get("IL_PERIODIC_W"):       # dev: weights periodic in the input index (period 48 = three k-steps): the operand stream the
    # -DIL3_ABLATE_WSTREAM=1 build feeds its MFMAs, but fetched in full from L2 -- separates the weight stream's cost from the lower
    # toggle rate of repeated operands
    W = [(w[:, torch.arange(w.shape[1], device=dev) % 48].contiguous(), b) for w, b in W]
g = torch.Generator(device=dev).manual_seed(1)
pos = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 0.8
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
nrm = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
depth = torch.rand(n, device=dev, generator=g) + 0.1
idx = torch.randperm(n, device=dev, generator=g)
if os.environ.get("IL_SPARSE"):
    # the bench's access pattern: the hit list is SORTED and covers ~15 % of the rows of much larger ray arrays (IL_SPARSE = density)
    dens = float(os.environ["IL_SPARSE"])
    n_all = int(n / dens)
    pos = (torch.rand(n_all, 3, device=dev, generator=g) * 2 - 1) * 0.8
    dirs = torch.nn.functional.normalize(torch.randn(n_all, 3, device=dev, generator=g), dim=-1)
    nrm = torch.nn.functional.normalize(torch.randn(n_all, 3, device=dev, generator=g), dim=-1)
    depth = torch.rand(n_all, device=dev, generator=g) + 0.1
    idx = torch.nonzero(torch.rand(n_all, device=dev, generator=g) < dens)[:, 0].contiguous()
    n = int(idx.numel())
    print(f"sparse sorted hit list: {n} of {n_all} rows")
    n_rows = n_all
else:
    n_rows = n
count = torch.tensor([n], dtype=torch.int64, device=dev)
if os.environ.get("IL_TEAMS"):      # 1: one team per workgroup (tf_set_launch_budget): a team's phases WITHOUT a partner on its SIMDs
    ops.set_launch_budget(inner_teams=int(os.environ["IL_TEAMS"]))
res = {}
for p in precs:
    cache = ops.PackCache()
    lights = torch.zeros(n_rows, 3, device=dev)
    for _ in range(2):
        ops.inner_light_indexed(W, pos, dirs, nrm, idx, count, depth, lights, precision=p, cache=cache)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.inner_light_indexed(W, pos, dirs, nrm, idx, count, depth, lights, precision=p, cache=cache)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    terms = {1: 3, 3: 2, 2: 1, 0: 0}[p & 0xff]
    res[p] = lights.clone()
    print(f"precision {p:#x}: {n} rays, {ms:.3f} ms, algorithmic {n * 326656 / ms / 1e9:.0f} TF/s, executed {n / 32 * 336 * terms * 32768 / ms / 1e9:.0f} TF/s, "
          f"finite {bool(torch.isfinite(lights).all())}, checksum {float(lights.double().sum()):.6f}", flush=True)
ks = list(res)
for a in ks[1:]:
    d = ((res[ks[0]] - res[a]).abs() / res[a].abs().clamp_min(1e-3)).max()
    print(f"max rel diff {ks[0]:#x} vs {a:#x}: {float(d):.3e}")
if os.environ.get("TF_LIB") or os.environ.get("TF_TIMING_ONLY"):
    import ctypes as C
    lib = L.load()
    if hasattr(lib, "tf_il3_stamps"):
        st = (C.c_ulonglong * 128)()
        lib.tf_il3_stamps.argtypes = [C.c_void_p]
        lib.tf_il3_stamps(st)
        print("raw stamps (relative to the earliest stamp 0; index 15 = stamp 0 of the next iteration):")
        t00 = min(st[wv * 16] for wv in range(8))
        for wv in range(8):
            print(f"  wave {wv}: " + " ".join(f"{q}:{(st[wv * 16 + q] - t00) if st[wv * 16 + q] else -1}" for q in range(16)))
        names = ["FE", "w", "M1", "w", "P1", "w", "M2", "w", "P2", "w", "M3", "w"]
        t0 = min(st[wv * 16] for wv in range(8))
        for wv in range(8):
            s_ = [st[wv * 16 + q] for q in range(13)]
            print(f"wave {wv} (team {wv >> 2}) start +{s_[0] - t0:6d}: " + " ".join(f"{nm} {s_[q + 1] - s_[q]:5d}" for q, nm in enumerate(names)) + f" | total {s_[12] - s_[0]}"
                  f" | F {st[wv * 16 + 13] - s_[0]} E {st[wv * 16 + 14] - st[wv * 16 + 13]}")
    sys.exit(0)
# small-count edge cases against the exact-fp32 kernel: 1, 63, 64, 65, 129 rays
for m in (1, 63, 64, 65, 129, 1000):
    c = torch.tensor([m], dtype=torch.int64, device=dev)
    a, b = torch.zeros(n_rows, 3, device=dev), torch.zeros(n_rows, 3, device=dev)
    ops.inner_light_indexed(W, pos, dirs, nrm, idx, c, depth, a, precision=1)
    ops.inner_light_indexed(W, pos, dirs, nrm, idx, c, depth, b, precision=0)
    sel = idx[:m]
    ok = bool(((a[sel] - b[sel]).abs() / b[sel].abs().clamp_min(1e-3)).max() < 1e-4) and int((a != 0).any(-1).sum()) == int((b != 0).any(-1).sum())
    print(f"count {m}: rows written {int((a != 0).any(-1).sum())}, agrees with the exact-fp32 kernel: {ok}")
