"""Dev (round 6): socket power and shader clock (rocm-smi, sampled every ~0.1 s from a side thread) while one kernel of the headline step
runs in a loop for a few seconds each: the f16x3 inner-light kernel, a pure MFMA stream (tf_probe_mfma_f16), the traversal, the flow
sampler.  Evidence for DESIGN.md section 3: which kernels run into the power limit and what clock they get.
  python tools/exp_power.py [seconds per kernel]        ->  gpurun_out/power_clock.json"""
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from tensoflow_amd import ops
from tensoflow_amd.synth import scene_surface_points

dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Package Power \(W\): ([0-9.]+)", out)
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            samples.append((time.time(), float(p.group(1)) if p else None, int(c.group(1)) if c else None))
        except Exception:
            pass
        time.sleep(0.05)


tr, tR = bench.HEADLINE_TORUS[1], bench.HEADLINE_TORUS[0]
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128), torus_r=tr, torus_R=tR)
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in scene_surface_points(131072, seed=6, torus_r=tr, torus_R=tR)]
out = sh.shade(pts, view, nrm, 128, 128)
dirs = out["_pos_dirs"].reshape(-1, 3).contiguous()
live = out["_pos_live"]
inters, nrmh, depth, _ = sh.bvh.trace(pts, dirs, 1e-5, 2 * unit, live=live, hit_rows_only=True, want_hit=False)
idx, count = ops.compact_below(depth, ops.MISS_DEPTH)
lights = torch.zeros_like(dirs)
va = ops.view_angles(nrm, view)
_, _, _, cond_d, _ = sh.point_prep(pts, va)


def k_inner():
    ops.inner_light_indexed(sh.inner, inters, dirs, nrmh, idx, count, depth, lights, near_eps=1e-5, exp_max=sh.exp_max,
                            precision=ops.PREC_F16X3, cache=sh.inner_cache)


def k_bvh():
    sh.bvh.trace(pts, dirs, 1e-5, 2 * unit, live=live, hit_rows_only=True, want_hit=False)


def k_flow():
    ops.flow_sample(sh.flow_d.nets, cond_d, sh.latent(128), None, precision=sh.precision)


def k_probe():
    ops.probe_mfma_f16_tflops(40000, dev, relu_like=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
res = {}
time.sleep(1.0)
marks = [("idle", time.time())]
for name, fn in (("inner_light_f16x3", k_inner), ("mfma_probe_relu_operands", k_probe), ("bvh_trace", k_bvh), ("flow_sample", k_flow)):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        n += 4
    marks.append((name, t0, time.time(), n))
    time.sleep(0.5)
stop.set()
th.join(timeout=2)
for m in marks[1:]:
    name, t0, t1, n = m
    s = [(p, c) for t, p, c in samples if t0 + 0.5 <= t <= t1 - 0.2 and p is not None and c is not None]
    if s:
        res[name] = dict(launches=n, ms_per_launch=(t1 - t0) / n * 1e3, samples=len(s), power_w_mean=sum(p for p, _ in s) / len(s),
                         power_w_max=max(p for p, _ in s), sclk_mhz_mean=sum(c for _, c in s) / len(s), sclk_mhz_min=min(c for _, c in s),
                         sclk_mhz_max=max(c for _, c in s))
idle = [(p, c) for t, p, c in samples if t < marks[0][1] and p is not None]
if idle:
    res["idle"] = dict(power_w_mean=sum(p for p, _ in idle) / len(idle), sclk_mhz_mean=sum(c for _, c in idle if c) / max(1, len(idle)))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/power_clock.json", "w"), indent=1)
print(json.dumps(res, indent=1))
