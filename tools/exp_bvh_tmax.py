"""Dev (round 6, verdict item 2): what is a conservative occupancy PRE-TEST worth to the traversal?  Measured before building it: the
variant library (-DBVH_DEV_TMAX) takes a per-ray upper bound of the hit distance from outside (tf_bvh_dev_tmax) and starts the ray's
`best` there; everything else is the product kernel.  Bounds tried on the bench's own rays (one shade() call's direction rows):

  * `oracle`: the true depth (+ margin) for rays that hit, X for rays that miss (X = 10: no knowledge; 0.3 ... 0.01: a perfect test that
    retires a miss within X of its origin) -- the upper bound of what ANY pre-test can return;
  * `grid N`: a real conservative bound -- an N^3 occupancy grid of the mesh (triangle boxes rasterised, dilated by one cell), every ray
    marched through it at one-cell steps on the device (torch): t_limit = end of the last occupied stretch (+ one step); rays that
    never meet an occupied cell get 0 (retired at the first box test).

Depths must equal the unbounded traversal's bit for bit in every variant.
  python tools/exp_bvh_tmax.py build_variants/lib_bvhtmax.so [points] [stats]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L

L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch

import bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points

dev = torch.device("cuda:0")
pn = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
want_stats = len(sys.argv) > 3 and sys.argv[3] == "stats"
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
lib = L.load()
lib.tf_bvh_dev_tmax.argtypes = [C.c_void_p]
lib.tf_bvh_dev_tmax.restype = None
if want_stats:
    lib.tf_bvh_stats.argtypes = [C.c_void_p]
    st = (C.c_ulonglong * 8)()

out = sh.shade(pts, view, nrm, 128, 128)
dirs = out["_pos_dirs"].reshape(-1, 3).contiguous()
live = out["_pos_live"]
T = out["_pos_dirs"].shape[1]
m = dirs.shape[0]
n_live = int(live.sum())


def trace():
    return sh.bvh.trace(pts, dirs, 1e-5, 2 * unit, live=live, hit_rows_only=True, want_hit=False)[2]


def timed(reps=6):
    trace()
    torch.cuda.synchronize()
    if want_stats:
        lib.tf_bvh_stats(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        d = trace()
    e1.record()
    torch.cuda.synchronize()
    res = {"ms": e0.elapsed_time(e1) / reps}
    if want_stats:
        lib.tf_bvh_stats(st)
        res.update(pair_steps_per_traced_ray=st[0] / reps / n_live, leaf_visits=st[1] / reps / n_live, tri_tests=st[6] / reps / n_live,
                   spine_pushes=st[5] / reps / n_live, inner_simd_eff=st[0] / max(st[2], 1))
    return d, res


lib.tf_bvh_dev_tmax(None)
depth0, base = timed()
hit0 = depth0 < 10.0
print(f"{pn} points x {T} rays = {m / 1e6:.1f} M issued, {n_live / 1e6:.1f} M traced, hit fraction {float(hit0.float().mean()):.4f}")
print("no bound:", base)
results = {"rays_issued": m, "rays_traced": n_live, "no_bound": base}


def run(name, tmax, extra=None):
    tmax = tmax.contiguous()
    lib.tf_bvh_dev_tmax(tmax.data_ptr())
    d, r = timed()
    lib.tf_bvh_dev_tmax(None)
    r["depths_bit_identical"] = bool(torch.equal(d, depth0))
    r["speedup"] = base["ms"] / r["ms"]
    if extra:
        r.update(extra)
    print(name, r)
    results[name] = r


ONLY = os.environ.get("BVH_TMAX_ONLY")        # e.g. "grid_256": that variant alone (rocprofv3 --pmc passes: launches 2..7 = no bound, 9..14 = the variant)
for X in (10.0, 0.3, 0.1, 0.03, 0.01):
    if ONLY and ONLY != f"oracle_miss_{X}":
        continue
    run(f"oracle_miss_{X}", torch.where(hit0, depth0 * 1.0001 + 1e-5, torch.full_like(depth0, X)))

# ---- a real conservative bound from an N^3 occupancy grid
V = torch.from_numpy(verts).to(dev)[torch.from_numpy(faces).long().to(dev)]          # [F,3,3]
lo_s, hi_s = V.reshape(-1, 3).amin(0) - 1e-3, V.reshape(-1, 3).amax(0) + 1e-3
o_row = pts[:, None, :].expand(pn, T, 3).reshape(-1, 3)
org = o_row + dirs * 1e-5 + dirs * (2 * unit)
for N in (32, 64, 128, 256):
    if ONLY and ONLY != f"grid_{N}":
        continue
    cell = (hi_s - lo_s) / N
    tlo = ((V.amin(1) - lo_s) / cell).floor().long().clamp(0, N - 1)
    thi = ((V.amax(1) - lo_s) / cell).floor().long().clamp(0, N - 1)
    occ = torch.zeros(N, N, N, dtype=torch.bool, device=dev)
    span = int((thi - tlo).max()) + 1
    for dx in range(span):
        for dy in range(span):
            for dz in range(span):
                off = torch.tensor([dx, dy, dz], device=dev)
                idx = tlo + off
                ok = (idx <= thi).all(-1)
                idx = idx[ok]
                occ[idx[:, 0], idx[:, 1], idx[:, 2]] = True
    raw_frac = float(occ.float().mean())
    occ = torch.nn.functional.max_pool3d(occ[None, None].float(), 3, 1, 1)[0, 0] > 0       # dilation: point samples at one-cell steps see every cell a ray crosses
    h = float(cell.min())
    # march every ray until it has left the scene box
    inv = 1.0 / torch.where(dirs.abs() < 1e-12, torch.full_like(dirs, 1e-12), dirs)
    t1, t2 = (lo_s - org) * inv, (hi_s - org) * inv
    t_exit = torch.minimum(torch.maximum(t1, t2).amin(-1), torch.full((m,), 10.0, device=dev)).clamp_min(0)
    K = int(float(t_exit.max()) / h) + 2
    last = torch.full((m,), -1, dtype=torch.int32, device=dev)
    CH = 1 << 23
    for c0 in range(0, m, CH):
        sl = slice(c0, min(c0 + CH, m))
        o_c, d_c, te = org[sl], dirs[sl], t_exit[sl]
        l_c = last[sl]
        for k in range(K):
            t = k * h
            p = o_c + d_c * t
            q = ((p - lo_s) / cell).floor().long()
            inside = ((q >= 0) & (q < N)).all(-1) & (t <= te + h)
            q = q.clamp(0, N - 1)
            hit_c = occ[q[:, 0], q[:, 1], q[:, 2]] & inside
            l_c = torch.where(hit_c, torch.full_like(l_c, k), l_c)
        last[sl] = l_c
    tlim = torch.where(last >= 0, (last.float() + 1.5) * h, torch.zeros(m, device=dev))
    never = float((last < 0).float().mean())
    miss = ~hit0 & live.reshape(-1).bool()
    if os.environ.get("BVH_TMAX_SAVE"):          # for tools/exp_bvh_tmax_prof.py (the PMC passes run a script without torch indexing kernels)
        torch.save(dict(pts=pts.cpu(), dirs=dirs.cpu(), live=live.cpu(), tmax=tlim.cpu(), unit=unit, n_live=n_live, name=f"grid_{N}"), os.environ["BVH_TMAX_SAVE"])
    run(f"grid_{N}", tlim, dict(cell=h, occupied_fraction_raw=raw_frac, occupied_fraction_dilated=float(occ.float().mean()),
                                rays_without_any_occupied_cell=never, median_tlimit_of_misses=float(tlim[miss].median()),
                                mean_tlimit_of_misses=float(tlim[miss].mean()), march_steps_max=K))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(results, open(f"gpurun_out/bvh_tmax{'_stats' if want_stats else ''}.json", "w"), indent=1)
