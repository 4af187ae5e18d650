"""Dev (round 5): which of the integral's three stage kernels can share a CU, and what it buys.
python tools/exp_cosched.py [points] -> ms alone at each launch budget (tf_set_launch_budget), then pairs on two streams.
Every timing is wall clock around R back-to-back launches per stream, inputs fixed, device synchronised on both sides."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
R = int(os.environ.get("REPS", 6))
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
prec_il = int(os.environ.get("IL_PREC", ops.PREC_F16X2))
res = {"points": pn, "il_precision": prec_il}
with torch.no_grad():
    va = ops.view_angles(nrm, view)
    metallic, rough, albedo, cond_d, cond_s = sh.point_prep(pts, va)
    order = sh.slot_order(128, 128)
    ang_d, lq_d = ops.flow_sample(sh.flow_d.nets, cond_d, sh.latent(128), None, precision=sh.precision, cache=sh.flow_d.cache)
    ang_s, lq_s = ops.flow_sample(sh.flow_s.nets, cond_s, sh.latent(128), None, precision=sh.precision, cache=sh.flow_s.cache)
    dirs, wgt, smask, live = ops.shade_dirs(nrm, view, metallic, rough, albedo, ang_d, lq_d, sh.fixed_d, ang_s, lq_s, slot_of_pos=order)
    d2 = dirs.reshape(-1, 3)

    def bvh(k=0):
        ops.set_launch_budget(bvh_blocks_per_cu=k)
        return sh.bvh.trace(pts, d2, 1e-5, 2 * sh.unit, live=live, hit_rows_only=True, want_hit=False)

    inters, nn, depth, _ = bvh()
    idx, count = ops.compact_below(depth, ops.MISS_DEPTH)
    torch.cuda.synchronize()
    n_hit = int(count)
    res["rays"], res["hit_rays"] = d2.shape[0], n_hit
    hl = torch.zeros_like(d2)
    hl_ref = None

    def inner(teams=0):
        ops.set_launch_budget(inner_teams=teams)
        ops.inner_light_indexed(sh.inner, inters, d2, nn, idx, count, depth, hl, near_eps=1e-5, exp_max=sh.exp_max,
                                precision=prec_il, cache=sh.inner_cache)

    def flow(wpb=0):
        ops.set_launch_budget(flow_waves_per_block=wpb)
        a = ops.flow_sample(sh.flow_d.nets, cond_d, sh.latent(128), None, precision=sh.precision, cache=sh.flow_d.cache)
        b = ops.flow_sample(sh.flow_s.nets, cond_s, sh.latent(128), None, precision=sh.precision, cache=sh.flow_s.cache)
        return a, b

    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fa, fb=None, reps=R):
        for _ in range(2):
            fa()
            if fb: fb()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            with torch.cuda.stream(sA):
                fa()
            if fb:
                with torch.cuda.stream(sB):
                    fb()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps

    # results must not depend on the budget
    inner(0); torch.cuda.synchronize(); hl_ref = hl.clone(); hl.zero_()
    inner(1); torch.cuda.synchronize()
    res["inner_one_team_bit_identical"] = bool(torch.equal(hl, hl_ref))
    (a0, _), _ = flow(0); (a1, _), _ = flow(8); (a2, _), _ = flow(4)
    torch.cuda.synchronize()
    res["flow_budget_bit_identical"] = bool(torch.equal(a0, a1) and torch.equal(a0, a2))
    _, _, dep3, _ = bvh(3)
    torch.cuda.synchronize()
    res["bvh_budget_bit_identical"] = bool(torch.equal(dep3, depth))
    print(json.dumps(res), flush=True)

    alone = {}
    for k in (0, 1, 2, 3, 4, 5, 6):
        alone[f"bvh_k{k}"] = timed(lambda: bvh(k))
    for w in (0, 8, 4):
        alone[f"flow_w{w}"] = timed(lambda: flow(w))
    for t in (0, 1):
        alone[f"inner_t{t}"] = timed(lambda: inner(t))
    res["alone_ms"] = {k: round(v, 3) for k, v in alone.items()}
    print(json.dumps(res["alone_ms"]), flush=True)

    if os.environ.get("MODE") == "reduce":
        # the per-pixel reduction (texel gathers of the environment light: dependent loads, few vector instructions) beside the
        # traversal (vector-issue bound): the one pairing tools/coexec.hip predicts to overlap (valu + chase 1.05-1.19 x)
        def reduce_():
            return ops.shade_reduce_env(wgt, dirs, depth.reshape(pn, -1), None, hl.reshape(pn, -1, 3), sh.env, 128 + sh.fixed_d.shape[0], 128, slot_of_pos=order)
        alone["reduce"] = timed(reduce_)
        out = {"reduce_alone_ms": round(alone["reduce"], 3)}
        for k in (0, 6, 5, 4):
            ms = timed(reduce_, lambda: bvh(k))
            out[f"reduce+bvh{k or 7}"] = dict(ms=round(ms, 3), serial_default=round(alone["reduce"] + alone["bvh_k0"], 3),
                                               serial_same_budget=round(alone["reduce"] + alone[f"bvh_k{k}"], 3))
        for w in (0, 8):
            ms = timed(reduce_, lambda: flow(w))
            out[f"reduce+flow{w or 12}"] = dict(ms=round(ms, 3), serial_default=round(alone["reduce"] + alone["flow_w0"], 3))
        ms = timed(reduce_, lambda: inner(0))
        out["reduce+inner2"] = dict(ms=round(ms, 3), serial_default=round(alone["reduce"] + alone["inner_t0"], 3))
        ops.set_launch_budget()
        print(json.dumps(out)); sys.exit(0)
    if os.environ.get("MODE") == "alone":
        ops.set_launch_budget()
        print(json.dumps(res)); sys.exit(0)
    pairs = {}

    def pair(name, fa, fb, ka, kb):
        ms = timed(fa, fb)
        pairs[name] = dict(ms=round(ms, 3), serial_default=round(alone[ka[0]] + alone[kb[0]], 3), serial_same_budget=round(alone[ka[1]] + alone[kb[1]], 3))
        print(name, json.dumps(pairs[name]), flush=True)

    pair("inner2+bvh7", lambda: inner(0), lambda: bvh(0), ("inner_t0", "inner_t0"), ("bvh_k0", "bvh_k0"))
    for k in (2, 3, 4):
        pair(f"inner1+bvh{k}", lambda: inner(1), lambda: bvh(k), ("inner_t0", "inner_t1"), ("bvh_k0", f"bvh_k{k}"))
    pair("flow12+bvh7", lambda: flow(0), lambda: bvh(0), ("flow_w0", "flow_w0"), ("bvh_k0", "bvh_k0"))
    for w, k in ((8, 2), (8, 3), (4, 3), (4, 4), (4, 5)):
        pair(f"flow{w}+bvh{k}", lambda: flow(w), lambda: bvh(k), ("flow_w0", f"flow_w{w}"), ("bvh_k0", f"bvh_k{k}"))
    pair("inner1+flow4", lambda: inner(1), lambda: flow(4), ("inner_t0", "inner_t1"), ("flow_w0", "flow_w4"))
    res["pairs"] = pairs

    # three streams: inner (1 team) + flow (4 waves) + traversal
    sC = torch.cuda.Stream()

    def triple(k, w, reps=R):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            with torch.cuda.stream(sA):
                inner(1)
            with torch.cuda.stream(sB):
                bvh(k)
            with torch.cuda.stream(sC):
                flow(w)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps

    for k, w in ((2, 4), (3, 4)):
        triple(k, w, 2)
        ms = triple(k, w)
        res[f"triple_inner1+bvh{k}+flow{w}"] = dict(ms=round(ms, 3), serial_default=round(alone["inner_t0"] + alone["bvh_k0"] + alone["flow_w0"], 3))
        print(f"triple k={k} w={w}", json.dumps(res[f"triple_inner1+bvh{k}+flow{w}"]), flush=True)
ops.set_launch_budget()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/cosched.json", "w"), indent=1)
print(json.dumps(res))
