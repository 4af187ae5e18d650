"""Dev: do the BVH traversal (vector / scalar issue bound) and the inner-light MLP (matrix pipe) overlap when launched on two
streams?  python tools/exp_overlap.py [points]   -> alone / concurrent timings for the two launch orders."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
with torch.no_grad():
    va = ops.view_angles(nrm, view)
    metallic, rough, albedo, cond_d, cond_s = sh.point_prep(pts, va)
    ang_d, lq_d = ops.flow_sample(sh.flow_d.nets, cond_d, sh.latent(128), None, precision=sh.precision, cache=sh.flow_d.cache)
    ang_s, lq_s = ops.flow_sample(sh.flow_s.nets, cond_s, sh.latent(128), None, precision=sh.precision, cache=sh.flow_s.cache)
    dirs, wgt, smask, live = ops.shade_dirs(nrm, view, metallic, rough, albedo, ang_d, lq_d, sh.fixed_d, ang_s, lq_s)
    d2 = dirs.reshape(-1, 3)
    order = sh.slot_order(128, 128)

    def bvh():
        return sh.bvh.trace(pts, d2, 1e-5, 2 * sh.unit, live=live, slot_order=order, hit_rows_only=True)

    inters, nn, depth, hit = bvh()
    idx, count = ops.compact_mask(hit.view(torch.uint8))
    hl = torch.empty_like(d2)

    def inner():
        ops.inner_light_indexed(sh.inner, inters, d2, nn, idx, count, depth, hl, near_eps=1e-5, exp_max=sh.exp_max,
                                precision=sh.inner_precision, cache=sh.inner_cache)

    def flow():
        ops.flow_sample(sh.flow_d.nets, cond_d, sh.latent(128), None, precision=sh.precision, cache=sh.flow_d.cache)
        ops.flow_sample(sh.flow_s.nets, cond_s, sh.latent(128), None, precision=sh.precision, cache=sh.flow_s.cache)

    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(first, second=None, reps=5):
        for _ in range(2):
            first()
            if second: second()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            with torch.cuda.stream(sA):
                first()
            if second:
                with torch.cuda.stream(sB):
                    second()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    tb, ti, tf = timed(bvh), timed(inner), timed(flow)
    print(f"points {pn}  hits {int(count)}  alone: bvh {tb:.2f} ms  inner {ti:.2f} ms  flow {tf:.2f} ms")
    print(f"inner || bvh   (inner first) {timed(inner, bvh):.2f} ms   (bvh first) {timed(bvh, inner):.2f} ms   serial {tb + ti:.2f}")
    print(f"inner || flow  (inner first) {timed(inner, flow):.2f} ms   (flow first) {timed(flow, inner):.2f} ms   serial {tf + ti:.2f}")
    print(f"bvh || flow    (bvh first) {timed(bvh, flow):.2f} ms   (flow first) {timed(flow, bvh):.2f} ms   serial {tf + tb:.2f}")
