"""Dev: chunked two-stream software pipeline of MCShader.shade (front = prep/flow/dirs + inner/reduce on one stream, BVH +
compaction on the other) against the plain pass.  python tools/exp_pipeline.py [points] [chunks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
nchunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
S = 128
order = sh.slot_order(S, S)
lat = sh.latent(S)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def front(p, n, v):
    va = ops.view_angles(n, v)
    metallic, rough, albedo, cond_d, cond_s = sh.point_prep(p, va)
    ang_d, lq_d = ops.flow_sample(sh.flow_d.nets, cond_d, lat, None, precision=sh.precision, cache=sh.flow_d.cache)
    ang_s, lq_s = ops.flow_sample(sh.flow_s.nets, cond_s, lat, None, precision=sh.precision, cache=sh.flow_s.cache)
    dirs, wgt, smask, live = ops.shade_dirs(n, v, metallic, rough, albedo, ang_d, lq_d, sh.fixed_d, ang_s, lq_s)
    return dirs, wgt, live


def trace(p, dirs, live):
    inters, nn, depth, hit = sh.bvh.trace(p, dirs.reshape(-1, 3), 1e-5, 2 * sh.unit, live=live, slot_order=order, hit_rows_only=True)
    idx, count = ops.compact_mask(hit.view(torch.uint8))
    return inters, nn, depth, hit, idx, count


def back(dirs, wgt, tr):
    inters, nn, depth, hit, idx, count = tr
    d2 = dirs.reshape(-1, 3)
    hl = torch.empty_like(d2)
    ops.inner_light_indexed(sh.inner, inters, d2, nn, idx, count, depth, hl, near_eps=1e-5, exp_max=sh.exp_max,
                            precision=sh.inner_precision, cache=sh.inner_cache)
    return ops.shade_reduce_env(wgt, dirs, depth, hit.view(torch.uint8), hl, sh.env, S + sh.fixed_d.shape[0], S)[0]


@torch.no_grad()
def pipelined(depth_ahead=1):
    keep, cols = [], []
    P, N, V = pts.chunk(nchunk), nrm.chunk(nchunk), view.chunk(nchunk)
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    F, T = {}, {}
    ev_f, ev_t = {}, {}

    def do_front(c):
        with torch.cuda.stream(s1):
            F[c] = front(P[c], N[c], V[c])
            ev_f[c] = torch.cuda.Event(); ev_f[c].record(s1)

    def do_trace(c):
        with torch.cuda.stream(s2):
            s2.wait_event(ev_f[c])
            T[c] = trace(P[c], F[c][0], F[c][2])
            ev_t[c] = torch.cuda.Event(); ev_t[c].record(s2)

    def do_back(c):
        with torch.cuda.stream(s1):
            s1.wait_event(ev_t[c])
            cols.append(back(F[c][0], F[c][1], T[c]))

    for c in range(min(depth_ahead + 1, nchunk)):
        do_front(c); do_trace(c)
    for c in range(nchunk):
        if c + depth_ahead + 1 < nchunk:
            do_front(c + depth_ahead + 1); do_trace(c + depth_ahead + 1)
        do_back(c)
    cur.wait_stream(s1); cur.wait_stream(s2)
    keep.append((F, T))
    return torch.cat(cols), keep


def bench_it(fn, reps=5):
    for _ in range(2): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


t_plain, o = bench_it(lambda: sh.shade(pts, view, nrm, S, S))
ref = o["colors"]
print(f"plain shade: {t_plain:.2f} ms  ({pn / t_plain / 1e3:.3f} M points/s)")
for ahead in (0, 1, 2):
    t, (col, _) = bench_it(lambda: pipelined(ahead))
    print(f"pipelined x{nchunk} ahead={ahead}: {t:.2f} ms  ({pn / t / 1e3:.3f} M points/s)  identical colours: {torch.equal(col, ref)}")
