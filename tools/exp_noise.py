"""Dev (round 5): does any forward kernel change a bit when ANOTHER kernel's waves are resident on the GPU beside it?  (hipcc's packed-fp32
form of view_angles_kernel did.)  Every op below runs alone, then again while a second stream keeps the GPU sprinkled with small
element-wise launches; outputs are compared bit for bit.  python tools/exp_noise.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tensoflow_amd.lib as L
if os.environ.get("TF_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["TF_LIB"])
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (64, 128, 96, 48))
pn = 49152
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=3)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(name, fn, reps=6):
    """fn() alone, then TWO instances of it in flight on two streams (the situation in which view_angles_kernel's packed form failed:
    the other stream's launches of the same pipeline resident beside it), started together, `reps` times."""
    ref = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    bad = 0
    for _ in range(reps):
        outs = []
        for st in (s1, s2, s1, s2):
            with torch.cuda.stream(st):
                outs.append(fn())
        torch.cuda.synchronize()
        for out in outs:
            for a, b in zip(ref, out):
                if not torch.equal(a, b):
                    bad += 1
    print(f"{name:40s}: {'IDENTICAL' if bad == 0 else f'{bad} outputs DIFFER'} with two instances in flight ({reps} x 4 launches)", flush=True)


with torch.no_grad():
    va = ops.view_angles(nrm, view)
    met, rough, alb, cond_d, cond_s = sh.point_prep(pts, va)
    lat = sh.latent(128)
    ang_d, lq_d = ops.flow_sample(sh.flow_d.nets, cond_d, lat, None, precision=sh.precision, cache=sh.flow_d.cache)
    ang_s, lq_s = ops.flow_sample(sh.flow_s.nets, cond_s, lat, None, precision=sh.precision, cache=sh.flow_s.cache)
    order = sh.slot_order(128, 128)
    dirs, wgt, smask, live = ops.shade_dirs(nrm, view, met, rough, alb, ang_d, lq_d, sh.fixed_d, ang_s, lq_s, slot_of_pos=order)
    d2 = dirs.reshape(-1, 3)
    inters, nn, depth, _ = sh.bvh.trace(pts, d2, 1e-5, 2 * unit, live=live, hit_rows_only=False, want_hit=False)
    idx, count = ops.compact_below(depth, ops.MISS_DEPTH)
    idx_sorted = torch.sort(idx[: int(count)])[0]
    cnt = count.clone()
    run("view_angles", lambda: [ops.view_angles(nrm, view)])
    run("point_prep", lambda: list(sh.point_prep(pts, va)))
    run("flow_sample f16x3", lambda: list(ops.flow_sample(sh.flow_d.nets, cond_d, lat, None, precision=ops.PREC_F16X3, cache=sh.flow_d.cache)))
    run("flow_logq f16x3", lambda: list(ops.flow_logq(sh.flow_d.nets, cond_d, ang_d, precision=ops.PREC_F16X3)))
    run("shade_dirs", lambda: list(ops.shade_dirs(nrm, view, met, rough, alb, ang_d, lq_d, sh.fixed_d, ang_s, lq_s, slot_of_pos=order)))
    run("bvh_trace", lambda: [t for t in sh.bvh.trace(pts, d2, 1e-5, 2 * unit, live=live, hit_rows_only=False, want_hit=False)[:3]])
    for ip, nm in ((ops.PREC_F16X3, "f16x3"), (ops.PREC_F16X2, "f16x2"), (ops.PREC_F16, "f16"), (ops.PREC_F32, "f32")):
        def il(ip=ip):
            hl = torch.zeros_like(d2)
            ops.inner_light_indexed(sh.inner, inters, d2, nn, idx_sorted, cnt, depth, hl, precision=ip)
            return [hl]
        run(f"inner_light {nm}", il, reps=3)
    hl = torch.zeros_like(d2)
    ops.inner_light_indexed(sh.inner, inters, d2, nn, idx_sorted, cnt, depth, hl, precision=ops.PREC_F16X3)
    run("shade_reduce_env", lambda: list(ops.shade_reduce_env(wgt, dirs, depth, None, hl, sh.env, 128 + 512, 128, slot_of_pos=order)))
    run("shade_reduce_aux", lambda: list(ops.shade_reduce_aux(wgt, smask, 128 + 512, 128, dirs=dirs, depth=depth, hit_lights=hl, env_base=sh.env, slot_of_pos=order)))
    run("cube_lookup", lambda: [ops.cube_lookup(sh.env, d2[: 1 << 20], apply_exp=True)])
    # shape stage
    from tests.conftest import Golden, AABB
    g = Golden("march_r32")
    from tensoflow_amd.march import SdfField
    from tensoflow_amd.shape_shading import ShapeShader
    f = SdfField(g.sd, AABB, [32, 32, 32], 3, device=dev)
    gen = torch.Generator().manual_seed(5)
    P = (torch.rand(400000, 3, generator=gen) * 2 - 1).to(dev)
    lv = (torch.rand(400000, generator=gen) * 3 - 0.5).to(dev)
    dd = torch.nn.functional.normalize(torch.randn(400000, 3, generator=gen), dim=-1).to(dev)
    dist = torch.full((400000,), 0.01, device=dev)
    for prec, nm in ((ops.PREC_F16X3, "f16x3"), (ops.PREC_F32, "f32")):
        run(f"sdf_alpha {nm}", lambda prec=prec: [t for t in f.sdf_alpha(P, lv, dist, dd, 20.0, 0.5, precision=prec) if t is not None], reps=4)
    run("sdf_forward", lambda: [t for t in ops.sdf_forward(f.packed, *f.W, P, lv, AABB) if t is not None], reps=4)
    ss = ShapeShader(g.sd, [g["env_spec0"], g["env_spec1"], g["env_spec2"]], g["env_diffuse"], g["fg_lut"], device=dev)
    nrm2 = torch.nn.functional.normalize(torch.randn(400000, 3, generator=gen), dim=-1).to(dev)
    feat = torch.randn(400000, 128, generator=gen).to(dev) * 0.3
    run("shape_shade", lambda: list(ss(P, nrm2, dd, feat)), reps=4)
    run("vm_gather", lambda: [ops.vm_gather(f.packed, P, lv, AABB)], reps=4)
print("done")
