"""Dev: the traversal at 5 / 6 / 7 resident workgroups per CU, interleaved, one process.  python tools/exp_bvh_k.py [points]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
with torch.no_grad():
    va = ops.view_angles(nrm, view)
    metallic, rough, albedo, cond_d, cond_s = sh.point_prep(pts, va)
    order = sh.slot_order(128, 128)
    ang_d, lq_d = ops.flow_sample(sh.flow_d.nets, cond_d, sh.latent(128), None, precision=sh.precision, cache=sh.flow_d.cache)
    ang_s, lq_s = ops.flow_sample(sh.flow_s.nets, cond_s, sh.latent(128), None, precision=sh.precision, cache=sh.flow_s.cache)
    dirs, wgt, smask, live = ops.shade_dirs(nrm, view, metallic, rough, albedo, ang_d, lq_d, sh.fixed_d, ang_s, lq_s, slot_of_pos=order)
    d2 = dirs.reshape(-1, 3)
    res = {}
    for rnd in range(4):
        for k in (7, 6, 5, 6, 7):
            ops.set_launch_budget(bvh_blocks_per_cu=k)
            sh.bvh.trace(pts, d2, 1e-5, 2 * sh.unit, live=live, hit_rows_only=True, want_hit=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                sh.bvh.trace(pts, d2, 1e-5, 2 * sh.unit, live=live, hit_rows_only=True, want_hit=False)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(k, []).append(e0.elapsed_time(e1) / 3)
    for k, v in sorted(res.items()):
        print(f"{k} workgroups per CU: " + " ".join(f"{x:.2f}" for x in v) + f"   mean {sum(v) / len(v):.2f} ms", flush=True)
