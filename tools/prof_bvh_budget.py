"""Dev (round 5): the traversal of one 65 536-point call of the bench at 3 .. 7 resident workgroups per CU, two launches each, in that order
(for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / --kernel-trace: dispatch order = budget order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = 65536
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
    print("traced rays", int(live.sum()), "of", d2.shape[0])
    for k in (3, 4, 5, 6, 8):           # 8 = all that fit (7)... the library caps the default at 6: ask for more explicitly
        ops.set_launch_budget(bvh_blocks_per_cu=k)
        for _ in range(2):
            sh.bvh.trace(pts, d2, 1e-5, 2 * sh.unit, live=live, hit_rows_only=True, want_hit=False)
    torch.cuda.synchronize()
