"""Dev tool: where do the launches of a shape-stage training step come from?  Runs bench.shape_train_probe's step under torch.profiler
and prints (a) device-kernel launches per top-level section of the step, (b) the autograd nodes / aten ops with the most launches."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import math
    from torch.profiler import ProfilerActivity, profile, record_function
    from tensoflow_amd.network.shapeRenderer import ShapeRenderer
    from tensoflow_amd.synth import pinhole_rays, random_sdf_state, random_shape_shader_state
    device = torch.device("cuda:0")
    R, n_rays = 300, 1024
    cfg = dict(gridSize=[R, R, R], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False, device="cuda",
               nerfDataType=True, clip_sample_variance=False, apply_occ_loss=False)
    r = ShapeRenderer(cfg, training=False)
    sd = {"sdf_network." + k: v for k, v in random_sdf_state(seed=1, R=R).items()}
    sd.update(random_shape_shader_state(seed=8))
    r.load_state_dict(sd, strict=False)
    r.train()
    o, d, radii, cos = [torch.from_numpy(a).to(device) for a in pinhole_rays(n_rays, seed=2)]
    near, far = r.near_far_from_sphere(o, d)
    batch = {"rays_o": o, "rays_d": d, "dirs": d, "radiis": radii, "rays_cos": cos}
    target = torch.rand(n_rays, 3, device=device)

    # finer sections: wrap sample_ray, the colour network and the sdf-alpha call of the renderer
    def wrap(obj, name, label):
        fn = getattr(obj, name)
        def inner(*a, **k):
            with record_function(label):
                return fn(*a, **k)
        setattr(obj, name, inner)
    wrap(r, "sample_ray", "SEC2 sample_ray")
    wrap(r, "compute_sdf_alpha", "SEC2 compute_sdf_alpha (fwd)")
    wrap(r.color_network, "forward", "SEC2 color_network (fwd)")
    wrap(r.sdf_network, "TV_loss_sdf", "SEC2 TV_loss_sdf (fwd)")

    def step(tag=False):
        rf = record_function if tag else (lambda name: __import__("contextlib").nullcontext())
        r.zero_grad(set_to_none=True)
        with rf("SEC build_mips"):
            r.color_network.envlight.build_mips()
        with rf("SEC render (sample_ray + render_core fwd)"):
            out = r.render(batch, near, far, None, perturb_overwrite=0, cos_anneal_ratio=0.5, is_train=True, step=2000)
        with rf("SEC loss"):
            loss = ((out["ray_rgb"] - target) ** 2).mean() + 0.1 * out["gradient_error"].mean() + 0.1 * out["loss_sparse"] \
                + 5e-4 * out["loss_hessian"] + out["loss_tv_sdf"]
        with rf("SEC backward"):
            loss.backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(True)
        torch.cuda.synchronize()
    ev = prof.events()
    kernels = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
    print("device kernels / memcpys in the step:", len(kernels))
    # launches per CPU op (self): an op's kernels are linked through e.kernels
    per_op = collections.Counter()
    per_op_time = collections.Counter()
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
            per_op[e.name] += len(e.kernels)
            per_op_time[e.name] += sum(k.duration for k in e.kernels)
    print("\nlaunches by the CPU op that issued them (top 40):")
    for name, n in per_op.most_common(40):
        print(f"  {n:5d}  {per_op_time[name] / 1e3:8.3f} ms  {name[:100]}")
    # per section / per autograd node: walk CPU events sorted by start; attribute each launching op to its outermost enclosing
    # range whose name starts with SEC, and to its innermost enclosing autograd node (name ends with Backward / Backward0 ...)
    cpu = sorted([e for e in ev if e.device_type == torch.autograd.DeviceType.CPU], key=lambda e: (e.time_range.start, -e.time_range.end))
    secs = [e for e in cpu if e.name.startswith("SEC ")]
    nodes = [e for e in cpu if "Backward" in e.name or e.name.endswith("Fn") or e.name.startswith("autograd::engine::evaluate_function")]
    sec_n, node_n = collections.Counter(), collections.Counter()
    for s_ in secs:
        if s_.name.startswith("SEC "):
            print(f"  {s_.name}: {(s_.time_range.end - s_.time_range.start) / 1e3:.3f} ms of host time")
    for e in cpu:
        if e.name in ("hipMemcpyWithStream", "hipStreamSynchronize", "hipDeviceSynchronize"):
            chain = [o.name for o in cpu if o is not e and o.time_range.start <= e.time_range.start and e.time_range.end <= o.time_range.end]
            print(f"  {e.name} ({(e.time_range.end - e.time_range.start):.0f} us) inside: {' > '.join(chain[-4:])}")
    # device launches in issue order by the outermost op under a top-level section (consecutive repeats folded)
    for sec in [x for x in secs if x.name.startswith("SEC ") and not x.name.startswith("SEC2")]:
        tops, cur_end = [], -1
        for e in cpu:
            if e is sec or e.name.startswith("SEC2") or not (sec.time_range.start <= e.time_range.start and e.time_range.end <= sec.time_range.end):
                continue
            if e.time_range.start >= cur_end:
                tops.append([e.name, 0])
                cur_end = e.time_range.end
            if e.kernels:
                tops[-1][1] += len(e.kernels)
        seq = []
        for name, n in tops:
            if n == 0:
                continue
            name = name.replace("autograd::engine::evaluate_function: ", "")
            if seq and seq[-1][0] == name:
                seq[-1][1] += n; seq[-1][2] += 1
            else:
                seq.append([name, n, 1])
        print(f"\n{sec.name}: launches in issue order (op x repeats: launches):")
        print("  " + " | ".join(f"{nm.replace('aten::', '')}x{r}:{n}" for nm, n, r in seq))
    for e in cpu:
        if not e.kernels:
            continue
        n = len(e.kernels)
        for s in secs:
            if s.time_range.start <= e.time_range.start and e.time_range.end <= s.time_range.end:
                sec_n[s.name] += n
                break
        best = None
        for s in nodes:
            if s.time_range.start <= e.time_range.start and e.time_range.end <= s.time_range.end and s is not e:
                if best is None or (s.time_range.end - s.time_range.start) < (best.time_range.end - best.time_range.start):
                    best = s
        node_n[best.name if best is not None else "(forward / no autograd node)"] += n
    secs2 = [e for e in cpu if e.name.startswith("SEC2 ")]
    sec2_n, sec2_t = collections.Counter(), collections.Counter()
    for e in cpu:
        if not e.kernels:
            continue
        for s2 in secs2:
            if s2.time_range.start <= e.time_range.start and e.time_range.end <= s2.time_range.end:
                sec2_n[s2.name] += len(e.kernels)
                sec2_t[s2.name] += sum(k.duration for k in e.kernels)
                break
    print("\nforward sub-sections (launches, device ms):")
    for k, v in sec2_n.most_common():
        print(f"  {v:5d}  {sec2_t[k] / 1e3:8.3f} ms  {k}")
    print("\nlaunches per section:")
    for k, v in sec_n.most_common():
        print(f"  {v:5d}  {k}")
    print("\nlaunches per innermost autograd node (top 40):")
    for k, v in node_n.most_common(40):
        print(f"  {v:5d}  {k[:110]}")
    # forward: by python function would need stacks; print the aten ops of the render section instead
    fwd = collections.Counter()
    rs = [s for s in secs if "render" in s.name]
    if rs:
        s = rs[0]
        for e in cpu:
            if e.kernels and s.time_range.start <= e.time_range.start and e.time_range.end <= s.time_range.end:
                fwd[e.name] += len(e.kernels)
        print("\nrender section, launches by op (top 30):")
        for k, v in fwd.most_common(30):
            print(f"  {v:5d}  {k[:100]}")


if __name__ == "__main__":
    main()
