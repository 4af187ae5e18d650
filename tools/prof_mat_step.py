"""Dev tool: where do the launches of a material-stage training step come from (bench.train_probe's step under torch.profiler)?"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from torch.profiler import ProfilerActivity, profile, record_function
    from tensoflow_amd.network.fields import MCShadingNetwork
    from tensoflow_amd.synth import sphere_surface_points, sphere_torus_mesh
    device = torch.device("cuda:0")
    verts, faces = sphere_torus_mesh(224, 448, 256, 128)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    S, pn = 128, 2048
    torch.manual_seed(6033)
    m = MCShadingNetwork({"nis_diffuse_sample_num": S, "nis_specular_sample_num": S, "outer_light_version": "envlight"}, (verts, faces), aabb, 2.0 / 511)
    for fl in (m.flow_diffuse_copy, m.flow_specular_copy):
        for p in fl.parameters():
            p.requires_grad = False
    m.train()
    m.use_flow_diffuse_copy = m.use_flow_specular_copy = True
    pts, nrm, view = [torch.from_numpy(a).to(device) for a in sphere_surface_points(pn, seed=99)]
    w = torch.rand(pn, 3, device=device)

    def step(tag=False):
        rf = record_function if tag else (lambda name: __import__("contextlib").nullcontext())
        m.zero_grad(set_to_none=True)
        with rf("SEC forward"):
            colors, out = m(pts, view, nrm, None, 600, True)
        with rf("SEC backward"):
            ((colors * w).sum() + out["loss_nis"]).backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(True)
        torch.cuda.synchronize()
    ev = prof.events()
    print("device kernels / memcpys in the step:", len([e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]))
    per_op, per_op_t = collections.Counter(), collections.Counter()
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
            per_op[e.name] += len(e.kernels)
            per_op_t[e.name] += sum(k.duration for k in e.kernels)
    print("\nlaunches by the CPU op that issued them (top 45):")
    for name, n in per_op.most_common(45):
        print(f"  {n:5d}  {per_op_t[name] / 1e3:8.3f} ms  {name[:100]}")
    cpu = sorted([e for e in ev if e.device_type == torch.autograd.DeviceType.CPU], key=lambda e: (e.time_range.start, -e.time_range.end))
    secs = [e for e in cpu if e.name.startswith("SEC ")]
    nodes = [e for e in cpu if "Backward" in e.name or e.name.endswith("Fn")]
    sec_n, node_n = collections.Counter(), collections.Counter()
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
        node_n[best.name if best is not None else "(no autograd node)"] += n
    sync = collections.Counter()
    for e in cpu:
        if any(k in e.name for k in ("Synchronize", "aten::item", "aten::nonzero", "_local_scalar_dense", "aten::masked_select", "aten::index", "hipMemcpy", "Memcpy")):
            sync[e.name] += 1
    print("\nhost syncs / dynamic-shape ops in the step:", dict(sync))
    for e in cpu:
        if e.name in ("hipMemcpyWithStream", "hipStreamSynchronize", "hipDeviceSynchronize", "hipMemcpyAsync"):
            chain = [o.name for o in cpu if o is not e and o.time_range.start <= e.time_range.start and e.time_range.end <= o.time_range.end]
            print(f"  {e.name} ({(e.time_range.end - e.time_range.start):.0f} us) inside: {' > '.join(chain[-4:])}")
    for s_ in secs:
        print(f"  {s_.name}: {(s_.time_range.end - s_.time_range.start) / 1e3:.3f} ms of host time")
    # the forward's device launches in issue order, by the outermost op under the section (consecutive repeats folded)
    fwd = [x for x in secs if x.name == "SEC forward"][0]
    tops, cur_end = [], -1
    for e in cpu:
        if e is fwd or not (fwd.time_range.start <= e.time_range.start and e.time_range.end <= fwd.time_range.end):
            continue
        if e.time_range.start >= cur_end:                 # not nested in the previous top-level op
            tops.append([e.name, 0])
            cur_end = e.time_range.end
        if e.kernels:
            tops[-1][1] += len(e.kernels)
    seq = []
    for name, n in tops:
        if n == 0:
            continue
        if seq and seq[-1][0] == name:
            seq[-1][1] += n; seq[-1][2] += 1
        else:
            seq.append([name, n, 1])
    print("\nforward launches in issue order (op x repeats: launches):")
    print("  " + " | ".join(f"{nm.replace('aten::', '')}x{r}:{n}" for nm, n, r in seq))
    by_line = collections.Counter()
    for e in cpu:
        if not e.kernels or not e.stack:
            continue
        fr = [f for f in e.stack if "/tensoflow_amd/" in f and "autograd.py" not in f and "/ops.py" not in f and "/lib.py" not in f]
        by_line[(fr[0] if fr else e.stack[0]).split("/tensoflow_amd/")[-1][:90]] += len(e.kernels)
    print("\nlaunches by innermost tensoflow_amd source line (forward ops; top 60):")
    for k, v in by_line.most_common(60):
        print(f"  {v:4d}  {k}")
    print("\nlaunches per section:", dict(sec_n))
    print("\nlaunches per innermost autograd node / Function (top 30):")
    for k, v in node_n.most_common(30):
        print(f"  {v:5d}  {k[:110]}")


if __name__ == "__main__":
    main()
