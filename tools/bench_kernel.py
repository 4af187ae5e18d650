"""Micro-benchmark of single entry points against alternative builds of the library (dev tool).
usage: python tools/bench_kernel.py sdf|inner|flow|bvh [path/to/lib.so]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tensoflow_amd.lib as L
if len(sys.argv) > 2:
    L.LIB_PATH = os.path.abspath(sys.argv[2])
from tensoflow_amd import ops
from tensoflow_amd.synth import pinhole_rays, random_sdf_state, random_mc_state, sphere_torus_mesh

dev = torch.device("cuda:0")
which = sys.argv[1]

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

if which == "sdf":
    R = 300
    sd = {k: v.to(dev) for k, v in random_sdf_state(seed=1, R=R).items()}
    packed = ops.VmPacked([sd[f"sdf_plane.{i}"] for i in range(3)], [sd[f"sdf_line.{i}"] for i in range(3)], 3)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    n_rays, n_steps = 16384, 64
    o, d, radii, cos = [torch.from_numpy(a).to(dev) for a in pinhole_rays(n_rays, seed=2)]
    t = torch.linspace(1.0, 3.0, n_steps, device=dev)[None, :].expand(n_rays, n_steps)
    pts = (o[:, None] + d[:, None] * t[..., None]).reshape(-1, 3)
    inside = (pts.abs() < 1).all(-1)
    pts = pts[inside].contiguous()
    dirs = d[:, None].expand(n_rays, n_steps, 3).reshape(-1, 3)[inside].contiguous()
    n = pts.shape[0]
    dists = torch.full((n,), 2.0 / n_steps, device=dev)
    W = [sd["sdf_mat.0.weight"], sd["sdf_mat.0.bias"], sd["sdf_mat.2.weight"], sd["sdf_mat.2.bias"]]
    for name, level, prec in (("level=0 f32", torch.zeros(n, device=dev), 0), ("level=0 f16x3", torch.zeros(n, device=dev), 1),
                              ("level=U(0,2) f32", torch.rand(n, device=dev) * 2, 0), ("level=U(0,2) f16x3", torch.rand(n, device=dev) * 2, 1)):
        ms = timeit(lambda: ops.sdf_alpha(packed, *W, pts, level, dists, dirs, aabb, [2.0 / (R - 1)] * 3, 20.0, 1.0, precision=prec))
        print(f"sdf_alpha {name}: n={n} {ms:.2f} ms  {n/ms*1e-3:.1f} Msamples/s  {n/ms*1e-6*466944/1e3:.1f} TF/s  alg {n/ms*1e-6*18144:.0f} GB/s")
    ms = timeit(lambda: ops.sdf_forward(packed, *W, pts, None, aabb, want_feat=False))
    print(f"sdf_forward sdf-only: {ms:.2f} ms {n/ms*1e-3:.1f} Mevals/s")
    ms = timeit(lambda: ops.sdf_forward(packed, *W, pts, None, aabb, want_feat=True))
    print(f"sdf_forward full: {ms:.2f} ms {n/ms*1e-3:.1f} Mevals/s")
elif which == "inner":
    from tensoflow_amd.shading import wn_weight
    sd = random_mc_state(R=16, flow_R=16, env_res=8)
    W = [(wn_weight(sd, f"inner_light.{i}").to(dev), sd[f"inner_light.{i}.bias"].to(dev)) for i in (0, 2, 4, 6)]
    m = 3_000_000
    pts = torch.rand(m, 3, device=dev) * 2 - 1
    v = torch.randn(m, 3, device=dev); nr = torch.randn(m, 3, device=dev)
    for prec in (ops.PREC_F32, ops.PREC_F16X3, ops.PREC_F16):
        ms = timeit(lambda: ops.inner_light(W, pts, v, nr, precision=prec))
        print(f"inner_light m={m} precision={prec}: {ms:.2f} ms  {m*326656/ms*1e-9:.1f} TF/s (algorithmic)")
elif which == "bvh":
    verts, faces = sphere_torus_mesh(224, 448, 256, 128)
    bvh = ops.Bvh(verts, faces, dev)
    from tensoflow_amd.synth import sphere_surface_points
    pn = 16384
    pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
    T = 768
    d = torch.nn.functional.normalize(torch.randn(pn, T, 3, device=dev) + nrm[:, None], dim=-1).reshape(-1, 3)
    o = pts[:, None].expand(pn, T, 3).reshape(-1, 3).contiguous()
    d = torch.where(((d.view(pn, T, 3) * nrm[:, None]).sum(-1, keepdim=True) < 0), -d.view(pn, T, 3), d.view(pn, T, 3)).reshape(-1, 3).contiguous()
    for dyn in (False, True):
        ms = timeit(lambda: bvh.trace(o, d, 1e-5, 2 * 2 / 511, dynamic=dyn))
        print(f"bvh_trace dynamic={dyn} {o.shape[0]} rays, {len(faces)} tris: {ms:.2f} ms  {o.shape[0]/ms*1e-6:.2f} Grays/s")
elif which == "flow":
    from tensoflow_amd.shading import FlowParams, sphere_latent
    sd = random_mc_state(seed=4, R=64, flow_R=64, env_res=8)
    fp = FlowParams(sd, "flow_diffuse_copy.", dev)
    pn, sn = 16384, 128
    cond = torch.rand(pn, 37, device=dev)
    lat = sphere_latent(sn).to(dev)
    for prec in (ops.PREC_F32, ops.PREC_F16X3, ops.PREC_F16):
        ms = timeit(lambda: ops.flow_sample(fp.nets, cond, lat, precision=prec, cache=fp.cache))
        print(f"flow_sample pn={pn} sn={sn} precision={prec}: {ms:.3f} ms  {pn*sn/ms*1e-6:.2f} Gsamples/s  {pn*sn*49408/ms*1e-9:.1f} TF/s (reference flop count)")
    x = torch.rand(pn, sn, 2, device=dev)
    for prec in (ops.PREC_F32, ops.PREC_F16X3, ops.PREC_F16):
        ms = timeit(lambda: ops.flow_logq(fp.nets, cond, x, precision=prec))
        print(f"flow_logq pn={pn} sn={sn} precision={prec}: {ms:.3f} ms")
