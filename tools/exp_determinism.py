"""Dev check: every fused kernel run twice on identical inputs must return identical bits (a missed hardware hazard shows up
as run-to-run differences long before it shows up in a tolerance test).   python tools/exp_determinism.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tensoflow_amd.lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from tensoflow_amd import march, ops
from tensoflow_amd.shading import MCShader, wn_weight
from tensoflow_amd.shape_shading import ShapeShader
from tensoflow_amd.synth import (pinhole_rays, random_mc_state, random_sdf_state, random_shape_shader_state, sphere_surface_points,
                                 sphere_torus_mesh, synthetic_fg_lut)

dev = torch.device("cuda:0")
bad = 0


def check(name, fn, n=3):
    global bad
    ref = [t.clone() for t in fn()]
    worst = 0
    for _ in range(n - 1):
        for a, b in zip(ref, fn()):
            worst = max(worst, int((a != b).sum()))
    print(f"{name:28s} {'identical' if worst == 0 else f'DIFFERS in {worst} values'}")
    bad += worst > 0


sd = random_mc_state(seed=4, R=128, flow_R=128, env_res=32)
verts, faces = sphere_torus_mesh(48, 96, 64, 32)
aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
sh = MCShader(sd, verts, faces, aabb, 2.0 / 127, device=dev, n_fixed_diffuse=512)
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(8192, seed=3)]
for prec, tag in ((ops.PREC_F16X3, "f16x3"), (ops.PREC_F32, "f32"), (ops.PREC_F16, "f16")):
    sh.precision = prec
    check(f"shade ({tag})", lambda: [sh.shade(pts, view, nrm, 128, 128)[k] for k in ("colors", "diffuse_angles", "specular_angles", "diffuse_logq", "depth", "hit")])
W = [(wn_weight(sd, f"inner_light.{i}").to(dev), sd[f"inner_light.{i}.bias"].to(dev)) for i in (0, 2, 4, 6)]
m = 500_000
p3 = torch.rand(m, 3, device=dev) * 2 - 1
v3, n3 = torch.randn(m, 3, device=dev), torch.randn(m, 3, device=dev)
for prec in (ops.PREC_F32, ops.PREC_F16X3, ops.PREC_F16):
    check(f"inner_light (prec {prec})", lambda: [ops.inner_light(W, p3, v3, n3, precision=prec)])
R = 128
ssd = {"sdf_network." + k: v for k, v in random_sdf_state(seed=1, R=R).items()}
ssd.update(random_shape_shader_state(seed=8))
field = march.SdfField(ssd, aabb, [R, R, R], 3, device=dev)
o, d, radii, cos = [torch.from_numpy(a).to(dev) for a in pinhole_rays(4096, seed=2)]
near, far = march.near_far_from_sphere(o, d)
t0, t1, ridx = march.march_uniform(field, o, d, near, far, n_steps=128)
mid = (t0 + t1) * 0.5
P = (o[ridx] + d[ridx] * mid[:, None]).contiguous()
lv = torch.rand(P.shape[0], device=dev) * 2
for prec in (ops.PREC_F32, ops.PREC_F16X3):
    check(f"sdf_alpha (prec {prec})", lambda: [t for t in field.sdf_alpha(P, lv, (t1 - t0), d[ridx].contiguous(), 20.0, 0.5, precision=prec) if t is not None])
g = torch.Generator().manual_seed(1)
spec = [(0.5 * torch.randn(6, s, s, 3, generator=g) - 0.7).to(dev) for s in (16, 8, 4)]
diff = (0.5 * torch.randn(6, 4, 4, 3, generator=g) - 0.7).to(dev)
shp = ShapeShader(ssd, spec, diff, synthetic_fg_lut(), device=dev)
feat = torch.randn(P.shape[0], 128, device=dev) * 0.5
nn_ = torch.randn(P.shape[0], 3, device=dev)
check("shape_shade", lambda: list(shp(P, nn_, -d[ridx].contiguous(), feat)))
print("FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)
