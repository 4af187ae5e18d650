"""Dev: accuracy / time of the inner-light decoder's operand splits (3-term f16x3, the two 2-term forms, plain f16) at pixel level."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from tensoflow_amd.shading import StageTimer, MCShader
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {1: "f16x3", 3: "f16x2 (128-ray staggered form)", 0x403: "f16x2 (column-owned kernel)", 2: "f16", 0x201: "ring f16x3", 0x203: "ring f16x2", 0x202: "ring f16", 0: "f32"}
# goldens
for g in ("shading_small", "shading_default"):
    z = np.load(os.path.join(REPO, "tests", "golden", g + ".npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    n_fd, n_fs, sn_d, sn_s = [int(v) for v in z["sn"]]
    pts, view, nrm = [torch.from_numpy(z[k]).to(dev) for k in ("pts", "view_in", "normals_in")]
    sh = MCShader(sd, z["verts"], z["faces"], aabb, float(z["unit_size"]), device=dev, n_fixed_diffuse=n_fd)
    for ip in (1, 3, 2, 0x201):
        sh.inner_precision = ip
        out = sh.shade(pts, view, nrm, sn_d, sn_s)["colors"].cpu()
        d = (out - torch.from_numpy(z["out/rgb_pr_nis"])).abs()
        print(g, names[ip], "vs golden: max", float(d.max()), "rms", float((d ** 2).mean().sqrt()), "pts", pts.shape[0], flush=True)
pn = 65536
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
base = None
for ip in (1, 3, 0x403, 2, 0x201, 0x203, 0x202, 0):
    sh.inner_precision = ip
    sh.timer = bench_t = StageTimer()
    for _ in range(2): out = sh.shade(pts, view, nrm, 128, 128)
    sh.timer = t = StageTimer()
    torch.cuda.synchronize()
    for _ in range(5): out = sh.shade(pts, view, nrm, 128, 128)
    torch.cuda.synchronize()
    c = out["colors"].double()
    hl = out["hit_lights"].double(); hit = out["hit"]
    if base is None:
        base = c; base_hl = hl
    d = (c - base).abs().amax(-1)
    dl = ((hl - base_hl).abs() / base_hl.abs().clamp_min(1e-3))[hit]
    print(names[ip], "inner_light ms", round(t.summary()["inner_light"][0] / 5, 3), "pixel max", float(d.max()), "rms", float((d ** 2).mean().sqrt()),
          "frac>1e-4", float((d > 1e-4).double().mean()), "| per-ray light rel: max", float(dl.max()), "rms", float((dl ** 2).mean().sqrt()), flush=True)
