"""Dev (round 5): the step's chunk loop with 1 / 2 / 3 shade() calls in flight (MCShader.shade_many): ms per 2^20 points, colours bit-identical?
python tools/exp_streams.py [points] [chunk]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
sh.inner_precision = int(os.environ.get("IL_PREC", ops.PREC_F16X2))
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
res = {}
ref = None
if os.environ.get("NO_OVERLAP_DIRS"):
    sh.overlap_dirs = False
if os.environ.get("SYNC_STAGE"):
    # serialise ONE stage across the streams: every call waits for the device before / after it
    name = os.environ["SYNC_STAGE"]
    orig = getattr(ops, name)
    def wrapped(*a, **k):
        torch.cuda.synchronize()
        r = orig(*a, **k)
        torch.cuda.synchronize()
        return r
    setattr(ops, name, wrapped)
for ns, ck in ((1, chunk), (2, chunk), (2, chunk), (1, chunk // 2), (1, chunk)):
    for _ in range(2):
        sh.shade_many(pts, view, nrm, 128, 128, ck, n_streams=ns)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 3
    for _ in range(R):
        outs = sh.shade_many(pts, view, nrm, 128, 128, ck, n_streams=ns)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / R
    col = torch.cat([o["colors"] for o in outs])
    if ref is None:
        ref = col.clone()
    d = (col - ref).abs()
    nd = int((d.amax(-1) > 0).sum())
    first = (d.amax(-1) > 0).nonzero()[:8, 0].tolist()
    res[f"streams{ns}_chunk{ck}"] = dict(ms=round(ms, 2), mpts=round(pn / ms / 1e3, 3), identical=bool(torch.equal(col, ref)), points_differing=nd, max_abs=float(d.max()))
    print(f"streams {ns} chunk {ck}: {ms:.2f} ms  {pn / ms / 1e3:.3f} M points/s  identical to serial: {torch.equal(col, ref)}; {nd} points differ, max |d| {float(d.max()):.3e}, "
          f"first {first}, finite {bool(torch.isfinite(col).all())}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/streams.json", "w"), indent=1)
