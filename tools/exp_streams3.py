"""Dev: inputs / outputs of flow_sample under two calls in flight vs serial."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tensoflow_amd import ops
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn, chunk = 524288, 131072
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
sh.inner_precision = ops.PREC_F16X2
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6)]
log = []
orig = ops.flow_sample
def spy(weights, cond, latent, jitter=None, want_bins=False, precision=1, cache=None):
    r = orig(weights, cond, latent, jitter, want_bins, precision, cache)
    ws = cache.ws
    n = int(ops.L.load().tf_flow_workspace_floats(0))
    log.append(dict(stream=int(torch.cuda.current_stream().cuda_stream), cond=cond.clone(), ang=r[0].clone(), P=ws[n:n + 2 * 64 * cond.shape[0]].clone(), frag=ws[:n].clone(), wsptr=ws.data_ptr()))
    return r
ops.flow_sample = spy
log2 = []
orig_va = ops.view_angles
def spy_va(normals, view):
    r = orig_va(normals, view)
    log2.append(dict(va=r.clone(), n=normals.clone(), v=view.clone(), ptr=r.data_ptr(), stream=int(torch.cuda.current_stream().cuda_stream)))
    return r
ops.view_angles = spy_va
import tensoflow_amd.shading as S
def run(ns):
    for _ in range(2):
        sh.shade_many(pts, view, nrm, 128, 128, chunk, n_streams=ns)
    torch.cuda.synchronize(); log.clear(); log2.clear()
    sh.shade_many(pts, view, nrm, 128, 128, chunk, n_streams=ns)
    torch.cuda.synchronize()
    return list(log), list(log2)
a, a2 = run(1)
for trial in range(2):
    b, b2 = run(2)
    for i, (x, y) in enumerate(zip(a2, b2)):
        print(f"view_angles call {i}: va equal {torch.equal(x['va'], y['va'])} normals equal {torch.equal(x['n'], y['n'])} view equal {torch.equal(x['v'], y['v'])} ptr {y['ptr']:#x} stream {y['stream']:#x}"
              + ("" if torch.equal(x['va'], y['va']) else f" differing rows {((x['va'] - y['va']).abs().amax(-1) > 0).nonzero()[:6, 0].tolist()}"), flush=True)
        if not torch.equal(x['va'], y['va']):
            rows = ((x['va'] - y['va']).abs().amax(-1) > 0).nonzero()[:5, 0]
            for rr in rows.tolist():
                again = orig_va(y['n'][rr:rr + 1].contiguous(), y['v'][rr:rr + 1].contiguous())
                print(f"   row {rr}: serial va {x['va'][rr].tolist()} concurrent va {y['va'][rr].tolist()} recomputed alone {again[0].tolist()} normal {y['n'][rr].tolist()} view {y['v'][rr].tolist()}", flush=True)
    for i, (x, y) in enumerate(zip(a, b)):
        msg = [f"call {i} (chunk {i // 2}, {'diffuse' if i % 2 == 0 else 'specular'}) stream {y['stream']:#x} ws {y['wsptr']:#x}:"]
        for k in ("cond", "frag", "P", "ang"):
            if not torch.equal(x[k], y[k]):
                d = (x[k] - y[k]).abs().reshape(-1)
                bad = (d > 0).nonzero()[:, 0]
                msg.append(f"{k} differs at {bad.numel()} elements (first {bad[:4].tolist()}, max {float(d.max()):.2e})")
        print(" ".join(msg), flush=True)
