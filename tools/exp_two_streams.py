"""Dev: K headline steps issued on ONE stream vs alternating between TWO (two batches in flight).  python tools/exp_two_streams.py [points] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tensoflow_amd.synth import sphere_surface_points
dev = torch.device("cuda:0")
pn = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sh, sd, verts, faces, aabb, unit = bench.build_scene(dev, 4, (224, 448, 256, 128))
batches = [[torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=6 + b)] for b in range(2)]
for _ in range(2): sh.shade(batches[0][0], batches[0][2], batches[0][1], 128, 128)


def run(n_streams, overlap_dirs=True):
    sh.overlap_dirs = overlap_dirs
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = []
    for k in range(K):
        p, n, v = batches[k % 2]
        if n_streams == 1:
            outs.append(sh.shade(p, v, n, 128, 128)["colors"])
        else:
            s = streams[k % n_streams]
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                outs.append(sh.shade(p, v, n, 128, 128)["colors"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    return dt * 1e3, [float(o.double().sum()) for o in outs[:2]]


for ns, ov in ((1, True), (2, True), (2, False), (3, False), (1, True)):
    ms, cs = run(ns, ov)
    print(f"{ns} stream(s), overlap_dirs={ov}: {ms:.2f} ms per step  {pn / ms / 1e3:.3f} M points/s  checksums {cs}")
