"""Dev: material-stage training step (bench.train_probe's step) eager vs captured in a HIP graph.  python tools/exp_train_graph.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
from tensoflow_amd.synth import sphere_torus_mesh, sphere_surface_points
from tensoflow_amd.network.fields import MCShadingNetwork
verts, faces = sphere_torus_mesh(224, 448, 256, 128)
aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
S, pn = 128, 2048
torch.manual_seed(6033)
m = MCShadingNetwork({"nis_diffuse_sample_num": S, "nis_specular_sample_num": S}, (verts, faces), aabb, 2.0 / 511)
for fl in (m.flow_diffuse_copy, m.flow_specular_copy):
    for p in fl.parameters():
        p.requires_grad = False
m.train()
m.use_flow_diffuse_copy = m.use_flow_specular_copy = True
pts, nrm, view = [torch.from_numpy(a).to(dev) for a in sphere_surface_points(pn, seed=99)]
w = torch.rand(pn, 3, device=dev)
params = [p for p in m.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-4, capturable=True)


def step():
    opt.zero_grad(set_to_none=False)
    colors, out = m(pts, view, nrm, None, 600, True)
    loss = (colors * w).sum() + out["loss_nis"]
    loss.backward()
    opt.step()
    return loss


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for _ in range(3): step()
print(f"eager: {timeit(step):.2f} ms per step")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = step()
    print(f"graph replay: {timeit(g.replay):.2f} ms per step   loss {float(loss):.6f}")
except Exception as e:
    import traceback; traceback.print_exc(limit=30)
