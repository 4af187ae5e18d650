"""Dev: per-key error of ShapeRenderer's validation branch / nvs against the reference golden, with the f16x3 and the exact-fp32 decoder."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import Golden, rel_err, true_rel_err
import test_gpu_renderers as T
from tensoflow_amd import ops

dev = torch.device("cuda:0")
cache = {}
golden = lambda n: cache.setdefault(n, Golden(n))


def run(tag):
    r, ge = T._eval_renderer(golden, dev)
    c = lambda k: ge[k].to(dev)
    with torch.no_grad():
        val = r.render_core(c("rays_o"), c("dirs"), c("dirs"), c("radiis"), c("rays_cos"), c("t_starts"), c("t_ends"), c("ray_indices"),
                            None, cos_anneal_ratio=1.0, step=300000, is_train=False)
    for k in T.VAL_KEYS:
        a, b = val[k].cpu().reshape(ge["val/" + k].shape), ge["val/" + k]
        print(f"{tag} val/{k:24s} rel {rel_err(a, b):.2e} true_rel {true_rel_err(a, b):.2e} max|ref| {float(b.abs().max()):.3f}")
    h, w = [int(v) for v in ge["nvs_hw"]]
    frame = r.nvs(ge["nvs_pose"].numpy(), ge["nvs_K"].numpy(), h, w)
    for k, v in frame.items():
        b = ge["nvs/" + k]
        print(f"{tag} nvs/{k:24s} rel {rel_err(torch.from_numpy(v), b):.2e} true_rel {true_rel_err(torch.from_numpy(v), b):.2e} max|ref| {float(b.abs().max()):.3f}")


run("f16x3")
sf, sa = ops.sdf_forward, ops.sdf_alpha
ops.sdf_forward = lambda *a, **k: sf(*a, **{**k, "precision": ops.PREC_F32})
ops.sdf_alpha = lambda *a, **k: sa(*a, **{**k, "precision": ops.PREC_F32})
run("f32  ")

# ---- which pixels of the nvs frame deviate
ops.sdf_forward, ops.sdf_alpha = sf, sa
r, ge = T._eval_renderer(golden, dev)
h, w = [int(v) for v in ge["nvs_hw"]]
frame = r.nvs(ge["nvs_pose"].numpy(), ge["nvs_K"].numpy(), h, w)
import numpy as np
bad = np.zeros((h, w), bool)
for k, v in frame.items():
    e = np.abs(v - ge["nvs/" + k].numpy()).max(-1)
    bad |= e > 1e-4
    print(k, "pixels > 1e-4:", int((e > 1e-4).sum()), "max", float(e.max()), "at", np.unravel_index(e.argmax(), e.shape))
ys, xs = np.nonzero(bad)
for y, x in zip(ys, xs):
    print("pixel", y, x, {k: (np.round(frame[k][y, x], 5).tolist(), np.round(ge["nvs/" + k].numpy()[y, x], 5).tolist()) for k in ("color", "normal", "occ_trace", "occ_predict", "spec_light")})
