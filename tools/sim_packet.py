"""Dev (CPU): how wide is the union of BVH nodes that the 64 direction-sorted rays of one surface point visit, against the
per-ray average?  Decides whether a wave-uniform (packet) traversal with scalar node fetches can pay.  python tools/sim_packet.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tensoflow_amd.lib as L
from tensoflow_amd.synth import sphere_torus_mesh, sphere_surface_points

lib = L.load()
verts, faces = sphere_torus_mesh(224, 448, 256, 128)
v = np.ascontiguousarray(verts, np.float32); f = np.ascontiguousarray(faces, np.int32)
nodes = np.zeros((2 * len(f), 8), np.float32); tris = np.zeros((len(f), 9), np.float32)
n = lib.tf_bvh_build_host(v.ctypes.data, len(v), f.ctypes.data, len(f), nodes.ctypes.data, tris.ctypes.data)
nodes = nodes[:n]
lo = nodes[:, 0:3]; hi = nodes[:, 4:7]
left = nodes[:, 3].copy().view(np.int32); count = nodes[:, 7].copy().view(np.int32)
print("nodes", n, "tris", len(f), "leaves", int((count > 0).sum()), "mean leaf", float(count[count > 0].mean()))

rng = np.random.default_rng(0)
pn = 48
pts, nrm, view = sphere_surface_points(pn, seed=6)

def frame(nv):
    a = np.where(np.abs(nv[0]) < 0.9, np.array([1.0, 0, 0]), np.array([0, 1.0, 0]))
    t = np.cross(nv, a); t /= np.linalg.norm(t); b = np.cross(nv, t)
    return t, b

def cosine_dirs(k):
    i = np.arange(k) + 0.5
    r = np.sqrt(i / k); phi = i * np.pi * (3 - np.sqrt(5))
    return np.stack([r * np.cos(phi), r * np.sin(phi), np.sqrt(1 - r * r)], -1)

def morton(a):
    q = np.clip((a * 16).astype(np.int64), 0, 15); code = np.zeros(len(a), np.int64)
    for b in range(4):
        code |= ((q[:, 0] >> b) & 1) << (2 * b + 1) | ((q[:, 1] >> b) & 1) << (2 * b)
    return np.argsort(code, kind="stable")

def visit(o, d):
    """-> per-ray inner-node visit counts, union count, per-ray leaf visits, union leaf visits, union triangles (no tmax culling)"""
    inv = 1.0 / d
    per_inner = np.zeros(len(d), np.int64); per_leaf = np.zeros(len(d), np.int64); per_tri = np.zeros(len(d), np.int64)
    u_inner = u_leaf = u_tri = 0
    stack = [(0, np.ones(len(d), bool))]
    while stack:
        k, act = stack.pop()
        t0 = (lo[k] - o) * inv; t1 = (hi[k] - o) * inv
        tn = np.minimum(t0, t1).max(-1); tf = np.maximum(t0, t1).min(-1)
        h = act & (tf >= np.maximum(tn, 0)) & (tn <= 10.0)
        if not h.any():
            continue
        if count[k] > 0:
            per_leaf += h; per_tri += h * count[k]; u_leaf += 1; u_tri += count[k]
        else:
            per_inner += h; u_inner += 1
            stack.append((left[k], h)); stack.append((left[k] + 1, h))
    return per_inner, u_inner, per_leaf, u_leaf, per_tri, u_tri

for label, sets in (("three sets sorted separately (today): 128 | 512 | 128", [128, 512, 128]), ("one set of 768 sorted jointly", [768])):
    A = dict(pi=[], ui=[], pl=[], ul=[], pt=[], ut=[])
    for p in range(pn):
        t, b = frame(nrm[p].astype(np.float64))
        for k in sets:
            loc = cosine_dirs(k)
            disc = loc[:, :2] * 0.5 + 0.5
            order = morton(disc)
            dirs = (loc[:, 0:1] * t + loc[:, 1:2] * b + loc[:, 2:3] * nrm[p]).astype(np.float64)[order]
            for w in range(0, k, 64):
                d = dirs[w:w + 64]; d = np.where(np.abs(d) < 1e-9, 1e-9, d)
                o = pts[p].astype(np.float64) + d * (1e-5 + 2 * 2 / 511)
                pi, ui, pl, ul, pt, ut = visit(o, d)
                A["pi"].append(pi.mean()); A["ui"].append(ui); A["pl"].append(pl.mean()); A["ul"].append(ul); A["pt"].append(pt.mean()); A["ut"].append(ut)
    m = {k: float(np.mean(v)) for k, v in A.items()}
    print(f"{label}: inner nodes per ray {m['pi']:.1f}, union per 64-ray packet {m['ui']:.1f} (x{m['ui'] / m['pi']:.1f});  leaves per ray {m['pl']:.2f}, union {m['ul']:.1f};  "
          f"triangles per ray {m['pt']:.2f}, union {m['ut']:.1f}")
