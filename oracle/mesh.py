"""Oracle first-hit ray/mesh intersection (TEST INFRASTRUCTURE).

The reference traces visibility rays with the third-party `raytracing` package
(github.com/ashawkey/raytracing, unpinned; wrapper copy at raytracing/raytracer.py:1-54,
used at network/materialRenderer.py:149,253-263).  Its result is geometry-defined: the
nearest triangle hit along the ray, `depth = t` (10.0 = MAX_DIST on a miss, which is what
materialRenderer.py:261 tests with `depth >= 10`), `positions = o + t d`, and the
(unnormalised-then-normalised) geometric face normal (b-a)x(c-a).  The per-triangle test is
the published Moeller-Trumbore/iq form with the acceptance window
u>=0, v>=0, u+v<=1, t>=0.  Tie-breaking between coplanar/adjacent faces is unpinned.

This brute-force O(rays x triangles) version is the oracle for the BVH kernel.
"""
import numpy as np
import torch

MAX_DIST = 10.0


def ray_triangles(o, d, tri):
    """o,d [M,3]; tri [T,3,3] float32 -> (t [M], face [M] (-1 on miss))."""
    M = o.shape[0]
    best_t = torch.full((M,), MAX_DIST, dtype=o.dtype)
    best_f = torch.full((M,), -1, dtype=torch.long)
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    v1, v2 = b - a, c - a                       # [T,3]
    n = torch.cross(v1, v2, dim=-1)             # [T,3]
    chunk = max(1, (1 << 24) // max(1, tri.shape[0]))
    for s in range(0, M, chunk):
        oo, dd = o[s:s + chunk, None, :], d[s:s + chunk, None, :]   # [m,1,3]
        rov0 = oo - a[None]                                         # [m,T,3]
        q = torch.cross(rov0, dd.expand_as(rov0), dim=-1)
        det = 1.0 / (dd * n[None]).sum(-1)
        u = det * -(q * v2[None]).sum(-1)
        v = det * (q * v1[None]).sum(-1)
        t = det * -(n[None] * rov0).sum(-1)
        bad = (u < 0) | (u > 1) | (v < 0) | (u + v > 1) | (t < 0) | torch.isnan(t)
        t = torch.where(bad, torch.full_like(t, 1e6), t)
        tm, fi = t.min(dim=1)
        hit = tm < MAX_DIST
        best_t[s:s + chunk] = torch.where(hit, tm, best_t[s:s + chunk])
        best_f[s:s + chunk] = torch.where(hit, fi, best_f[s:s + chunk])
    return best_t, best_f


class BruteForceRayTracer:
    """Same call surface as raytracing.RayTracer (raytracing/raytracer.py:7-54)."""

    def __init__(self, vertices, triangles):
        v = torch.as_tensor(np.asarray(vertices), dtype=torch.float32)
        f = torch.as_tensor(np.asarray(triangles).astype(np.int64))
        self.tri = v[f]                                             # [T,3,3]
        n = torch.cross(self.tri[:, 1] - self.tri[:, 0], self.tri[:, 2] - self.tri[:, 0], dim=-1)
        self.normals = torch.nn.functional.normalize(n, dim=-1)

    def trace(self, rays_o, rays_d, inplace=False):
        prefix = rays_o.shape[:-1]
        o = rays_o.reshape(-1, 3).float()
        d = rays_d.reshape(-1, 3).float()
        t, f = ray_triangles(o, d, self.tri)
        pos = o + t[:, None] * d
        nrm = torch.where((f >= 0)[:, None], self.normals[f.clamp(min=0)], torch.zeros_like(pos))
        return pos.reshape(*prefix, 3), nrm.reshape(*prefix, 3), t.reshape(*prefix)


# ---------------------------------------------------------------- CPU BVH (oracle/bvh_cpu.c): the same first-hit contract at
# mesh sizes the brute-force form cannot reach (the bench scene has 265 k triangles)
_LIB = None


def build_bvh_cpu_lib(force=False):
    """gcc -O2 -fopenmp -ffp-contract=off oracle/bvh_cpu.c -> oracle/_bvh_cpu.so (test infrastructure; also built by
    __graft_entry__.build()).  Returns the path."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src, out = os.path.join(here, "bvh_cpu.c"), os.path.join(here, "_bvh_cpu.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out, "-lm"], check=True)
    return out


def _lib():
    global _LIB
    if _LIB is None:
        import ctypes as C
        L = C.CDLL(build_bvh_cpu_lib())
        L.obvh_build.restype = C.c_void_p
        L.obvh_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.obvh_free.argtypes = [C.c_void_p]
        L.obvh_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class BvhRayTracer:
    """BruteForceRayTracer's call surface over the C BVH: same per-triangle arithmetic as the brute-force form (t within an ulp or two, identical hit sets; tests/test_oracle_shading.py::test_cpu_bvh_equals_brute_force)."""

    def __init__(self, vertices, triangles):
        self.v = np.ascontiguousarray(np.asarray(vertices, dtype=np.float32))
        self.f = np.ascontiguousarray(np.asarray(triangles).astype(np.int32))
        self.h = _lib().obvh_build(self.v.ctypes.data, self.f.ctypes.data, len(self.f))
        tri = torch.from_numpy(self.v)[torch.from_numpy(self.f.astype(np.int64))]
        self.tri = tri
        self.normals = torch.nn.functional.normalize(torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=-1), dim=-1)

    def __del__(self):
        try:
            _lib().obvh_free(self.h)
        except Exception:
            pass

    def first_hit(self, o, d):
        """o, d [M,3] float32 tensors -> (t [M], face [M] long, -1 on a miss)."""
        o = np.ascontiguousarray(o.detach().numpy().astype(np.float32).reshape(-1, 3))
        d = np.ascontiguousarray(d.detach().numpy().astype(np.float32).reshape(-1, 3))
        t = np.empty(len(o), np.float32)
        f = np.empty(len(o), np.int32)
        _lib().obvh_trace(self.h, o.ctypes.data, d.ctypes.data, len(o), t.ctypes.data, f.ctypes.data)
        return torch.from_numpy(t), torch.from_numpy(f.astype(np.int64))

    def trace(self, rays_o, rays_d, inplace=False):
        prefix = rays_o.shape[:-1]
        o, d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
        t, f = self.first_hit(o, d)
        pos = o + t[:, None] * d
        nrm = torch.where((f >= 0)[:, None], self.normals[f.clamp(min=0)], torch.zeros_like(pos))
        return pos.reshape(*prefix, 3), nrm.reshape(*prefix, 3), t.reshape(*prefix)
