"""CPU checks of the drop-in boundary: the shared library builds, loads, and exports every symbol
include/tensoflow_hip.h declares; the ctypes table mirrors the header; host-side argument
validation fails loudly without touching a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def lib():
    from tensoflow_amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return L.load()


def _declared():
    txt = open(os.path.join(REPO, "include", "tensoflow_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", txt)) - {"tf_stream_t"})


def test_exports_every_declared_symbol(lib):
    from tensoflow_amd import lib as L
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(L.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.tf_version() >= 100


def test_packed_floats_and_validation(lib):
    from tensoflow_amd import lib as L
    d = L.TfVmDesc()
    d.C, d.n_levels = 36, 3
    for i in range(3):
        d.ph[i] = d.pw[i] = d.ll[i] = 300
    n = lib.tf_vm_packed_floats(C.byref(d))
    expect = 3 * 36 * (300 * 300 + 150 * 150 + 75 * 75) + 3 * 36 * (300 + 150 + 75)
    assert n == expect
    d.ph[0] = 302                         # not divisible by 4 -> invalid geometry
    assert lib.tf_vm_packed_floats(C.byref(d)) == 0
    # null-pointer / shape errors are reported through the return code + tf_last_error, before any launch
    aabb = (C.c_float * 6)(-1, -1, -1, 1, 1, 1)
    rc = lib.tf_vm_gather_fwd(C.byref(d), None, None, None, C.byref(aabb), 5, None, None)
    assert rc == -2 and b"divisible" in lib.tf_last_error()
    d.ph[0] = 300
    rc = lib.tf_vm_gather_fwd(C.byref(d), None, None, None, C.byref(aabb), 5, None, None)
    assert rc == -1 and b"null" in lib.tf_last_error()
    assert lib.tf_vm_gather_fwd(C.byref(d), None, None, None, C.byref(aabb), 0, None, None) == 0   # empty input is fine
    assert lib.tf_composite_fwd(None, None, None, 0, 0, 0, None, None, None, None) == 0
    assert lib.tf_flow_workspace_floats(10) > 2 * 64 * 10
    assert lib.tf_sdf_workspace_floats() > 8 * 56 * 64


def test_fused_stage_validation(lib):
    """The fused per-point, shape-shading and trace entry points validate their arguments on the host (no launch)."""
    from tensoflow_amd import lib as L
    assert lib.tf_point_workspace_floats() > 3 * 4 * 56 * 64 and lib.tf_shape_shade_workspace_floats() > 26 * 4096
    assert lib.tf_point_pack(None, None, 0, None) == -1 and b"null" in lib.tf_last_error()
    nets = L.TfPointNets()
    dummy = (C.c_float * 4)()
    assert lib.tf_point_pack(C.byref(nets), C.addressof(dummy), 4, None) == -2 and b"workspace too small" in lib.tf_last_error()
    assert lib.tf_shape_shade_pack(None, None, 0, None) == -1
    # n == 0 is a no-op, n < 0 a shape error, a bad mip count a shape error (checked before any pointer is dereferenced on the device)
    assert lib.tf_shape_shade_fwd(None, None, None, 3, None, 16, None, 256, 256, 0.08, 0.5, 0.0, None, None, None, None, 0,
                                  None, None, None, None, None) == 0
    assert lib.tf_shape_shade_fwd(None, None, None, 3, None, 16, None, 256, 256, 0.08, 0.5, 0.0, None, None, None, None, -1,
                                  None, None, None, None, None) == -2
    one = (C.c_float * 4)()
    ptrs = (C.c_void_p * 1)(C.addressof(one))
    res = (C.c_int32 * 1)(16)
    a = C.addressof(one)
    assert lib.tf_shape_shade_fwd(a, C.addressof(ptrs), C.addressof(res), 1, a, 16, a, 256, 256, 0.08, 0.5, 0.0, a, a, a, a, 5,
                                  a, None, None, None, None) == -2 and b"specular mips" in lib.tf_last_error()
    assert lib.tf_point_fwd(None, None, None, None, None, None, None, None, None, None, 0, 0.04, None, None, None, None, None, None) == 0
    frame = (C.c_float * 6)()
    assert lib.tf_bvh_trace(None, None, C.byref(frame), 1, None, None, 0, None, 0.0, 0.0, None, 5, None, None, None, None, 0, None,
                            None, None) == -2 and b"rays_per_origin" in lib.tf_last_error()
    assert lib.tf_bvh_trace(None, None, C.byref(frame), 1, None, None, 1, None, 0.0, 0.0, None, 5, None, None, None, None, 0, None,
                            None, None) == -1
    # unknown precision codes are rejected
    assert lib.tf_sdf_forward(None, None, None, None, None, None, 5, None, None, 7, None, 0, None) != 0


def test_bvh_build_host(lib):
    from tensoflow_amd.synth import sphere_torus_mesh
    v, f = sphere_torus_mesh(8, 12, 16, 8)
    nodes = np.zeros((2 * len(f), 8), np.float32)
    tris = np.zeros((len(f), 9), np.float32)
    n = lib.tf_bvh_build_host(v.ctypes.data, len(v), f.ctypes.data, len(f), nodes.ctypes.data, tris.ctypes.data)
    assert 0 < n <= 2 * len(f)
    nd = nodes[:n]
    cnt = nd[:, 7].view(np.int32)
    left = nd[:, 3].view(np.int32)
    leaves = cnt > 0
    assert cnt[leaves].sum() == len(f) and cnt.max() <= 4          # every triangle in exactly one leaf
    # every triangle lies inside its leaf box, every child box inside the root box
    for i in np.nonzero(leaves)[0]:
        t = tris[left[i]:left[i] + cnt[i]].reshape(-1, 3)
        assert (t >= nd[i, 0:3] - 1e-6).all() and (t <= nd[i, 4:7] + 1e-6).all()
    assert (nd[:, 0:3] >= nd[0, 0:3] - 1e-6).all() and (nd[:, 4:7] <= nd[0, 4:7] + 1e-6).all()
    # the reordered soup is a permutation of the input triangles
    a = np.sort(v[f].reshape(len(f), 9).round(6).view([("", np.float32)] * 9), axis=0)
    b = np.sort(tris.round(6).view([("", np.float32)] * 9), axis=0)
    assert (a == b).all()
    bad = f.copy(); bad[0, 0] = 10 ** 6
    assert lib.tf_bvh_build_host(v.ctypes.data, len(v), bad.ctypes.data, len(f), nodes.ctypes.data, tris.ctypes.data) == -2


def test_bvh_pack_host(lib):
    """Traversal layout: every triangle is reachable through exactly one leaf reference, the quantised child boxes contain
    the triangles of their leaves (rounded outward, within a few grid cells), the tree respects the depth bound of the device
    stack, triangles are stored as (a, b - a, c - a)."""
    from tensoflow_amd.synth import sphere_torus_mesh
    v, f = sphere_torus_mesh(16, 24, 32, 12)
    nodes = np.zeros((2 * len(f), 8), np.float32)
    tris = np.zeros((len(f), 9), np.float32)
    n = lib.tf_bvh_build_host(v.ctypes.data, len(v), f.ctypes.data, len(f), nodes.ctypes.data, tris.ctypes.data)
    pairs = np.zeros((n // 2 + 1, 8), np.uint32)
    t12 = np.zeros((len(f), 12), np.float32)
    frame = np.zeros(6, np.float32)
    npair = lib.tf_bvh_pack_host(nodes.ctypes.data, n, tris.ctypes.data, len(f), pairs.ctypes.data, t12.ctypes.data, frame.ctypes.data)
    assert npair == (n - 1) // 2
    org, scl = frame[:3].astype(np.float64), frame[3:].astype(np.float64)
    assert (org < nodes[0, 0:3]).all() and (org + 65535 * scl > nodes[0, 4:7]).all()      # the grid covers the root box

    def box(i, c):
        w = pairs[i, 3 * c:3 * c + 3].astype(np.int64)
        qlo = np.array([w[0] & 0xffff, w[0] >> 16, w[1] & 0xffff])
        qhi = np.array([w[1] >> 16, w[2] & 0xffff, w[2] >> 16])
        return org + qlo * scl, org + qhi * scl

    refs = pairs[:npair, 6:8].view(np.int32)
    seen = np.zeros(len(f), np.int32)
    visited = np.zeros(npair, np.int32)
    stack, max_depth = [(0, 1)], 0
    while stack:
        i, depth = stack.pop()
        visited[i] += 1
        max_depth = max(max_depth, depth)
        for c in range(2):
            r = int(refs[i, c])
            lo, hi = box(i, c)
            if r >= 0:
                assert r > i                                   # depth-first numbering
                stack.append((r, depth + 1))
            else:
                enc = ~r
                first, cnt = enc >> 3, enc & 7
                assert 1 <= cnt <= 4
                seen[first:first + cnt] += 1
                t = tris[first:first + cnt].reshape(-1, 3)
                assert (t.min(0) >= lo).all() and (t.max(0) <= hi).all()                   # conservative ...
                assert (t.min(0) - lo < 3 * scl + 1e-4).all() and (hi - t.max(0) < 3 * scl + 1e-4).all()   # ... and tight
    assert (seen == 1).all() and (visited == 1).all() and max_depth <= 31
    # numbering: the first records are the top of the tree in breadth-first order (level by level), the rest depth-first
    level = np.zeros(npair, np.int32)
    for i in range(npair):
        for c in range(2):
            r = int(refs[i, c])
            if r >= 0:
                level[r] = level[i] + 1
    n_top = min(127, npair)
    assert (np.diff(level[:n_top]) >= 0).all() and level[n_top - 1] <= level[n_top:].max()
    assert np.array_equal(t12[:, 0:3], tris[:, 0:3]) and np.array_equal(t12[:, 3:6], tris[:, 3:6] - tris[:, 0:3])
    assert np.array_equal(t12[:, 6:9], tris[:, 6:9] - tris[:, 0:3])
    # a mesh that fits one leaf still packs to one (degenerate) pair whose second box is empty
    v1 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    f1 = np.array([[0, 1, 2], [0, 1, 3]], np.int32)
    nodes1 = np.zeros((4, 8), np.float32); tris1 = np.zeros((2, 9), np.float32)
    n1 = lib.tf_bvh_build_host(v1.ctypes.data, 4, f1.ctypes.data, 2, nodes1.ctypes.data, tris1.ctypes.data)
    p1 = np.zeros((2, 8), np.uint32); t1 = np.zeros((2, 12), np.float32); fr1 = np.zeros(6, np.float32)
    assert n1 == 1 and lib.tf_bvh_pack_host(nodes1.ctypes.data, 1, tris1.ctypes.data, 2, p1.ctypes.data, t1.ctypes.data, fr1.ctypes.data) == 1
    assert p1[0, 6:8].view(np.int32).tolist() == [~((0 << 3) | 2), -1]
    assert (p1[0, 3] & 0xffff) > (p1[0, 4] >> 16)                                          # empty: lo.x > hi.x


def test_bvh_depth_bound_on_degenerate_mesh(lib):
    """A sliver fan (SAH degenerates to a chain) must still respect the depth bound."""
    k = 4000
    ang = np.linspace(0, 1e-3, k + 1).astype(np.float32)
    v = np.concatenate([np.zeros((1, 3), np.float32), np.stack([np.cos(ang), np.sin(ang), ang * 0 + np.arange(k + 1, dtype=np.float32) ** 3 * 1e-9], 1)])
    f = np.stack([np.zeros(k, np.int32), np.arange(1, k + 1, dtype=np.int32), np.arange(2, k + 2, dtype=np.int32)], 1)
    nodes = np.zeros((2 * k, 8), np.float32); tris = np.zeros((k, 9), np.float32)
    n = lib.tf_bvh_build_host(v.ctypes.data, len(v), f.ctypes.data, k, nodes.ctypes.data, tris.ctypes.data)
    assert n > 0
    cnt = nodes[:n, 7].view(np.int32); left = nodes[:n, 3].view(np.int32)
    depth = np.zeros(n, np.int32)
    for i in range(n):
        if cnt[i] == 0:
            depth[left[i]] = depth[left[i] + 1] = depth[i] + 1        # children always follow their parent
    assert depth.max() <= 31 and cnt[cnt > 0].sum() == k


def test_ops_fail_loudly_without_gpu():
    import torch
    from tensoflow_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.cube_lookup(torch.zeros(6, 4, 4, 3), torch.zeros(5, 3))


def test_late_round1_entry_points_validate(lib):
    """Entry points added late in round 1 (fixed-sampler direction sets, cube-map direction gradient, the plain-f16 mode) validate
    on the host before any launch."""
    a = C.addressof((C.c_float * 4)())
    assert lib.tf_bvh_record_dwords() in (8, 16)
    # tf_shade_dirs_fixed: empty work is a no-op, negative sizes a shape error, a missing sample table an argument error
    assert lib.tf_shade_dirs_fixed(None, None, None, None, None, None, None, 0, None, None, 0, 5, None, None, None, None, None) == 0
    assert lib.tf_shade_dirs_fixed(None, None, None, None, None, None, None, -1, None, None, 4, 5, None, None, None, None, None) == -2
    assert lib.tf_shade_dirs_fixed(a, a, a, a, a, a, None, 4, None, None, 4, 5, a, a, a, None, None) == -1 \
        and b"null sample pointer" in lib.tf_last_error()
    # tf_cube_lookup_bwd_dirs needs at least one of the two gradient outputs
    assert lib.tf_cube_lookup_bwd_dirs(a, 4, a, 0, 0, a, None, None, None) == 0
    assert lib.tf_cube_lookup_bwd_dirs(a, 4, a, 3, 0, a, None, None, None) == -1
    assert lib.tf_cube_lookup_bwd_dirs(a, 0, a, 3, 0, a, a, a, None) == -2
    # TF_PREC_F16 (2) is accepted by the flow / inner-light entry points (they proceed to the next check) and refused elsewhere
    assert lib.tf_inner_light_fwd(None, None, None, None, 5, 5.0, 2, None, None, 0, None) == -1 and b"null pointer" in lib.tf_last_error()
    assert lib.tf_inner_light_fwd(None, None, None, None, 5, 5.0, 9, None, None, 0, None) == -1 and b"unknown precision" in lib.tf_last_error()
    assert lib.tf_sdf_forward(None, None, None, None, None, None, 5, None, None, 2, None, 0, None) != 0


def test_round2_entry_points_validate(lib):
    """Arguments added in round 2 are checked on the host before any launch: the row range of tf_shade_dirs, the precision of
    tf_linear_*, the float-threshold compaction, the optional hit flags of tf_shade_reduce_env."""
    a = C.addressof((C.c_float * 16)())
    sd = lambda r0, rc: lib.tf_shade_dirs(a, a, a, a, a, a, a, 2, a, None, 3, a, a, 2, 5, a, a, a, None, None, None, r0, rc, None)
    assert sd(5, 4) == -2 and b"row range" in lib.tf_last_error()              # rows [5, 9) of T = 7
    assert sd(-1, 2) == -2
    assert lib.tf_shade_dirs(a, a, a, a, a, a, a, 2, a, None, 3, a, a, 2, 0, a, a, a, None, None, None, 0, -1, None) == 0      # no points: no-op
    assert lib.tf_linear_fwd(a, a, None, 4, 2, 2, 0, 0.0, 2, a, None, None) == -1 and b"precision" in lib.tf_last_error()      # TF_PREC_F16 is not offered
    assert lib.tf_linear_bwd(a, a, a, a, 4, 2, 2, 0, 0.0, 7, a, None, None, None, None, None) == -1
    assert lib.tf_linear_fwd(a, a, None, 0, 2, 2, 0, 0.0, 1, a, None, None) == 0                                              # n = 0 with the f16x3 option
    assert lib.tf_compact_below(None, 10.0, -1, None, None, None) == -2
    assert lib.tf_compact_below(None, 10.0, 4, None, None, None) == -1 and b"count is null" in lib.tf_last_error()
    # hit may be NULL (depth < TF_MISS_DEPTH); the other arrays may not
    # (env_base may be NULL as well since round 3 -- outer_light_version='direction': every light is read from hit_lights)
    assert lib.tf_shade_reduce_env(a, a, a, None, None, a, 4, 1e-5, 3, 2, 1, a, None, None, None, None) == -1 and b"null pointer" in lib.tf_last_error()
    # the direction-encoded outer light runs on the fp32-grade kernel only; idx / count are required
    assert lib.tf_outer_light_indexed_fwd(None, a, None, None, 4, 5.0, 1, a, a, 0, None) == -1 and b"idx / count_dev" in lib.tf_last_error()
    assert lib.tf_outer_light_indexed_fwd(None, a, a, a, 4, 5.0, 2, a, a, 0, None) == -1 and b"TF_PREC_F16X3" in lib.tf_last_error()
    assert lib.tf_outer_light_indexed_fwd(None, a, a, a, 0, 5.0, 1, a, a, 0, None) == 0
    assert lib.tf_shade_reduce_env(a, a, a, None, a, a, 0, 1e-5, 3, 2, 1, a, None, None, None, None) == -2
