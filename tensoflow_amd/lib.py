"""ctypes binding of libtensoflow_hip.so (the C ABI declared in include/tensoflow_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is
raised (the reference's native ops fail the same way through TORCH_CHECK,
network/renderutils/c_src/torch_bindings.cpp:27-31).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtensoflow_hip.so")

c_f = C.c_void_p  # device pointers are passed as raw addresses
i32, i64, f32, sz = C.c_int32, C.c_int64, C.c_float, C.c_size_t


class TfVmDesc(C.Structure):
    _fields_ = [("C", i32), ("n_levels", i32), ("ph", i32 * 3), ("pw", i32 * 3), ("ll", i32 * 3), ("texel_f16", i32)]


class TfSdfMlp(C.Structure):
    _fields_ = [("w1", c_f), ("b1", c_f), ("w2", c_f), ("b2", c_f), ("hidden", i32), ("app_dim", i32)]


class TfCouplingNet(C.Structure):
    _fields_ = [("w", c_f * 4), ("b", c_f * 4)]


class TfCouplingNetGrad(C.Structure):
    _fields_ = [("w", c_f * 4), ("b", c_f * 4)]


class TfMlp4(C.Structure):
    _fields_ = [("w", c_f * 4), ("b", c_f * 4)]


class TfMlp3(C.Structure):
    _fields_ = [("w", c_f * 3), ("b", c_f * 3)]


class TfShapeNets(C.Structure):
    _fields_ = [("mat_mlp", TfMlp3), ("inner_light", TfMlp3), ("inner_weight", TfMlp3)]


class TfPointNets(C.Structure):
    _fields_ = [("mat_w1", c_f * 3), ("mat_b1", c_f * 3), ("mat_w2", c_f * 3), ("mat_b2", c_f * 3),
                ("nis_w1", c_f * 2), ("nis_b1", c_f * 2), ("nis_w2", c_f * 2), ("nis_b2", c_f * 2)]


class TfBvhNode(C.Structure):
    _fields_ = [("lo", f32 * 3), ("left", i32), ("hi", f32 * 3), ("count", i32)]


P = C.POINTER
F3 = c_f * 3

# name -> (restype, argtypes); mirrors include/tensoflow_hip.h one to one
SIGNATURES = {
    "tf_version": (C.c_int, []),
    "tf_last_error": (C.c_char_p, []),
    "tf_set_launch_budget": (C.c_int, [i32, i32, i32]),
    "tf_probe_mfma_f16": (C.c_int, [i32, i32, c_f, i64, P(C.c_double), c_f]),
    "tf_vm_packed_floats": (sz, [P(TfVmDesc)]),
    "tf_vm_pack_fwd": (C.c_int, [P(TfVmDesc), P(F3), P(F3), c_f, c_f]),
    "tf_vm_pack_to_f16": (C.c_int, [P(TfVmDesc), c_f, c_f, c_f]),
    "tf_vm_pack_bwd": (C.c_int, [P(TfVmDesc), c_f, P(F3), P(F3), c_f]),
    "tf_vm_gather_fwd": (C.c_int, [P(TfVmDesc), c_f, c_f, c_f, P(f32 * 6), i64, c_f, c_f]),
    "tf_vm_gather_bwd": (C.c_int, [P(TfVmDesc), c_f, c_f, c_f, P(f32 * 6), i64, c_f, c_f, c_f]),
    "tf_sdf_workspace_floats": (sz, []),
    "tf_sdf_forward": (C.c_int, [P(TfVmDesc), c_f, P(TfSdfMlp), c_f, c_f, P(f32 * 6), i64, c_f, c_f, i32, c_f, sz, c_f]),
    "tf_sdf_alpha_fwd": (C.c_int, [P(TfVmDesc), c_f, P(TfSdfMlp), c_f, c_f, c_f, c_f, P(f32 * 6), P(f32 * 3), f32, f32,
                                   i64, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, sz, c_f]),
    "tf_sdf_alpha_bwd_workspace_floats": (sz, [i64]),
    "tf_sdf_alpha_bwd": (C.c_int, [P(TfVmDesc), c_f, P(TfSdfMlp), c_f, c_f, c_f, c_f, P(f32 * 6), P(f32 * 3), f32, f32, i64,
                                   c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, sz, c_f]),
    "tf_composite_fwd": (C.c_int, [c_f, c_f, c_f, i64, i64, i32, c_f, c_f, c_f, c_f]),
    "tf_composite_bwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, i64, i64, i32, c_f, c_f, c_f]),
    "tf_flow_workspace_floats": (sz, [i64]),
    "tf_flow_sample_fwd": (C.c_int, [P(TfCouplingNet * 2), c_f, c_f, c_f, i64, i32, c_f, c_f, c_f, i32, c_f, sz, c_f]),
    "tf_flow_logq_fwd": (C.c_int, [P(TfCouplingNet * 2), c_f, c_f, c_f, i64, i32, i64, c_f, c_f, c_f, i32, c_f, sz, c_f]),
    "tf_shape_glue_pre_fwd": (C.c_int, [c_f, c_f, c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, f32, f32, i32, c_f]),
    "tf_shape_glue_pre_bwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, f32, f32, i32, i64, c_f, c_f, c_f]),
    "tf_shape_glue_post_fwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i64, c_f, c_f, c_f]),
    "tf_shape_glue_post_bwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, c_f, c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "tf_normalize3_fwd": (C.c_int, [c_f, c_f, P(f32 * 3), i64, c_f, c_f, c_f]),
    "tf_normalize3_bwd": (C.c_int, [c_f, c_f, P(f32 * 3), c_f, c_f, i64, c_f, c_f, c_f]),
    "tf_linear_fwd": (C.c_int, [c_f, c_f, c_f, i64, i32, i32, i32, f32, i32, c_f, c_f, c_f]),
    "tf_linear_bwd": (C.c_int, [c_f, c_f, c_f, c_f, i64, i32, i32, i32, f32, i32, c_f, c_f, c_f, c_f, c_f, c_f]),
    "tf_linear_bwd_fused": (C.c_int, [c_f, c_f, c_f, c_f, i64, i32, i32, i32, f32, i32, i32, f32, i32, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "tf_pwquad_eval": (C.c_int, [c_f, c_f, i64, i32, c_f, c_f, c_f, c_f]),
    "tf_flow_bwd_workspace_floats": (sz, [i64]),
    "tf_flow_logq_bwd": (C.c_int, [P(TfCouplingNet * 2), c_f, c_f, c_f, c_f, i64, i32, i64, c_f, P(TfCouplingNetGrad * 2), c_f, c_f, c_f, c_f, sz, c_f]),
    "tf_bvh_record_dwords": (C.c_int32, []),
    "tf_cube_lookup_fwd": (C.c_int, [c_f, i32, c_f, i64, i32, c_f, f32, c_f, c_f]),
    "tf_cube_lookup_bwd": (C.c_int, [c_f, i32, c_f, i64, i32, c_f, c_f, c_f]),
    "tf_cube_lookup_bwd_dirs": (C.c_int, [c_f, i32, c_f, i64, i32, c_f, c_f, c_f, c_f]),
    "tf_cube_lookup_mips_fwd": (C.c_int, [C.POINTER(c_f), C.POINTER(i32), i32, c_f, c_f, i64, i32, c_f, c_f]),
    "tf_cube_lookup_mips_bwd": (C.c_int, [C.POINTER(c_f), C.POINTER(i32), i32, c_f, c_f, i64, i32, c_f, C.POINTER(c_f), c_f, c_f, c_f]),
    "tf_sample_ray_init": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, P(f32 * 6), c_f, c_f, i64, i32, f32, c_f, c_f, c_f, c_f]),
    "tf_sample_ray_upsample": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, i32, f32, c_f, f32, c_f, c_f, c_f, c_f]),
    "tf_sample_ray_merge": (C.c_int, [c_f, c_f, c_f, c_f, i64, i32, i32, c_f, c_f, c_f]),
    "tf_sample_ray_intervals": (C.c_int, [c_f, c_f, c_f, i64, i32, P(f32 * 6), c_f, c_f, c_f, c_f]),
    "tf_sample_points": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, f32, c_f, c_f, c_f, c_f, c_f, c_f]),
    "tf_alpha_mask_sample": (C.c_int, [c_f, i32, i32, i32, C.c_void_p, c_f, i64, c_f, c_f]),
    "tf_march_uniform": (C.c_int, [c_f, c_f, c_f, c_f, i64, i32, f32, C.c_void_p, c_f, i32, i32, i32, C.c_void_p, i32, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "tf_cubemap_mip_fwd": (C.c_int, [c_f, i32, c_f, c_f]),
    "tf_cubemap_diffuse_fwd": (C.c_int, [c_f, i32, c_f, c_f]),
    "tf_cubemap_diffuse_bwd": (C.c_int, [c_f, i32, c_f, c_f]),
    "tf_cubemap_texel_table": (C.c_int, [i32, c_f, c_f]),
    "tf_cubemap_specular_fwd": (C.c_int, [c_f, i32, f32, f32, c_f, c_f, c_f, c_f]),
    "tf_cubemap_specular_bwd": (C.c_int, [c_f, c_f, i32, f32, f32, c_f, c_f, c_f]),
    "tf_bvh_build_host": (i64, [C.c_void_p, i64, C.c_void_p, i64, C.c_void_p, C.c_void_p]),
    "tf_bvh_pack_host": (i64, [C.c_void_p, i64, C.c_void_p, i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tf_bvh_trace": (C.c_int, [c_f, c_f, P(f32 * 6), i64, c_f, c_f, i64, c_f, f32, f32, c_f, i64, c_f, c_f, c_f, c_f, i32, c_f, c_f, c_f]),
    "tf_inner_light_workspace_floats": (sz, []),
    "tf_inner_light_fwd": (C.c_int, [P(TfMlp4), c_f, c_f, c_f, i64, f32, i32, c_f, c_f, sz, c_f]),
    "tf_inner_light_indexed_fwd": (C.c_int, [P(TfMlp4), c_f, c_f, c_f, c_f, c_f, i64, c_f, f32, f32, i32, c_f, c_f, sz, c_f]),
    "tf_inner_light_indexed_train_fwd": (C.c_int, [P(TfMlp4), c_f, c_f, c_f, c_f, c_f, i64, c_f, f32, f32, i32, c_f, c_f, c_f, sz, c_f]),
    "tf_outer_light_indexed_fwd": (C.c_int, [P(TfMlp4), c_f, c_f, c_f, i64, f32, i32, c_f, c_f, sz, c_f]),
    "tf_compact_mask": (C.c_int, [c_f, i64, c_f, c_f, c_f]),
    "tf_compact_below": (C.c_int, [c_f, f32, i64, c_f, c_f, c_f]),
    "tf_shape_shade_workspace_floats": (sz, []),
    "tf_shape_shade_pack": (C.c_int, [P(TfShapeNets), c_f, sz, c_f]),
    "tf_shape_shade_fwd": (C.c_int, [c_f, C.c_void_p, C.c_void_p, i32, c_f, i32, c_f, i32, i32, f32, f32, f32, c_f, c_f, c_f, c_f, i64,
                                     c_f, c_f, c_f, c_f, c_f]),
    "tf_point_workspace_floats": (sz, []),
    "tf_point_pack": (C.c_int, [P(TfPointNets), c_f, sz, c_f]),
    "tf_point_fwd": (C.c_int, [c_f, P(TfVmDesc), c_f, P(TfVmDesc), c_f, P(TfVmDesc), c_f, P(f32 * 6), c_f, c_f, i64, f32,
                               c_f, c_f, c_f, c_f, c_f, c_f]),
    "tf_view_angles": (C.c_int, [c_f, c_f, i64, c_f, c_f]),
    "tf_shade_dirs": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, i32, c_f, c_f, i32, i64, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, c_f]),
    "tf_shade_dirs_whole": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, i32, c_f, c_f, i32, i64, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_f]),
    "tf_shade_dirs_fixed": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, i32, i64, c_f, c_f, c_f, c_f, c_f]),
    "tf_shade_dirs_fixed_mode": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, c_f, c_f, i32, i64, c_f, c_f, c_f, c_f, i32, c_f]),
    "tf_shade_dirs_bwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, i64, c_f, c_f, c_f, c_f]),
    "tf_shade_dirs_bwd_mode": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, i64, c_f, c_f, c_f, i32, c_f]),
    "tf_inner_light_encode": (C.c_int, [c_f, c_f, c_f, c_f, c_f, i64, c_f, i32, c_f, sz, c_f]),
    "tf_tv_partials": (C.c_int32, []),
    "tf_tv_fwd": (C.c_int, [c_f, i32, i32, i32, c_f, c_f]),
    "tf_tv_finish": (C.c_int, [c_f, f32, f32, c_f, c_f]),
    "tf_tv_bwd": (C.c_int, [c_f, i32, i32, i32, c_f, f32, f32, c_f, c_f]),
    "tf_ide5_fwd": (C.c_int, [c_f, c_f, c_f, i64, c_f, c_f]),
    "tf_ide5_bwd": (C.c_int, [c_f, c_f, c_f, c_f, i64, c_f, c_f, c_f]),
    "tf_posenc_fwd": (C.c_int, [c_f, i64, i32, i32, c_f, c_f]),
    "tf_linear_to_srgb_fwd": (C.c_int, [c_f, i64, i32, c_f, c_f]),
    "tf_linear_to_srgb_bwd": (C.c_int, [c_f, c_f, i64, i32, c_f, c_f]),
    "tf_shade_reduce": (C.c_int, [c_f, c_f, i64, i32, i32, c_f, c_f, c_f, c_f]),
    "tf_shade_reduce_aux": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, f32, c_f, i64, i32, i32, c_f, c_f, c_f, c_f, c_f, c_f]),
    "tf_shade_reduce_env": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, i32, f32, i64, i32, i32, c_f, c_f, c_f, c_f, c_f]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # The HIP runtime must be the one PyTorch-ROCm already loaded (its bundled libamdhip64): streams and
    # device pointers are only meaningful inside one runtime instance, so import torch FIRST.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C tensoflow_amd/csrc`). tensoflow_amd has no CPU / PyTorch fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().tf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
