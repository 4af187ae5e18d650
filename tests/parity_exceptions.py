"""Documented exceptions to the 1e-4 RELATIVE bar of conftest.parity (max |a-b| / max(|b|, 1e-3 max|b|)): quantities whose golden is
the reference's fp32 value of a DIFFERENCE or a saturating function of O(1) terms, so that elements far below the array's scale carry
the terms' ~1e-7 rounding as a large relative error.  Every entry: label pattern, the bound held, the measured value (1 x MI355X,
round 5, f16x3 and exact-fp32 decoders alike) and the cause.  The absolute-floor measure (max |a-b| / max(|b|, 1)) is held to 1e-4 on
all of them regardless.  No bound is wider than 4 x the value measured for it (round 6: a bound 20 x its measurement pins nothing).  Where the oracle can evaluate the function in fp64 the tests pass `truth=` instead and need no entry here
(alpha / gradient / hessian / sdf / features / flow densities / spline values: tests/test_gpu_parity.py) -- those tests are the
evidence for the first block below: the reference's OWN fp32 alpha is up to 2.5e-3 (relative, same measure) from its fp64 value, its
finite-difference gradient 5.0e-3, its hessian term 3.8e-2."""
import re

_FD = ("the same quantity as tests/test_gpu_parity.py::test_sdf_alpha_golden, where the oracle's fp64 evaluation shows the reference's own "
       "fp32 value this far from the function (alpha 2.5e-3, gradient 5.0e-3, hessian term 3.8e-2: differences of O(1) decoder outputs)")
TABLE = [
    # ---- compute_sdf_alpha through the renderer modules
    (r"compute_sdf_alpha alpha .*", 1e-2, "measured 2.8e-3 at 3.2e-7 absolute; " + _FD),
    (r"module_tensosdf:195\.0", 1e-2, "measured 2.7e-3 at 1.4e-5 absolute (TensoSDF.gradient); " + _FD),
    (r"module_tensosdf:195\.1", 2e-2, "measured 5.1e-3 at 6.6e-4 absolute (normal_hessian: second differences / eps^2); " + _FD),
    (r"sdf_alpha_training_golden:675", 8e-3, "measured 2.2e-3 at 1.7e-7 absolute; " + _FD),
    (r"render_core(\(train\))? gradient_error", 4e-3, "measured 1.2e-3 at 6.0e-6 absolute: (|gradient| - 1)^2 of the finite-difference gradient; " + _FD),
    (r"render_core(\(train\))? acc", 4e-4, "measured 1.17e-4 at 6.3e-7 absolute: opacity = sum of alpha-composited weights (alphas as above)"),
    (r"render_core\(late training\) normal", 5e-4, "measured 1.28e-4 at 1.1e-6 absolute: weighted sum of finite-difference gradients; " + _FD),
    # ---- normals from a NORMALISED finite-difference gradient at the surface (|gradient| ~ 1e-3-sized sdf differences / 2 eps)
    (r"material nvs refined normals|material nvs frame normal", 2e-3, "measured 4.8e-4 / 5.3e-4 at 3.5e-6 absolute: normalize(FD gradient) at the refined hit"),
    (r"trace_sdf_with_mesh:327", 4e-3, "measured 1.1e-3 at 4.6e-6 absolute: normalize(FD gradient) of ~1e-3-sized sdf differences"),
    # ---- sigmoid outputs far below 1 (occlusion probabilities, scale 0.13, elements down to 1e-4): logit rounding ~1e-6
    (r"shape shading (fused|composed) shade_occ_prob|shape_shading:114|shape_shading_ragged_and_degenerate:138\.1", 5e-3,
     "measured 1.55e-3 / 1.39e-3 / 9.2e-4 at 5e-7 absolute: sigmoid of a 128-wide net's output, the element is 3e-4"),
    (r"shape shading variants (occ|inter) (occ_prob|indirect_light)", 1e-3,
     "measured 5.07e-4 at 8.9e-8 absolute with the bf16 triple split of round 5 (1.69e-4 at 1.2e-7 with the exact-fp32 product instruction): "
     "either way ONE ulp of the sigmoid's O(1) operand, on elements of 1e-4"),
    (r"render_core\(validation\) (occ_prob|occ_prob_gt|indirect_light)", 2e-3,
     "measured 4.8e-4 / 7.1e-4 at 1.4e-6 / 4.7e-5 absolute: occlusion at the expected-depth point; the traced value resamples 128 sdf evaluations by inverse CDF"),
    (r"shape shading (fused|composed) shade_reflective|shape_shading:116", 9e-4, "measured 2.4e-4 at 6.0e-7 absolute: reflect(view, normal), a difference of unit vectors, components down to 1e-3"),
    (r"update_alpha_mask_golden:311", 4.5e-4, "measured 1.19e-4 at 1.19e-7 absolute: ONE ulp of 1 - exp(-x) at x ~ 1e-3"),
    (r"MCShadingNetwork training outputs specular_color", 4.5e-4, "measured 1.14e-4 at 4.5e-5 absolute: sRGB of a Monte-Carlo mean over flow samples, one of which sits at an ill-conditioned spline root (tests/test_oracle_flow.py)"),
    (r"pwquad forward log-Jacobian", 1e-3, "measured 3.4e-4 at 2.8e-5 absolute on 2 of 2 048 rows of the reference's EDGE-CASE vectors (w_tilde = -12: alpha = (x - wsum) / w divides by a bin width of e^-12)"),
]


def lookup(label):
    for pat, bound, why in TABLE:
        if re.fullmatch(pat, label or ""):
            return bound, why
    return None
