// Local tangent frame of a surface normal (get_orthogonal_directions + cross, network/fields.py:812-822, :829), shared by shade.hip and
// view_angles.hip.
#pragma once
#include <hip/hip_runtime.h>

struct Frame {
  float n[3], x[3], y[3];
};

__device__ __forceinline__ void normalize3(float* v) {  // F.normalize(dim=-1), eps 1e-12
  float inv = 1.f / fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
  // Round 5 traced a wrong tangent-frame choice (make_frame's l0 > l1) under a second stream's load to hipcc's packed form of the three
  // multiplies below reading the reciprocal in the instruction right behind the v_div_fixup_f32 that writes it (view_angles.hip).  The
  // root cause is unproven, so EVERY user of this function keeps an instruction of distance between the two: the reciprocal is made
  // opaque behind two wait states -- whatever the translation unit's vectoriser flags are (shade.hip's direction kernels build the
  // same frame and must agree with view_angles_kernel about it).
  asm volatile("s_nop 1" : "+v"(inv));
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

// get_orthogonal_directions + cross (fields.py:812-822, :829)
__device__ __forceinline__ void make_frame(const float* nin, Frame& F) {
  F.n[0] = nin[0]; F.n[1] = nin[1]; F.n[2] = nin[2];
  normalize3(F.n);
  const float o0[3] = {F.n[1], -F.n[0], 0.f};
  const float o1[3] = {-F.n[2], 0.f, F.n[0]};
  const float l0 = sqrtf(o0[0] * o0[0] + o0[1] * o0[1]), l1 = sqrtf(o1[0] * o1[0] + o1[2] * o1[2]);
  const bool use0 = l0 > l1;
  F.x[0] = use0 ? o0[0] : o1[0]; F.x[1] = use0 ? o0[1] : o1[1]; F.x[2] = use0 ? o0[2] : o1[2];
  normalize3(F.x);
  F.y[0] = F.n[1] * F.x[2] - F.n[2] * F.x[1];
  F.y[1] = F.n[2] * F.x[0] - F.n[0] * F.x[2];
  F.y[2] = F.n[0] * F.x[1] - F.n[1] * F.x[0];
}
