// direction_to_angle(normals, view) / (2 pi, pi / 2) (network/fields.py:1035-1048, :1077-1079): the view direction's azimuth and polar angle
// in the local frame of the normal, the first kernel of every shade() call.
//
// This kernel is a translation unit of its own because it is BUILT WITHOUT SLP VECTORISATION (Makefile: -fno-slp-vectorize): round 5
// found that with two shade() calls in flight on two HIP streams a few lanes of this kernel took the OTHER tangent-frame candidate
// (make_frame's l0 > l1 came out the other way round) although its inputs were bit-identical and the same rows recomputed alone were
// right -- only when another kernel's waves were resident on the GPU, only with hipcc's packed fp32 instructions in it (a
// v_pk_mul_f32 reads the register a v_div_fixup_f32 wrote in the instruction in front of it), never with the scalar form
// (tools/exp_streams3.py).  With the scalar form two calls in flight are bit-identical to the serial loop.
#include "shade_frame.h"
#include "tf_common.h"

static constexpr float kTwoPi = 6.28318530717958647692f;
static constexpr float kHalfPi_ = 1.57079632679489661923f;
static constexpr float kEPS = 1e-6f;
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// direction_to_angle(normals, view)/(2pi, pi/2)  (fields.py:1035-1048, :1077-1079)
__global__ void __launch_bounds__(256) view_angles_kernel(const float* __restrict__ normals, const float* __restrict__ view,
                                                          long long pn, float* __restrict__ va) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pn) return;
  Frame F;
  make_frame(normals + 3 * i, F);
  float v[3] = {view[3 * i], view[3 * i + 1], view[3 * i + 2]};
  normalize3(v);
  const float cx = dot3(F.x, v), cy = dot3(F.y, v);
  const float cz = fminf(fmaxf(dot3(F.n, v), -1.f + kEPS), 1.f - kEPS);
  const float phi = fmodf(atan2f(cy, cx) + kTwoPi, kTwoPi);
  va[2 * i] = phi / kTwoPi;
  va[2 * i + 1] = acosf(cz) / kHalfPi_;
}


extern "C" int tf_view_angles(const float* normals, const float* view, int64_t pn, float* view_angles, tf_stream_t stream) {
  TF_REQUIRE(pn >= 0, TF_ESHAPE, "tf_view_angles: pn < 0");
  if (pn == 0) return TF_OK;
  TF_REQUIRE(normals && view && view_angles, TF_EINVAL, "tf_view_angles: null pointer");
  view_angles_kernel<<<tf_blocks(pn, 256), 256, 0, (hipStream_t)stream>>>(normals, view, pn, view_angles);
  TF_LAUNCH_CHECK("tf_view_angles");
  return TF_OK;
}

