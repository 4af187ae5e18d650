// Dev probe: does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs on gfx950?  And does the hi/lo split produce them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  h8 a, b;
  const _Float16 sub = (_Float16)3.0e-6f;        // f16 subnormal (min normal 6.1e-5)
  const _Float16 one = (_Float16)1.0f;
  for (int e = 0; e < 8; ++e) { a[e] = (lane < 32 && e == 0 && (lane & 31) == 0) ? sub : (_Float16)0.f; b[e] = (e == 0 && lane < 32) ? one : (_Float16)0.f; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) { out[0] = c[0]; out[1] = (float)sub; }
  // B-side subnormal
  for (int e = 0; e < 8; ++e) { a[e] = (e == 0 && lane < 32) ? one : (_Float16)0.f; b[e] = (lane < 32 && e == 0) ? sub : (_Float16)0.f; }
  f16v d = {0};
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
  if (lane == 0) out[2] = d[0];
  // split of x = 0.05 (lo is subnormal)
  float x = 0.05f + 1e-9f * lane;
  _Float16 hi = (_Float16)x;
  _Float16 lo = (_Float16)(x - (float)hi);
  if (lane == 0) { out[3] = x; out[4] = (float)hi; out[5] = (float)lo; out[6] = x - (float)hi; }
}
int main() {
  float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  probe<<<1, 64>>>(d);
  float h[16]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("A-side subnormal %.9g x 1 -> mfma %.9g ; B-side -> %.9g\n", h[1], h[0], h[2]);
  printf("split x=%.9g hi=%.9g lo=%.9g exact residual=%.9g\n", h[3], h[4], h[5], h[6]);
  return 0;
}
