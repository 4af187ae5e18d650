// Dev tool: what does the matrix pipe sustain in EXACT fp32 (v_mfma_f32_32x32x2_f32: the arithmetic of the training direction's dense
// layers)?  Pure MFMA loops, no memory traffic.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_f32.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) loop(int iters, float* out) {
  float a = 0.001f * threadIdx.x, b = 0.002f;
  f16v acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
static void run(const char* name, int blocks, int iters) {
  float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  loop<NACC><<<blocks, 256>>>(iters / 10, out); hipDeviceSynchronize();
  hipEventRecord(e0); loop<NACC><<<blocks, 256>>>(iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mf = (double)blocks * 4 * iters * 8.0;
  printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, mf * 4096.0 / ms * 1e-9);
  hipFree(out);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s  CUs %d\n", p.name, p.multiProcessorCount);
  const int cu = p.multiProcessorCount;
  run<4>("1 wave/SIMD, 4 independent accumulators", cu, 100000);
  run<4>("2 waves/SIMD, 4 independent accumulators", cu * 2, 100000);
  run<4>("4 waves/SIMD, 4 independent accumulators", cu * 4, 100000);
  run<1>("1 wave/SIMD, dependent chain", cu, 100000);
  return 0;
}
