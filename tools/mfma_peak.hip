// Dev tool: what does the matrix pipe of THIS box sustain?  Pure v_mfma_f32_32x32x16_f16 loops (no memory traffic), 1 and 2
// waves per SIMD, independent accumulators vs one dependent chain; prints TFLOP/s and the shader clock implied by
// s_memtime / wall time.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(int iters, float* out, unsigned long long* cyc) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f16v acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
static void run(const char* name, int blocks, int threads, int iters) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, threads>>>(iters / 10, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, threads>>>(iters, out, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double waves = (double)blocks * threads / 64, mf = waves * iters * 8.0;
  printf("%-44s %8.3f ms  %8.1f TFLOP/s   counter ticks/MFMA %.1f  (counter %.0f MHz-equivalent)\n", name, ms, mf * 32768.0 / ms * 1e-9,
         (double)c / (iters * 8.0), (double)c / ms * 1e-3);
  hipFree(out); hipFree(cyc);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s  CUs %d  clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  const int cu = p.multiProcessorCount;
  run<8>("1 wave/SIMD, 8 independent accumulators", cu, 256, 200000);
  run<1>("1 wave/SIMD, 1 accumulator (dependent chain)", cu, 256, 200000);
  run<8>("2 waves/SIMD, 8 independent accumulators", cu * 2, 256, 200000);
  run<4>("1 wave/SIMD, 4 accumulators", cu, 256, 200000);
  return 0;
}
