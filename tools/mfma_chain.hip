// Dev microbenchmark: what does a 32x32x16 f16 MFMA stream cost when only NACC accumulators are in flight (chains of 3 dependent MFMAs
// per accumulator, as in a f16x3 layer with NACC output tiles) and NV independent vector instructions follow two of every three MFMAs?
// Prints shader cycles per MFMA (s_memtime) for 1, 2 and 3 waves per SIMD.  Floor: 32 cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, int NV>
__global__ void __launch_bounds__(768) k(int iters, float* out, unsigned long long* ticks) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x % 7 + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f16v acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 24 / (3 * NACC); ++g)
#pragma unroll
      for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
          if (c < 2) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q % 8] = __builtin_fmaf(v[q % 8], 1.0001f, 0.5f);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NACC, int NV>
static void run(int cu, int iters) {
  float* out; unsigned long long* tk; hipMalloc(&out, sizeof(float) * cu * 768); hipMalloc(&tk, 8);
  for (int waves = 1; waves <= 3; ++waves) {
    k<NACC, NV><<<cu, 256 * waves>>>(iters, out, tk); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<NACC, NV><<<cu, 256 * waves>>>(iters, out, tk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
    printf("accumulators %2d, %d vector instr behind 2 of 3 MFMAs, %d wave(s)/SIMD: %6.1f cycles per MFMA of one wave, %6.1f per MFMA per SIMD  (%.3f ms)\n", NACC, NV, waves,
           (double)t / (24.0 * iters), (double)t / (24.0 * iters) / waves, ms);
  }
  hipFree(out); hipFree(tk);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cu = p.multiProcessorCount, iters = 4000;
  run<2, 0>(cu, iters); run<2, 4>(cu, iters); run<2, 7>(cu, iters); run<2, 10>(cu, iters);
  run<8, 0>(cu, iters); run<8, 4>(cu, iters); run<8, 7>(cu, iters); run<8, 10>(cu, iters);
  run<1, 0>(cu, iters); run<1, 7>(cu, iters);
  return 0;
}
