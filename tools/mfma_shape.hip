// Dev tool (round 6): does the MFMA SHAPE change what the matrix pipe sustains under load on this box?  MI355X_MICROARCH.md (7): on random
// data a v_mfma_f32_16x16x32 loop delivered ~1.15 x the FLOP/s of the 32x32x16 loop at equal cycles (the chip holds a higher clock).
// Same flop per iteration in both arms: 64 units x 64 rays x 32 k per "k-step pair" (32x32x16: 2 x 2 tiles x 2 k-steps = 8 MFMAs of 32
// cycles; 16x16x32: 4 x 4 tiles = 16 MFMAs of 16 cycles), x 3 product terms; operands re-read from LDS (ds_read_b128) every step,
// random f16 data.   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape.hip -o build_variants/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void __launch_bounds__(256) k(int iters, const h8* __restrict__ src, float* out, unsigned long long* cyc) {
  __shared__ h8 frag[4096];      // 64 KB of random fragments
  for (int i = threadIdx.x; i < 4096; i += 256) frag[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f16v acc32[4];
  f4v acc16[16];
  for (int n = 0; n < 4; ++n) for (int j = 0; j < 16; ++j) acc32[n][j] = 0.f;
  for (int n = 0; n < 16; ++n) for (int j = 0; j < 4; ++j) acc16[n][j] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const h8* f = frag + ((it & 7) * 512) + lane;
    if (SHAPE == 32) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        h8 a[2][2], b[2][2];      // [tile][hi|lo]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < 2; ++p) { a[t][p] = f[((s * 2 + t) * 2 + p) * 64]; b[t][p] = f[(((s + 2) * 2 + t) * 2 + p) * 64 % 512]; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            acc32[t * 2 + r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[r][0], acc32[t * 2 + r], 0, 0, 0);
            acc32[t * 2 + r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[r][1], acc32[t * 2 + r], 0, 0, 0);
            acc32[t * 2 + r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][1], b[r][0], acc32[t * 2 + r], 0, 0, 0);
          }
      }
    } else {
      h8 a[4][2], b[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) { a[t][p] = f[(t * 2 + p) * 64 % 512]; b[t][p] = f[((t + 4) * 2 + p) * 64 % 512]; }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc16[t * 4 + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][0], b[r][0], acc16[t * 4 + r], 0, 0, 0);
          acc16[t * 4 + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][0], b[r][1], acc16[t * 4 + r], 0, 0, 0);
          acc16[t * 4 + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][1], b[r][0], acc16[t * 4 + r], 0, 0, 0);
        }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int n = 0; n < 4; ++n) for (int j = 0; j < 16; ++j) s += acc32[n][j];
  for (int n = 0; n < 16; ++n) for (int j = 0; j < 4; ++j) s += acc16[n][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int SHAPE>
static double run(int cu, int iters, const h8* src, bool print) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * cu * 256); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<SHAPE><<<cu, 256>>>(iters, src, out, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double flop = (double)cu * 4 * iters * 3.0 * 2.0 * 64 * 64 * 32;
  if (print) printf("shape %2d: %8.3f ms  %7.1f TFLOP/s executed   %6.1f cycles per iteration (ideal 768)   clock %.0f MHz\n", SHAPE, ms, flop / ms * 1e-9,
                    (double)c / iters, (double)c / ms * 1e-3);
  hipFree(out); hipFree(cyc);
  return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cu = p.multiProcessorCount;
  _Float16* h = (_Float16*)malloc(4096 * 16);
  srand(1);
  for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  h8* src; hipMalloc(&src, 4096 * 16); hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
  run<32>(cu, 20000, src, false); run<16>(cu, 20000, src, false);
  for (int rep = 0; rep < 3; ++rep) { run<32>(cu, 400000, src, true); run<16>(cu, 400000, src, true); }
  return 0;
}
