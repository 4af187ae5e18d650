// Dev tool: does a second wave on a SIMD hide one wave's vector work behind the other's MFMAs?  Each wave runs `iters` x
// [24 v_mfma_f32_32x32x16_f16 (3 per accumulator, 8 accumulators) + NV dependent-free vector instructions]; launched with one and
// with two waves per SIMD.  Prints time per iteration per SIMD; MFMA-only floor = 24 x 32 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NV, bool BAR = false, int LDSR = 0>
__global__ void __launch_bounds__(512) k(int iters, float* out) {
  __shared__ h8 frag[4 * 1024];   // 64 KB: a 4-slab ring's worth of 16-byte fragments
  for (int i = threadIdx.x; i < 4 * 1024; i += blockDim.x) for (int e = 0; e < 8; ++e) frag[i][e] = (_Float16)(0.001f * (i % 13 + e));
  __syncthreads();
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x % 7 + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f16v acc[8];
  for (int n = 0; n < 8; ++n) for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  h8 fa[16], fb[16];
  for (int q = 0; q < 16; ++q) { fa[q] = a; fb[q] = a; }
  auto body = [&](const h8 (&cur)[16], h8 (&nxt)[16], int slot) {
    if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      if (LDSR == 1 && n >= 1 && n <= 4) {
#pragma unroll
        for (int q = 4 * (n - 1); q < 4 * n; ++q) nxt[q] = frag[slot * 1024 + q * 64 + (threadIdx.x & 63)];   // next step's fragments
      }
      if (LDSR == 2) {   // two reads per MFMA group, all eight groups
#pragma unroll
        for (int q = 2 * n; q < 2 * n + 2; ++q) nxt[q] = frag[slot * 1024 + q * 64 + (threadIdx.x & 63)];
      }
      const h8 ah = LDSR ? cur[2 * n] : a, al = LDSR ? cur[2 * n + 1] : a;
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc[n], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV / 8; ++q) v[q % 8] = __builtin_fmaf(v[q % 8], 1.0001f, 0.5f);   // NV/8 vector instructions per group
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int it = 0; it < iters; it += 2) {
    body(fa, fb, it & 3);
    body(fb, fa, (it + 1) & 3);
  }
  float s = 0.f;
  for (int n = 0; n < 8; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool BAR = false, int LDSR = 0>
static void run(int cu, int threads, int iters) {
  float* out; hipMalloc(&out, sizeof(float) * cu * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV, BAR, LDSR><<<cu, threads>>>(iters / 10, out); hipDeviceSynchronize();
  hipEventRecord(e0); k<NV, BAR, LDSR><<<cu, threads>>>(iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int waves_per_simd = threads / 256;
  printf("bar=%d lds=%d NV=%3d vector instr per 24 MFMAs, %d wave(s)/SIMD: %7.3f ms  -> %6.0f ns per (wave-iteration), %6.0f ns per SIMD-iteration-pair\n",
         (int)BAR, LDSR, NV, waves_per_simd, ms, ms * 1e6 / iters, ms * 1e6 / iters / waves_per_simd);
  hipFree(out);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cu = p.multiProcessorCount, iters = 20000;
  run<0>(cu, 256, iters); run<0>(cu, 512, iters);
  run<24>(cu, 256, iters); run<24>(cu, 512, iters);
  run<64>(cu, 256, iters); run<64>(cu, 512, iters);
  run<128>(cu, 256, iters); run<128>(cu, 512, iters);
  run<64, true, 0>(cu, 256, iters);      // + one s_barrier per 24 MFMAs (4 waves)
  run<64, false, 1>(cu, 256, iters);     // + 16 ds_read_b128 per 24 MFMAs
  run<64, true, 1>(cu, 256, iters);      // both
  run<64, true, 1>(cu, 512, iters);
  run<64, true, 2>(cu, 256, iters);      // reads spread 2 per group
  run<0, true, 2>(cu, 256, iters);
  run<0, true, 1>(cu, 256, iters);
  return 0;
}
