// Dev tool (round 5): what can a SECOND wave on a SIMD do while the first keeps the matrix pipe busy?
// One workgroup per CU; waves 0..3 (one per SIMD) run role A, waves 4..7 role B (role -1 = absent).  Roles:
//   0  MFMA chain: v_mfma_f32_32x32x16_f16 on 8 independent accumulators, nothing else
//   1  VALU stream: independent v_fma_f32 on 16 registers
//   2  dependent global loads (pointer chase through an L2-resident table, one 16-byte load per step)
//   3  LDS reads (ds_read_b128, independent)
//   4  plain v_fma_f32 (inline asm: the compiler packs role 1's stream into v_pk_fma_f32, two passes per instruction)
//   5  integer VALU (v_add_u32 / v_xor_b32)
// Prints ns per iteration for A alone, B alone and A + B together: "sum" = the units serialise, "max" = they overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float role_mfma(int iters, int prio) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x % 7 + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f16v acc[8];
  for (int n = 0; n < 8; ++n) for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
  if (prio) __builtin_amdgcn_s_setprio(3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);     // 8 MFMAs = 256 cycles of the matrix pipe
  }
  float s = 0.f;
  for (int n = 0; n < 8; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
  return s;
}
__device__ __forceinline__ float role_valu(int iters) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 64; ++q) v[q % 16] = __builtin_fmaf(v[q % 16], 1.0001f, 0.5f);      // 64 vector instructions = 256 issue cycles of a lone wave
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  return s;
}
__device__ __forceinline__ float role_fma1(int iters) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  const float m = 1.0001f, a = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 64; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q % 16]) : "v"(m), "v"(a));
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  return s;
}
__device__ __forceinline__ float role_int(int iters) {
  unsigned v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 2654435761u + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 64; ++q) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(v[q % 16]) : "v"(0x9e3779b9u), "v"((unsigned)q));
  }
  unsigned s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  return (float)s;
}
__device__ __forceinline__ float role_chase(int iters, const uint4* table, unsigned mask) {
  unsigned p = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) & mask;
  float s = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { const uint4 r = table[p]; p = r.x & mask; s += __uint_as_float(r.y); }     // 4 dependent 16-byte loads
  }
  return s + p;
}
__device__ __forceinline__ float role_lds(int iters, const h8* frag) {
  h8 acc = frag[threadIdx.x & 63];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) { const h8 r = frag[((it + q) & 15) * 64 + (threadIdx.x & 63)]; acc = acc + r; }     // 16 ds_read_b128
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += (float)acc[i];
  return s;
}

template <int roleA, int roleB>      // (template arguments so that every pair is a kernel of its own name in a rocprofv3 trace)
__global__ void __launch_bounds__(512) k(int itersA, int itersB, const uint4* table, unsigned mask, int prio, float* out) {
  __shared__ h8 frag[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) for (int e = 0; e < 8; ++e) frag[i][e] = (_Float16)(0.001f * (i % 13 + e));
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB, iters = wave < 4 ? itersA : itersB;
  float s = 0.f;
  if (role == 0) s = role_mfma(iters, prio);
  else if (role == 1) s = role_valu(iters);
  else if (role == 2) s = role_chase(iters, table, mask);
  else if (role == 3) s = role_lds(iters, frag);
  else if (role == 4) s = role_fma1(iters);
  else if (role == 5) s = role_int(iters);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(int, int, const uint4*, unsigned, int, float*);
template <int A> static kern_t pickB(int b) {
  switch (b) { case -1: return k<A, -1>; case 0: return k<A, 0>; case 1: return k<A, 1>; case 2: return k<A, 2>; case 3: return k<A, 3>; case 4: return k<A, 4>; default: return k<A, 5>; }
}
static kern_t pick(int a, int b) {
  switch (a) { case 0: return pickB<0>(b); case 1: return pickB<1>(b); case 2: return pickB<2>(b); case 3: return pickB<3>(b); case 4: return pickB<4>(b); default: return pickB<5>(b); }
}
static float run(int cu, int roleA, int roleB, int itA, int itB, const uint4* table, unsigned mask, float* out, int prio = 0) {
  const int threads = roleB < 0 ? 256 : 512;
  kern_t kk = pick(roleA, roleB);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kk, dim3(cu), dim3(threads), 0, 0, itA / 10, itB / 10, table, mask, prio, out); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(kk, dim3(cu), dim3(threads), 0, 0, itA, itB, table, mask, prio, out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cu = p.multiProcessorCount;
  const unsigned n = 1u << 17;       // 2 MB table: L2 resident, far beyond the 32 KB vector L1
  std::vector<uint4> h(n);
  unsigned x = 12345u;
  for (unsigned i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = make_uint4(x >> 8, i, 0, 0); }
  uint4* table; hipMalloc(&table, n * sizeof(uint4)); hipMemcpy(table, h.data(), n * sizeof(uint4), hipMemcpyHostToDevice);
  float* out; hipMalloc(&out, sizeof(float) * cu * 512);
  const char* names[6] = {"mfma", "valu", "chase", "lds", "fma1", "int"};
  const int iters[6] = {20000, 20000, 2000, 20000, 20000, 20000};
  float alone[6];
  for (int r = 0; r < 6; ++r) {
    alone[r] = run(cu, r, -1, iters[r], 0, table, n - 1, out);
    printf("%-6s alone (1 wave/SIMD): %8.3f ms = %7.1f ns per iteration\n", names[r], alone[r], alone[r] * 1e6 / iters[r]);
  }
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) {
      if (a >= 1 && b >= 4 && !(a == b)) continue;      // (the extra vector roles are paired with the matrix role and with themselves)
      // scale B's iteration count so that both roles last about as long alone
      const int itB = (int)(iters[b] * (alone[a] / alone[b]));
      const float ms = run(cu, a, b, iters[a], itB, table, n - 1, out);
      printf("%-6s + %-6s: %8.3f ms together; alone %8.3f each -> %.2f x (1.00 = full overlap, 2.00 = serialised)\n", names[a], names[b], ms, alone[a], ms / alone[a]);
    }
  {
    const int itB = (int)(iters[1] * (alone[0] / alone[1]));
    const float ms = run(cu, 0, 1, iters[0], itB, table, n - 1, out, 1);
    printf("mfma (s_setprio 3) + valu: %8.3f ms -> %.2f x\n", ms, ms / alone[0]);
  }
  return 0;
}
