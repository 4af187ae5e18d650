// fp32 MFMA building blocks for the tiny MLPs on the hot path (gfx950, wave64).
//
// All decoders on the path (flow coupling nets 44-64-64-64-21, inner-light net 123-256-256-256-3,
// SDF decoder 111-256-129) are evaluated TRANSPOSED: activations live as H^T tiles of
// [32 units x 32 rows], rows (samples) on the MFMA column/lane index.  With
// v_mfma_f32_32x32x2_f32 (exact fp32, D = A*B + C):
//     lane l supplies  A[i = l&31][k = l>>5]   and   B[k = l>>5][j = l&31]
//     lane l receives  D[i = rho(reg, l>>5)][j = l&31],  rho(reg,h) = (reg&3) + 8*(reg>>2) + 4*h
// so an accumulator register `reg` of unit-tile t IS the B operand of the next layer's k-step
// (t, reg) -- k = 32*t + rho(reg, h) for lane half h -- with no cross-lane movement and no LDS
// round trip between layers.  Weights are pre-packed in that fragment order ("wfrag"):
//     wfrag[((tout * ksteps) + s) * 64 + lane] = W[32*tout + (lane&31)][kmap(s, lane>>5)],
//     s = 16*t + reg,  kmap(s,h) = 32*(s>>4) + rho(s&15, h)
// which makes every A-operand fetch one contiguous 256-byte wave access (global or LDS).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int tf_rho(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
__host__ __device__ __forceinline__ int tf_kmap(int s, int h) {
  int reg = s & 15;
  return 32 * (s >> 4) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
}

__device__ __forceinline__ f32x16 tf_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Pack W [nout, ld] (columns col0 .. col0+kin-1 used) into fragment order; zero padded.
static __global__ void __launch_bounds__(256) tf_pack_wfrag_kernel(const float* __restrict__ W, int nout, int ld, int col0,
                                                            int kin, int tout_tiles, int ksteps,
                                                            float* __restrict__ dst, int s_major = 0) {
  int e = blockIdx.x * 256 + threadIdx.x;
  int total = tout_tiles * ksteps * 64;
  if (e >= total) return;
  int lane = e & 63;
  int s, tout;
  if (s_major) {  // [s][tout][lane]: one k-group of all unit tiles is contiguous (LDS-staged streaming)
    tout = (e >> 6) % tout_tiles;
    s = (e >> 6) / tout_tiles;
  } else {        // [tout][s][lane]
    s = (e >> 6) % ksteps;
    tout = (e >> 6) / ksteps;
  }
  int row = 32 * tout + (lane & 31);
  int k = tf_kmap(s, lane >> 5);
  dst[e] = (row < nout && k < kin) ? W[(long long)row * ld + col0 + k] : 0.f;
}

// Pack a bias vector [n] into accumulator order: dst[(tout*16 + reg)*2 + h] = b[32*tout + rho(reg,h)].
static __global__ void __launch_bounds__(256) tf_pack_bias_kernel(const float* __restrict__ b, int n, int tout_tiles,
                                                           float* __restrict__ dst) {
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= tout_tiles * 32) return;
  int h = e & 1, reg = (e >> 1) & 15, tout = e >> 5;
  int row = 32 * tout + (reg & 3) + 8 * (reg >> 2) + 4 * h;
  dst[e] = row < n ? b[row] : 0.f;
}

// One dense layer: out[TOUT] (+)= Wfrag * in[...]; KSTEPS k-steps taken from in[s>>4][s&15].
// `wf` points at this layer's fragment block (LDS or global), already offset by `lane`.
template <int KSTEPS, int TOUT, int TIN>
__device__ __forceinline__ void tf_layer(const float* __restrict__ wf, const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const float b = in[s >> 4][s & 15];
#pragma unroll
    for (int t = 0; t < TOUT; ++t) out[t] = tf_mfma(wf[(t * KSTEPS + s) * 64], b, out[t]);
  }
}

// Same, with a scheduling barrier every SB k-steps: bounds how far ahead the compiler hoists the
// (global-memory) A-operand loads, i.e. the registers they pin.
template <int KSTEPS, int TOUT, int TIN, int SB>
__device__ __forceinline__ void tf_layer_sb(const float* __restrict__ wf, const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
#pragma unroll
  for (int s0 = 0; s0 < KSTEPS; s0 += SB) {
#pragma unroll
    for (int s = s0; s < s0 + SB && s < KSTEPS; ++s) {
      const float b = in[s >> 4][s & 15];
#pragma unroll
      for (int t = 0; t < TOUT; ++t) out[t] = tf_mfma(wf[(t * KSTEPS + s) * 64], b, out[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Dense layer with the weights STREAMED through LDS by the whole 256-thread workgroup:
// wslab is s-major fragment order [KSTEPS][TOUT][64]; it is consumed in groups of SL k-steps
// (SL*TOUT*64 == 4096 floats = 16 KB), double buffered in `lds` (2 x 4096 floats).  Each thread moves
// 4 float4 per group global->register while the previous group's MFMAs run, then register->LDS.
// All four waves must call this with the same trip counts (it contains workgroup barriers).
typedef const __attribute__((address_space(1))) void* tf_gptr_t;
typedef __attribute__((address_space(3))) void* tf_lptr_t;

// One 16 KB weight slab global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, asynchronous):
// each of the 4 waves moves 4 pieces of 1 KB (lane l supplies the source address of its 16 bytes).
__device__ __forceinline__ void tf_slab_dma(const float* gthread /* slab + wave*1024 + lane*4 */, float* __restrict__ lbuf,
                                            int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_global_load_lds((tf_gptr_t)(gthread + i * 256), (tf_lptr_t)(lbuf + (wave * 4 + i) * 256), 16, 0, 0);
}

template <int KSTEPS, int TOUT, int TIN, int SL>
__device__ __forceinline__ void tf_layer_stream(const float* __restrict__ wslab, float* __restrict__ lds, int tid, int lane,
                                                const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
  static_assert(SL * TOUT * 64 == 4096, "one group must be 16 KB");
  static_assert(KSTEPS % SL == 0, "KSTEPS must be a multiple of the group size");
  constexpr int G = KSTEPS / SL;
  const int wave = tid >> 6;
  // ONE running per-thread source pointer, advanced by 16 KB per slab and made opaque each step: otherwise the
  // compiler materialises every slab address as its own loop-invariant 64-bit VGPR pair, hoists them out of the
  // tile loop and spills them (each reload then serialises the DMAs behind an s_waitcnt vmcnt(0)).
  const float* gp = wslab + wave * 1024 + lane * 4;
  asm volatile("" : "+v"(gp));
  tf_slab_dma(gp, lds, wave);
  __syncthreads();   // (the compiler drains vmcnt before the barrier: slab 0 has landed for every wave)
#pragma unroll
  for (int g = 0; g < G; ++g) {
    // slab g+1 streams into the other buffer while slab g feeds the MFMAs
    if (g + 1 < G) {
      gp += 4096;
      asm volatile("" : "+v"(gp));
      tf_slab_dma(gp, lds + ((g + 1) & 1) * 4096, wave);
    }
    const float* buf = lds + (g & 1) * 4096 + lane;
    // A operands are read from LDS one k-step AHEAD of the MFMAs that consume them (register double buffer),
    // so the LDS latency hides under the previous step's MFMAs instead of stalling every issue.
    float a_cur[TOUT], a_nxt[TOUT];
#pragma unroll
    for (int t = 0; t < TOUT; ++t) a_cur[t] = buf[t * 64];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int s = g * SL + sl;
      const float b = in[s >> 4][s & 15];
      if (sl + 1 < SL) {
#pragma unroll
        for (int t = 0; t < TOUT; ++t) a_nxt[t] = buf[((sl + 1) * TOUT + t) * 64];
      }
#pragma unroll
      for (int t = 0; t < TOUT; ++t) out[t] = tf_mfma(a_cur[t], b, out[t]);
#pragma unroll
      for (int t = 0; t < TOUT; ++t) a_cur[t] = a_nxt[t];
    }
    __syncthreads();
  }
}
