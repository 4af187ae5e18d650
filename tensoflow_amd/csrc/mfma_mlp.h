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
template <int KSTEPS, int TOUT, int TIN, int SL>
__device__ __forceinline__ void tf_layer_stream(const float* __restrict__ wslab, float* __restrict__ lds, int tid, int lane,
                                                const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
  static_assert(SL * TOUT * 64 == 4096, "one group must be 16 KB");
  static_assert(KSTEPS % SL == 0, "KSTEPS must be a multiple of the group size");
  constexpr int G = KSTEPS / SL;
  float4 st[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) st[i] = reinterpret_cast<const float4*>(wslab)[tid + 256 * i];
#pragma unroll
  for (int i = 0; i < 4; ++i) reinterpret_cast<float4*>(lds)[tid + 256 * i] = st[i];
  __syncthreads();
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g + 1 < G) {
#pragma unroll
      for (int i = 0; i < 4; ++i) st[i] = reinterpret_cast<const float4*>(wslab + (g + 1) * 4096)[tid + 256 * i];
    }
    const float* buf = lds + (g & 1) * 4096 + lane;
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int s = g * SL + sl;
      const float b = in[s >> 4][s & 15];
#pragma unroll
      for (int t = 0; t < TOUT; ++t) out[t] = tf_mfma(buf[(sl * TOUT + t) * 64], b, out[t]);
    }
    if (g + 1 < G) {
#pragma unroll
      for (int i = 0; i < 4; ++i) reinterpret_cast<float4*>(lds + ((g + 1) & 1) * 4096)[tid + 256 * i] = st[i];
    }
    __syncthreads();
  }
}
