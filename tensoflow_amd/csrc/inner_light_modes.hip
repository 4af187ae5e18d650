// The two operand modes of the inner-light net (MCShadingNetwork.get_inner_lights, network/fields.py:905-911) that do NOT run on the
// staggered kernel of inner_light.hip:
//   TF_PREC_F32  exact fp32 MFMA (v_mfma_f32_32x32x2_f32; 157 TF/s peak): the slab-ring kernel, 4 waves x 32 rays in lockstep, weight
//                slabs streamed L2 -> LDS by LDS-DMA.  The yardstick the parity tests hold the faster modes against.
//   TF_PREC_F16  plain f16 operands (weights AND activations rounded: narrower than the reference, opt-in, BASELINE configs[4]): the
//                column-owned kernel.
// TF_PREC_F16X3 / TF_PREC_F16X2 -> inner_light3_kernel (inner_light.hip).
#include <cmath>
#include <type_traits>

#include "inner_light_ws.h"
#include "mfma_mlp.h"
#include "tf_common.h"

__device__ __forceinline__ float relu(float x) { return tf_relu(x); }


template <int KSTEPS, int TIN>
__device__ __forceinline__ void hidden_layer(const float* __restrict__ wslab, const float* __restrict__ bias,
                                             float* __restrict__ lds, int tid, int lane, int h,
                                             const f32x16 (&in)[TIN], f32x16 (&out)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) out[t][j] = bias[(t * 16 + j) * 2 + h];
  tf_layer_stream<KSTEPS, 8, TIN, 8, 3>(wslab, lds, tid, lane, in, out);
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) out[t][j] = relu(out[t][j]);
}

// Dense mode (idx == nullptr): row r reads pts/view/nrm[r] and writes out[r].
// Indexed mode: row r stands for ray i = idx[r], r < *count_dev (device-side count: no host sync between the BVH
// trace and this kernel); view = -view[i] (the ray direction is passed), out[i] = light * (depth[i] > near_eps).
// Exact fp32 MFMA (TF_PREC_F32).
__global__ void __launch_bounds__(256) inner_light_kernel(const float* __restrict__ ws_arg, const float* __restrict__ pts,
                                                          const float* __restrict__ view, const float* __restrict__ nrm,
                                                          long long m_arg, const long long* __restrict__ idx,
                                                          const long long* __restrict__ count_dev,
                                                          const float* __restrict__ depth, float near_eps, float exp_max,
                                                          float* __restrict__ out) {
  long long m = m_arg;
  if (count_dev) m = min(m_arg, *count_dev);
  if (m <= 0) return;
  __shared__ __attribute__((aligned(16))) float lds[4 * 4096];   // weight-slab ring
  const int tid = threadIdx.x, lane = threadIdx.x & 63, h = lane >> 5;
  const long long n_groups = (m + 127) / 128;   // a workgroup advances 4 tiles (128 rays) in lockstep
  for (long long tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
    // opaque per-iteration copy of the workspace base: biases / IDE table / slab addresses are loop-invariant and
    // would otherwise be hoisted out of the tile loop, spilled, and reloaded behind s_waitcnt vmcnt(0)
    const float* ws = ws_arg;
    asm volatile("" : "+s"(ws));
    const long long tile = tg * 4 + (threadIdx.x >> 6);
    long long row = tile * 32 + (lane & 31);
    const bool valid = row < m;
    if (!valid) row = m - 1;
    const long long src = idx ? idx[row] : row;
    const float vsign = idx ? -1.f : 1.f;
    // ---- encodings (each lane computes all 123 and keeps the half its MFMA operand slots need)
    float enc[128];
    const float p[3] = {pts[3 * src], pts[3 * src + 1], pts[3 * src + 2]};
    {
#pragma unroll
    for (int k = 0; k < 3; ++k) enc[k] = p[k];
    // arguments p * 2^f, f < 8: inside |p| < 3 (any scene in the unit sphere) the fp32 three-constant reduction is exact
    // (|k| < 2^8); the double-precision reduction (24 x ~8 f64 instructions per tile) is kept for out-of-range callers only
    if (__all(fabsf(p[0]) < 3.f && fabsf(p[1]) < 3.f && fabsf(p[2]) < 3.f)) {
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int k = 0; k < 3; ++k) tf_sincos_small(p[k] * (float)(1 << f), enc[3 + 6 * f + k], enc[3 + 6 * f + 3 + k]);
    } else {
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int k = 0; k < 3; ++k) tf_sincos(p[k] * (float)(1 << f), enc[3 + 6 * f + k], enc[3 + 6 * f + 3 + k]);
    }
    float n[3] = {nrm[3 * src], nrm[3 * src + 1], nrm[3 * src + 2]};
    float v[3] = {vsign * view[3 * src], vsign * view[3 * src + 1], vsign * view[3 * src + 2]};
    float inv = 1.f / fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);
    n[0] *= inv; n[1] *= inv; n[2] *= inv;
    inv = 1.f / fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
    const float vn = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
    const float rx = vn * n[0] * 2.f - v[0], ry = vn * n[1] * 2.f - v[1], rz = vn * n[2] * 2.f - v[2];
    {
      // IDE: sph[i] = (rx + i ry)^m_i * sum_k mat[k][i] rz^k ; output = [Re(36) | Im(36)]
      float zp[17];
      zp[0] = 1.f;
#pragma unroll
      for (int k = 1; k < 17; ++k) zp[k] = zp[k - 1] * rz;
      float cre[17], cim[17];
      cre[0] = 1.f; cim[0] = 0.f;
#pragma unroll
      for (int k = 1; k < 17; ++k) {
        cre[k] = cre[k - 1] * rx - cim[k - 1] * ry;
        cim[k] = cre[k - 1] * ry + cim[k - 1] * rx;
      }
      // the 17 x 36 polynomial table is wave-uniform: read it through the scalar cache (s_load) -- as a generic pointer the reads
      // were 44 flat_load_dwordx4 per tile, each tile start waiting on vmcnt(0) behind the weight DMAs in flight
      const __attribute__((address_space(4))) float* mat =
          (const __attribute__((address_space(4))) float*)(unsigned long long)(ws + kIdeMat);
      // column of (d, mm) = (2^d - 1) + d + mm: every index below is a compile-time constant once the loops are unrolled
      // (a running `col++` counter left enc[] dynamically indexed, i.e. in scratch memory)
#pragma unroll
      for (int d = 0; d < 5; ++d) {
#pragma unroll
        for (int mm = 0; mm <= (1 << d); ++mm) {
          const int col = (1 << d) - 1 + d + mm;
          float poly = 0.f;
#pragma unroll
          for (int k = 0; k <= (1 << d) - mm; ++k) poly += zp[k] * mat[k * 36 + col];
          enc[51 + col] = cre[mm] * poly;
          enc[51 + 36 + col] = cim[mm] * poly;
        }
      }
    }
#pragma unroll
    for (int k = 123; k < 128; ++k) enc[k] = 0.f;
    }
    f32x16 a[8], b[8];
    {
      const unsigned long long upper_half = 0xFFFFFFFF00000000ULL;   // lanes 32..63 (h = 1)
      f32x16 in1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k0 = 32 * t + (j & 3) + 8 * (j >> 2);
          // explicit v_cndmask: written as `h ? enc[k0 + 4] : enc[k0]` the compiler folds the select into the ADDRESS
          // and keeps enc[] as a lane-indexed array in scratch memory (128 stores + 16 loads per ray)
          asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(in1[t][j]) : "v"(enc[k0]), "v"(enc[k0 + 4]), "s"(upper_half));
        }
      hidden_layer<64, 4>(ws + kI1, ws + kIB1, lds, tid, lane, h, in1, a);
    }
    hidden_layer<128, 8>(ws + kI2, ws + kIB2, lds, tid, lane, h, a, b);
    hidden_layer<128, 8>(ws + kI3, ws + kIB3, lds, tid, lane, h, b, a);
    f32x16 o[1];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[0][j] = ws[kIB4 + j * 2 + h];
    tf_layer_stream<128, 1, 8, 64, 3>(ws + kI4, lds, tid, lane, a, o);
    if (valid && h == 0) {
      const float near = (depth && !(depth[src] > near_eps)) ? 0.f : 1.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) out[3 * src + c] = expf(fminf(o[0][c], exp_max)) * near;
    }
  }
}


// =====================================================================================================================
// Column-owned form (f16x3 / f16x2 / f16 arithmetic).  The kernel above walks the whole weight image once per 128 rays through
// an LDS ring: 42 slab steps per pass, each with a workgroup barrier, four LDS-DMA pieces and 16 fragment reads per wave in
// front of 24 MFMAs -- the matrix pipe is busy 57 % of the time and every weight byte crosses the LDS once per 32 rays.
// Here the roles of weights and activations are swapped:
//   * a wave OWNS 64 of a layer's 256 output units (two 32-unit tiles) for all R = 128 rays of the pass (four 32-ray
//     tiles): its weight fragments come straight from L2 into registers (1 KB coalesced wave loads, nobody else in the
//     workgroup needs them) and each one feeds 4 ray tiles x TERMS MFMAs;
//   * the ACTIVATIONS of a layer live in LDS as f16 MFMA B-fragments ([k-step][ray tile][hi|lo][lane][8 halves], written
//     by the wave that produced them straight from its accumulator registers -- the unit permutation folded into the weight
//     packing makes an accumulator lane's 8 registers one 16-byte B-fragment), every wave reads all of them;
//   * two barriers per layer (inputs read / outputs written) instead of one per 16 KB of weights; per k-step a wave issues 2-4
//     global loads + 4-8 ds_read_b128 for 8-24 MFMAs.
// TERMS = 3: weights and activations split hi + lo (fp32-grade products, 128 KB of LDS, one workgroup per CU).
// TERMS = 2: weights split, activations rounded to f16 once per layer.  TERMS = 1: plain f16 operands.  (64 KB of LDS: two
// workgroups per CU -- one computes its encodings / epilogues under the other's MFMAs.)
// The input row is cat[IDE (72), pos_enc8 (51), 0 (5)] -- IDE first so that no 4-value store granule straddles the two encoders
// (waves 0-1 compute the positional encoding of the pass's 128 rays, waves 2-3 the IDE); layer 1's weight columns are packed in
// that order (kP1).
template <int TERMS>
struct IL2 {
  static constexpr int R = 128, RT = 4;
  static constexpr int XP = TERMS == 3 ? 2 : 1;     // activation planes in LDS (hi | lo)
  static constexpr int AP = TERMS >= 2 ? 2 : 1;     // weight planes fetched
  static constexpr int ACT16 = 16 * RT * XP * 64;   // 16-byte units
};

// 8-byte slot of feature k (k % 4 == 0) of ray (r, j) in the layer-1 B-fragment image; `plane` 0 = hi, 1 = lo
template <int XP>
__device__ __forceinline__ int il2_slot8(int k, int r, int j, int plane) {
  const int s = 2 * (k >> 5) + ((k >> 4) & 1), c = (k >> 3) & 1, h = (k >> 2) & 1;
  return ((((s * 4 + r) * XP + plane) * 64 + j + 32 * h) << 1) + c;   // in 8-byte units
}

template <int XP>
__device__ __forceinline__ void il2_store4(uint2* act8, int k, int r, int j, float a, float b, float c, float d) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 h01 = {(_Float16)a, (_Float16)b}, h23 = {(_Float16)c, (_Float16)d};
  uint2 v;
  v.x = __builtin_bit_cast(unsigned, h01); v.y = __builtin_bit_cast(unsigned, h23);
  act8[il2_slot8<XP>(k, r, j, 0)] = v;
  if (XP == 2) {
    const h2 l01 = {(_Float16)(a - (float)h01[0]), (_Float16)(b - (float)h01[1])};
    const h2 l23 = {(_Float16)(c - (float)h23[0]), (_Float16)(d - (float)h23[1])};
    v.x = __builtin_bit_cast(unsigned, l01); v.y = __builtin_bit_cast(unsigned, l23);
    act8[il2_slot8<XP>(k, r, j, 1)] = v;
  }
}

// Weight fragments in flight: a ring of PF + 1 k-steps (two unit tiles x AP planes each).  A layer starts with its first PF
// k-steps already requested (il2_prefetch, issued before the barriers / epilogue of the layer in front of it: an L2 round trip
// per layer start was otherwise exposed four times per pass).
#ifndef IL2_PF
#define IL2_PF 3
#endif
constexpr int kIl2Pf = IL2_PF;
template <int TERMS>
struct Il2Ring { tf_h8 a[kIl2Pf + 1][2][IL2<TERMS>::AP]; };

template <int TERMS>
__device__ __forceinline__ void il2_prefetch(const tf_h8* __restrict__ Wl /* wave-uniform */, int T0, int lane, Il2Ring<TERMS>& ring) {
  typedef IL2<TERMS> C;
  // scalar base + 32-bit lane offset (global_load ... v_off, s[base:base+1]): written as a per-lane 64-bit pointer every
  // k-step's address became a loop-invariant VGPR pair, hoisted out of the pass loop and spilled
  const tf_h8* wp = Wl + T0 * 128;
#pragma unroll
  for (int s = 0; s < kIl2Pf; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < C::AP; ++p) ring.a[s][t][p] = wp[(unsigned)((s * 8 + t) * 128 + p * 64 + lane)];
}

// One hidden layer for this wave's two unit tiles T0, T0 + 1: acc[t][r] = bias + W x over K16 k-steps.
// Wl: the layer's fragment image [k-step][8 unit tiles][hi|lo][lane] in 16-byte units, already offset by `lane`.
template <int K16, int TERMS>
__device__ __forceinline__ void il2_layer(const tf_h8* __restrict__ Wl /* wave-uniform */, int T0, int lane,
                                          const tf_h8* __restrict__ actl /* + lane */,
                                          Il2Ring<TERMS>& ring, const f32x16 (&bias)[2], f32x16 (&acc)[2][4]) {
  typedef IL2<TERMS> C;
  constexpr int PF = kIl2Pf;
  const tf_h8* wp = Wl + T0 * 128;                    // (unit tile T0, plane 0) of k-step 0
  // B fragments (activations) of k-step s + 1 are requested BEFORE the MFMAs of k-step s (register double buffer): read, wait and
  // multiply in sequence left the matrix pipe idle for an LDS round trip (~200 cycles) in front of every 256 cycles of MFMAs.
  tf_h8 bq[2][4][C::XP];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int p = 0; p < C::XP; ++p) bq[0][r][p] = actl[((0 * 4 + r) * C::XP + p) * 64];
#pragma unroll
  for (int s = 0; s < K16; ++s) {
    if (s + PF < K16) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < C::AP; ++p) ring.a[(s + PF) % (PF + 1)][t][p] = wp[(unsigned)(((s + PF) * 8 + t) * 128 + p * 64 + lane)];
    }
    if (s + 1 < K16) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int p = 0; p < C::XP; ++p) bq[(s + 1) & 1][r][p] = actl[(((s + 1) * 4 + r) * C::XP + p) * 64];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const tf_h8 b_hi = bq[s & 1][r][0];
      const tf_h8 b_lo = bq[s & 1][r][C::XP - 1];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const tf_h8 a_hi = ring.a[s % (PF + 1)][t][0];
        acc[t][r] = tf_mfma_h(a_hi, b_hi, s == 0 ? bias[t] : acc[t][r]);
        if (TERMS == 3) acc[t][r] = tf_mfma_h(a_hi, b_lo, acc[t][r]);
        if (TERMS >= 2) acc[t][r] = tf_mfma_h(ring.a[s % (PF + 1)][t][C::AP - 1], b_hi, acc[t][r]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // bounds how far the loads of later k-steps are hoisted (registers)
  }
}

// ReLU + conversion of this wave's 64 output units into the next layer's B-fragments (k-steps 2 T0 .. 2 T0 + 3).
template <int TERMS>
__device__ __forceinline__ void il2_publish(tf_h8* __restrict__ actl /* + lane */, int T0, const f32x16 (&acc)[2][4]) {
  typedef IL2<TERMS> C;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float x8[8];
        tf_h8* dst = actl + (((2 * (T0 + t) + u) * 4 + r) * C::XP) * 64;
        if (TERMS == 3) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = tf_relu(acc[t][r][8 * u + e]);
          tf_h8 hi, lo;
          tf_split8(x8, hi, lo);
          dst[0] = hi; dst[64] = lo;
        } else {
          // one rounded operand per value: ReLU AFTER the conversion, on the packed halves (v_pk_max_f16: one instruction per two
          // values instead of one v_max_f32 per value).  Rounding to f16 is monotonic and keeps the sign, so max(f16(x), 0) = f16(max(x, 0)):
          // colours bit-identical; time unchanged (9.6 ms: the epilogue's vector instructions are not what a pass waits for).
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = acc[t][r][8 * u + e];
          tf_h8 hi;
          tf_cvt8(x8, hi);
          const tf_h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
          dst[0] = __builtin_elementwise_max(hi, zero);
        }
      }
}

template <int TERMS>
__global__ void __launch_bounds__(256, TERMS == 3 ? 1 : 2)
inner_light2_kernel(const float* __restrict__ ws_arg, const float* __restrict__ pts, const float* __restrict__ view,
                    const float* __restrict__ nrm, long long m_arg, const long long* __restrict__ idx,
                    const long long* __restrict__ count_dev, const float* __restrict__ depth, float near_eps, float exp_max,
                    float* __restrict__ out) {
  typedef IL2<TERMS> C;
  long long m = m_arg;
  if (count_dev) m = min(m_arg, *count_dev);
  if (m <= 0) return;
  __shared__ __attribute__((aligned(16))) tf_h8 act[C::ACT16];
  __shared__ __attribute__((aligned(16))) float lbias[4 * 2 * 256];   // [layer][lane half][tile * 16 + reg]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5;
  for (int i = tid; i < 3 * 256 + 32; i += 256) {
    const int layer = i < 768 ? i / 256 : 3, r = i < 768 ? i % 256 : i - 768;   // packed order: r = n * 2 + half
    lbias[layer * 512 + (r & 1) * 256 + (r >> 1)] = ws_arg[kIB1 + i];
  }
  const long long n_pass = (m + C::R - 1) / C::R;
  const float vsign = idx ? -1.f : 1.f;
  const int T0 = 2 * wave;
  uint2* act8 = reinterpret_cast<uint2*>(act);
  const int q_enc = 64 * (wave & 1) + lane;                // ray of the pass this lane encodes
  const int q_out = 32 * wave + (lane & 31);               // ray of the pass whose radiance this lane stores (lanes 0..31)
  // inputs of the first pass; inside the loop the NEXT pass's index row and input rows are requested a layer or two ahead
  // of their use (two dependent L2 / HBM round trips otherwise open every pass)
  float in9[9];                                            // pts | nrm | view of the ray this lane encodes
  long long nsrc;
  auto load_inputs = [&](long long src) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { in9[k] = pts[3 * src + k]; in9[3 + k] = nrm[3 * src + k]; in9[6 + k] = view[3 * src + k]; }
  };
  {
    long long row = (long long)blockIdx.x * C::R + q_enc;
    if (row >= m) row = m - 1;
    nsrc = idx ? idx[row] : row;
    load_inputs(nsrc);
  }
  Il2Ring<TERMS> ring;
  il2_prefetch<TERMS>(reinterpret_cast<const tf_h8*>(ws_arg) + kP1 / 4, T0, lane, ring);
#define IL2_STAMP() do {} while (0)
  for (long long pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    const float* ws = ws_arg;
    asm volatile("" : "+s"(ws));
    IL2_STAMP();
    const tf_h8* W = reinterpret_cast<const tf_h8*>(ws);     // wave-uniform, 16-byte units (kP1 etc. are float offsets)
    // ---- encodings of the pass's 128 rays, 64 per wave, the work of a ray split over TWO waves so that all four are equally
    // busy (the whole IDE on one wave pair cost 6.9 k cycles against 5.3 k for the positional encoding on the other):
    //   waves 0-1: positional encoding (features 72..122) + IDE columns 0..11  (l = 1, 2, 4 and the first two orders of l = 8)
    //   waves 2-3: IDE columns 12..35 (rest of l = 8, all of l = 16)
    // Every 4-feature store granule has one owner (12 and 36 + 12 are multiples of 4).
    {
      const int r = q_enc >> 5, j = q_enc & 31;
      // reflected direction (both wave pairs need it)
      float n[3] = {in9[3], in9[4], in9[5]};
      float v[3] = {vsign * in9[6], vsign * in9[7], vsign * in9[8]};
      float inv = 1.f / fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);
      n[0] *= inv; n[1] *= inv; n[2] *= inv;
      inv = 1.f / fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
      v[0] *= inv; v[1] *= inv; v[2] *= inv;
      const float vn = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
      const float rx = vn * n[0] * 2.f - v[0], ry = vn * n[1] * 2.f - v[1], rz = vn * n[2] * 2.f - v[2];
      const __attribute__((address_space(4))) float* mat =
          (const __attribute__((address_space(4))) float*)(unsigned long long)(ws + kIdeMat);
      // IDE columns [C0, C1): sph[col] = (rx + i ry)^mm * sum_k mat[k][col] rz^k; Re -> feature col, Im -> feature 36 + col
      auto ide_cols = [&](auto c0_, auto c1_) {
        constexpr int C0 = decltype(c0_)::value, C1 = decltype(c1_)::value;
        constexpr int KMAX = C1 <= 12 ? 8 : 16, MMAX = C1 <= 12 ? 4 : 16;
        float zp[KMAX + 1], cre[MMAX + 1], cim[MMAX + 1];
        zp[0] = 1.f; cre[0] = 1.f; cim[0] = 0.f;
#pragma unroll
        for (int k = 1; k <= KMAX; ++k) zp[k] = zp[k - 1] * rz;
#pragma unroll
        for (int k = 1; k <= MMAX; ++k) {
          cre[k] = cre[k - 1] * rx - cim[k - 1] * ry;
          cim[k] = cre[k - 1] * ry + cim[k - 1] * rx;
        }
        float re[C1 - C0], im[C1 - C0];
#pragma unroll
        for (int d = 0; d < 5; ++d) {
#pragma unroll
          for (int mm = 0; mm <= (1 << d); ++mm) {
            const int col = (1 << d) - 1 + d + mm;
            if (col >= C0 && col < C1) {
              float poly = 0.f;
#pragma unroll
              for (int k = 0; k <= (1 << d) - mm; ++k) poly += zp[k] * mat[k * 36 + col];
              re[col - C0] = cre[mm] * poly;
              im[col - C0] = cim[mm] * poly;
            }
          }
        }
#pragma unroll
        for (int g = 0; g < (C1 - C0) / 4; ++g) {
          il2_store4<C::XP>(act8, C0 + 4 * g, r, j, re[4 * g], re[4 * g + 1], re[4 * g + 2], re[4 * g + 3]);
          il2_store4<C::XP>(act8, 36 + C0 + 4 * g, r, j, im[4 * g], im[4 * g + 1], im[4 * g + 2], im[4 * g + 3]);
        }
      };
      if (wave < 2) {
        const float p[3] = {in9[0], in9[1], in9[2]};
        float enc[56];
#pragma unroll
        for (int k = 0; k < 3; ++k) enc[k] = p[k];
        if (TERMS <= 2) {
          // operands are rounded to f16 (2^-12) on their way into the matrix cores: octaves 1..7 by angle doubling from ONE
          // accurate sincos per coordinate (error doubles per octave: <= 1.3e-5 at 2^7 p) instead of 24 range-reduced evaluations
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float sn, cs;
            if (fabsf(p[k]) < 3.f) tf_sincos_small(p[k], sn, cs); else tf_sincos(p[k], sn, cs);
            enc[3 + k] = sn; enc[6 + k] = cs;
#pragma unroll
            for (int f = 1; f < 8; ++f) {
              const float s2 = 2.f * sn * cs, c2 = fmaf(-2.f * sn, sn, 1.f);
              sn = s2; cs = c2;
              enc[3 + 6 * f + k] = sn; enc[3 + 6 * f + 3 + k] = cs;
            }
          }
        } else if (__all(fabsf(p[0]) < 3.f && fabsf(p[1]) < 3.f && fabsf(p[2]) < 3.f)) {
#pragma unroll
          for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int k = 0; k < 3; ++k) tf_sincos_small(p[k] * (float)(1 << f), enc[3 + 6 * f + k], enc[3 + 6 * f + 3 + k]);
        } else {
#pragma unroll
          for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int k = 0; k < 3; ++k) tf_sincos(p[k] * (float)(1 << f), enc[3 + 6 * f + k], enc[3 + 6 * f + 3 + k]);
        }
#pragma unroll
        for (int k = 51; k < 56; ++k) enc[k] = 0.f;
#pragma unroll
        for (int g = 0; g < 14; ++g) il2_store4<C::XP>(act8, 72 + 4 * g, r, j, enc[4 * g], enc[4 * g + 1], enc[4 * g + 2], enc[4 * g + 3]);
        ide_cols(std::integral_constant<int, 0>{}, std::integral_constant<int, 12>{});
      } else {
        ide_cols(std::integral_constant<int, 12>{}, std::integral_constant<int, 36>{});
      }
    }
    // index rows: this pass's output ray, the next pass's input ray
    long long orow = pass * C::R + q_out;
    const bool ovalid = orow < m;
    if (!ovalid) orow = m - 1;
    const long long osrc = idx ? idx[orow] : orow;
    const long long npass = pass + gridDim.x;
    if (npass < n_pass) {
      long long row = npass * C::R + q_enc;
      if (row >= m) row = m - 1;
      nsrc = idx ? idx[row] : row;
    }
    IL2_STAMP();
    __syncthreads();
    IL2_STAMP();
    f32x16 acc[2][4], bias[2];
    // ---- layer 1 (K = 128)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 v4 = *reinterpret_cast<const float4*>(lbias + hh * 256 + (T0 + t) * 16 + 4 * qd);
        bias[t][4 * qd] = v4.x; bias[t][4 * qd + 1] = v4.y; bias[t][4 * qd + 2] = v4.z; bias[t][4 * qd + 3] = v4.w;
      }
    il2_layer<8, TERMS>(W + kP1 / 4, T0, lane, act + lane, ring, bias, acc);
    il2_prefetch<TERMS>(W + kH2 / 4, T0, lane, ring);
    IL2_STAMP();
    __syncthreads();
    IL2_STAMP();
    il2_publish<TERMS>(act + lane, T0, acc);
    IL2_STAMP();
    __syncthreads();
    IL2_STAMP();
    // ---- layers 2, 3 (K = 256)
    const float dep = depth ? depth[osrc] : 1.f;
    if (npass < n_pass) load_inputs(nsrc);                  // next pass's input rows (consumed at the top of the next iteration)
    tf_h8 a4[16];                                            // layer 4's weight fragments (hi), requested under layer 3's epilogue
#pragma unroll
    for (int layer = 1; layer < 3; ++layer) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 v4 = *reinterpret_cast<const float4*>(lbias + layer * 512 + hh * 256 + (T0 + t) * 16 + 4 * qd);
          bias[t][4 * qd] = v4.x; bias[t][4 * qd + 1] = v4.y; bias[t][4 * qd + 2] = v4.z; bias[t][4 * qd + 3] = v4.w;
        }
      il2_layer<16, TERMS>(W + (layer == 1 ? kH2 : kH3) / 4, T0, lane, act + lane, ring, bias, acc);
      if (layer == 1) il2_prefetch<TERMS>(W + kH3 / 4, T0, lane, ring);
      if (layer == 2) {
        const tf_h8* W4 = W + kH4 / 4;
#pragma unroll
        for (int s = 0; s < 16; ++s) a4[s] = W4[(unsigned)(s * 128 + lane)];
      }
      if (layer == 1) IL2_STAMP();
      __syncthreads();
      il2_publish<TERMS>(act + lane, T0, acc);
      __syncthreads();
      if (layer == 1) IL2_STAMP();
    }
    IL2_STAMP();
    // ---- layer 4 (256 -> 3): wave w takes ray tile w
    {
      f32x16 o;
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] = lbias[1536 + hh * 256 + j];
      const tf_h8* W4 = W + kH4 / 4;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const tf_h8 b_hi = act[((s * 4 + wave) * C::XP) * 64 + lane];
        o = tf_mfma_h(a4[s], b_hi, o);
        if (TERMS == 3) o = tf_mfma_h(a4[s], act[((s * 4 + wave) * C::XP + 1) * 64 + lane], o);
        if (TERMS >= 2) o = tf_mfma_h(W4[(unsigned)(s * 128 + 64 + lane)], b_hi, o);
      }
      il2_prefetch<TERMS>(W + kP1 / 4, T0, lane, ring);           // the next pass's first layer
      if (ovalid && hh == 0) {
        const float near = (depth && !(dep > near_eps)) ? 0.f : 1.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * osrc + c] = expf(fminf(o[c], exp_max)) * near;
      }
    }
    IL2_STAMP();
    __syncthreads();      // the next pass's encodings overwrite the activation image
  }
}

#define IL_ARGS workspace, pts, view, nrm, m, idx, count_dev, depth, near_eps, exp_max, out
void tf_inner_light_launch_f32(const float* workspace, const float* pts, const float* view, const float* nrm, long long m, const long long* idx,
                               const long long* count_dev, const float* depth, float near_eps, float exp_max, float* out, hipStream_t stream) {
  long long blocks = (m + 127) / 128;
  if (blocks > 1024) blocks = 1024;      // persistent workgroups (one resident per CU at a time)
  inner_light_kernel<<<(unsigned)blocks, 256, 0, stream>>>(IL_ARGS);
}
void tf_inner_light_launch_f16(const float* workspace, const float* pts, const float* view, const float* nrm, long long m, const long long* idx,
                               const long long* count_dev, const float* depth, float near_eps, float exp_max, float* out, hipStream_t stream) {
  long long blocks = (m + 127) / 128;      // 128 rays per pass, persistent workgroups (two resident per CU)
  if (blocks > 512) blocks = 512;
  inner_light2_kernel<1><<<(unsigned)blocks, 256, 0, stream>>>(IL_ARGS);
}
#undef IL_ARGS
