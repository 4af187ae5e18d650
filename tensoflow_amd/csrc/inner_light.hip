// Indirect ("inner") light network of the rendering integral:
// MCShadingNetwork.get_inner_lights (network/fields.py:905-911) =
//   cat[pos_enc8(p) (51), IDE5(reflect(view, n), kappa_inv = 0) (72)] -> 123-256-256-256-3 (ReLU),
//   out = exp(min(x, exp_max))            (make_predictor_4layer, network/other_field.py:86-119)
// evaluated for every secondary ray that hits geometry (327 kflop per ray: the largest flop term of the integral); the same kernel
// serves predict_outer_lights('direction') (fields.py:913-916) on the rays that miss.  Weight-norm is folded by the caller.
//
//   TfPrecision     kernel                                         file                      operands of a product
//   TF_PREC_F16X3   inner_light3_kernel<., 3>  (64-ray passes)     this file                 weights hi + lo, activations hi + lo (3 MFMAs): fp32-grade, the library default
//   TF_PREC_F16X2   inner_light3_kernel<., 2>  (128-ray passes)    this file                 weights hi + lo, activations f16 once per layer (2 MFMAs): opt-in
//   TF_PREC_F16     inner_light2_kernel<1>     (column-owned)      inner_light_modes.hip     plain f16 (1 MFMA): opt-in, not parity grade
//   TF_PREC_F32     inner_light_kernel         (slab ring)         inner_light_modes.hip     exact fp32 MFMA: the yardstick of the parity tests
// (the training forward, tf_inner_light_indexed_train_fwd, is the SAVE instantiation of the 64-ray form; tf_set_launch_budget's
// inner_teams = 1 the TEAMS = 1 instantiations.)  Dev-only stamps are compiled in with -DTF_DEV.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "mfma_mlp.h"
#include "tf_common.h"
#include "tf_internal.h"
#include "inner_light_ws.h"


extern "C" size_t tf_inner_light_workspace_floats(void) { return kInnerWsFloats; }
static void ide_tables_host(float* mat);

// ---- IDE tables (Ref-NeRF eq. 6-8; utils/ref_utils.py:8-78): (l, m) for l = 1,2,4,8,16, m = 0..l
static void ide_tables_host(float* mat /*[17][36]*/) {
  auto fact = [](int n) { double r = 1; for (int i = 2; i <= n; ++i) r *= i; return r; };
  int col = 0;
  for (int i = 0; i < 17 * 36; ++i) mat[i] = 0.f;
  for (int d = 0; d < 5; ++d) {
    const int l = 1 << d;
    for (int m = 0; m <= l; ++m, ++col) {
      for (int k = 0; k <= l - m; ++k) {
        const double a = 0.5 * (l + k + m - 1.0);
        double gb = 1.0;
        for (int j = 0; j < l; ++j) gb *= (a - j);
        gb /= fact(l);
        const double leg = std::pow(-1.0, m) * std::pow(2.0, l) * fact(l) / fact(k) / fact(l - k - m) * gb;
        const double sph = std::sqrt((2.0 * l + 1.0) * fact(l - m) / (4.0 * M_PI * fact(l + m))) * leg;
        mat[k * 36 + col] = (float)sph;
      }
    }
  }
}

// computed once per process (C++11 magic static: thread-safe), read-only afterwards
static const float* ide_tables_cached() {
  struct Table { float v[17 * 36]; Table() { ide_tables_host(v); } };
  static const Table t;
  return t.v;
}


// =====================================================================================================================
// Staggered two-team form of the column-owned kernel: fp32-grade (f16x3) products at two waves per SIMD.
// With hi + lo activation planes the column-owned image of a 128-ray pass fills 128 KB of LDS: one workgroup per CU, nothing
// to issue while a wave converts / publishes / encodes (the matrix pipe idle 47 % of the pass), which is why f16x3 stayed on the
// slab-ring kernel (24.7 ms per 29.7 M rays against the 9.5 ms of the plain-f16 column-owned form).  Here ONE 512-thread
// workgroup per CU runs two TEAMS of four waves, each on its own 64-ray pass (2 ray tiles; 64 KB of activation planes per team):
// wave w of team A and wave w of team B share a SIMD, and team A runs three steps ahead of team B in the six-step cycle
//     [FE  M1  P1  M2  P2  M3]      FE = finish the previous pass (256 -> 3 layer on the vector unit, exact fp32) + encode this one
//                                   Mk = layer k's matrix products, Pk = ReLU + hi/lo split + publish to LDS
// so that in every step exactly one of the two waves of a SIMD is in a matrix phase and its partner in a vector / LDS phase
// (B: FE | A: M2), (M1 | P2), (P1 | M3), (M2 | FE), (P2 | M1), (M3 | P1): the matrix pipe always has one wave feeding it, the
// encodings (the longest vector phase) sit under the partner's longest matrix phase.  One workgroup barrier per step.
// A wave owns 64 output units (2 unit tiles) x 64 rays (2 ray tiles): per k-step 4 coalesced 1 KB weight loads from L2 into a
// register ring, 4 ds_read_b128 of activation fragments, 12 MFMAs.  The 256 -> 3 layer needs 3 of a 32-unit tile's rows: run on
// the matrix cores it costs 48 MFMAs per 64 rays and a publish of layer 3; here every wave multiplies its own 64 x 64 post-ReLU
// accumulators with the three weight rows (192 FMAs per lane, exact fp32), lane halves and the four waves are summed through LDS
// in a fixed order (bit-reproducible).
// Input row of the staggered kernel (128 columns): [IDE Re (36) | IDE Im (36) | positional, wave-major (48) | p (3) | 0 (5)].
// The positional block of wave w (12 columns) holds octaves 2w, 2w + 1: pair i = 0..5 -> (octave 2w + i / 3, coordinate i % 3), sin at
// 2i, cos at 2i + 1 -- every wave of a team evaluates 6 of a ray's 24 sincos pairs and owns whole 4-column store granules.
// -> column of the reference's row (network/fields.py:905-911: [p, sin / cos per octave (51) | IDE (72)]) or -1 (zero)
__host__ __device__ __forceinline__ int il3_orig_col(int k) {
  if (k < 72) return 51 + k;
  if (k < 120) {
    const int w = (k - 72) / 12, e = (k - 72) % 12, i = e >> 1, f = 2 * w + i / 3, q = i % 3;
    return 3 + 6 * f + ((e & 1) ? 3 : 0) + q;
  }
  return k < 123 ? k - 120 : -1;
}
// in_cols = 123: the inner-light net's first layer; in_cols = 72: the outer-light net's ('direction': IDE columns only, the positional
// columns of the image are zero)
static __global__ void __launch_bounds__(256) inner_light_cols3_kernel(const float* __restrict__ W /*[256,in_cols]*/, int in_cols,
                                                                       float* __restrict__ Wp /*[256,128]*/) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 256 * 128) return;
  const int row = e >> 7, k = e & 127;
  int c = il3_orig_col(k);
  if (in_cols == 72) c = c >= 51 ? c - 51 : -1;
  Wp[e] = c >= 0 ? W[row * in_cols + c] : 0.f;
}

#ifdef TF_DEV
__device__ unsigned long long g_il3_stamps[8 * 16];
extern "C" void tf_il3_stamps(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_il3_stamps), sizeof(unsigned long long) * 128); }
#endif
#ifndef IL3_PF
#define IL3_PF 2
#endif
#ifndef IL3_INTERLEAVE
#define IL3_INTERLEAVE 1
#endif
#ifndef IL3_OUT_LATE
#define IL3_OUT_LATE 1
#endif
#ifndef IL3_POS_EARLY
#define IL3_POS_EARLY 1
#endif
// Two operand forms of the staggered kernel:
//   TERMS = 3 (TF_PREC_F16X3): activations AND weights split hi + lo, a_hi w_hi + a_lo w_hi + a_hi w_lo; a team pass is 64 rays
//     (2 ray tiles x 2 planes = 64 KB of B-fragments per team);
//   TERMS = 2 (TF_PREC_F16X2): activations rounded to f16 ONCE per layer, weights split hi + lo (x w_hi + x w_lo); one plane per
//     ray tile, so the same 64 KB hold a 128-ray team pass (4 ray tiles): every weight fragment fetched from L2 feeds twice the
//     rays (the f16x3 form's roof is the L2 -> CU fill rate of its weight stream) and a product term costs two MFMAs, not three.
//     Every lane stands for two rays of the pass (ray 64 e + lane, e = 0, 1) in the vector steps.
template <int TERMS>
struct IL3 {
  static constexpr int RT = TERMS == 3 ? 2 : 4;      // ray tiles per team pass
  static constexpr int XP = TERMS == 3 ? 2 : 1;      // activation planes (hi | lo)
  static constexpr int NR = RT / 2;                  // rays per lane in the vector steps
  static constexpr int RAYS = 32 * RT;               // rays per team pass
  static constexpr int TEAM16 = 16 * RT * XP * 64;   // 16-byte units of one team's activation image (64 KB in both forms)
  static constexpr int STAGE = 3 * RAYS * 4 + RAYS;  // floats of one team's input stage: hit point | normal | direction rows (16 B each) | depths
  static constexpr int PART = 4 * 3 * RT * 32;       // floats of one team's partial sums of the 256 -> 3 layer: [wave][ray tile * 3 + output][ray]
  static constexpr int PF = IL3_PF;                  // k-steps of weight fragments in flight
};
struct Il3Ring { tf_h8 a[IL3_PF + 1][2][2]; };

// 8-byte slot of input column k (k % 4 == 0) in a team's layer-1 B-fragment image, RELATIVE to the slot of the lane's ray
// (il3_ray_slot8): [k-step][ray tile][hi|lo][lane][8 halves]; a compile-time constant for a compile-time k
template <int TERMS>
__device__ __forceinline__ constexpr int il3_col_slot8(int k, int plane) {
  return (((k >> 4) * IL3<TERMS>::RT * IL3<TERMS>::XP + plane) * 64 + 32 * ((k >> 2) & 1)) * 2 + ((k >> 3) & 1);
}
template <int TERMS>
__device__ __forceinline__ int il3_ray_slot8(int r, int j) { return (r * IL3<TERMS>::XP * 64 + j) * 2; }
// hi = f16(x), lo = f16(x - hi) of four values, as packed halves (the split of il3_store4 without the store)
__device__ __forceinline__ void il3_split4(float a, float b, float c, float d, uint2& hv, uint2& lv) {
  asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
      "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(hv.x), "=&v"(hv.y), "=&v"(lv.x), "=&v"(lv.y)
      : "v"(a), "v"(b), "v"(c), "v"(d));
}
template <int TERMS>
__device__ __forceinline__ void il3_store4(uint2* a8 /* team image + the ray's slot */, int k, float a, float b, float c, float d) {
  if (TERMS == 3) {
    // hi = f16(x), lo = f16(x - hi): 6 instructions per four values (v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16, as tf_split8); written as
    // (_Float16)(x - (float)hi) the compiler spends ~5 instructions per VALUE -- 700 of an encoding pass's vector instructions per ray
    uint2 hv, lv;
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hv.x), "=&v"(hv.y), "=&v"(lv.x), "=&v"(lv.y)
        : "v"(a), "v"(b), "v"(c), "v"(d));
    a8[il3_col_slot8<TERMS>(k, 0)] = hv;
    a8[il3_col_slot8<TERMS>(k, 1)] = lv;
  } else {
    uint2 hv;
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_cvt_pk_f16_f32 %1, %4, %5"
        : "=&v"(hv.x), "=&v"(hv.y)
        : "v"(a), "v"(b), "v"(c), "v"(d));
    a8[il3_col_slot8<TERMS>(k, 0)] = hv;
  }
}

typedef const __attribute__((address_space(1))) tf_h8* il3_gw_t;
// Weight-fragment addresses: ONE lane offset register (16 * lane bytes) for the whole kernel, a wave-uniform base per k-step kept in
// scalar registers (advanced by scalar adds, made opaque so that it is neither folded into per-load vector offsets nor hoisted) and
// the (unit tile, plane) sub-block in the instruction's immediate offset (0 / 1 / 2 / 3 KB).  Written as one index expression
// wp[(s * 8 + t) * 128 + p * 64 + lane] every load of every k-step got its own loop-invariant offset register (96 of the 256).
__device__ __forceinline__ il3_gw_t il3_kstep_base(il3_gw_t Wl, int T0, int s) {
#if defined(IL3_ABLATE_WSTREAM) && IL3_ABLATE_WSTREAM == 3   // dev-only (garbage results): every load of every wave reads the SAME 4 KB -- the loads,
  T0 = 0; s = 0;                                              // the L1 look-ups and the register writes stay, the L2 / fabric traffic goes
#endif
  il3_gw_t b = Wl + T0 * 128 + s * 1024;           // k-step stride: 8 unit tiles x (hi | lo) x 64 lanes = 1024 fragments of 16 bytes
  asm volatile("" : "+s"(b));
  return b;
}
__device__ __forceinline__ void il3_prefetch(il3_gw_t Wl /* wave-uniform */, int T0, int lane, Il3Ring& ring) {
#pragma unroll
  for (int s = 0; s < IL3_PF; ++s) {
    il3_gw_t b = il3_kstep_base(Wl, T0, s);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) ring.a[s][t][p] = b[lane + t * 128 + p * 64];
  }
}

// acc[t][r] (+)= W x over K16 k-steps for this wave's unit tiles T0, T0 + 1 and the team's ray tiles; the first PF k-steps
// of weight fragments are already in `ring` (il3_prefetch, issued a step earlier).
template <int K16, int TERMS>
__device__ __forceinline__ void il3_layer(il3_gw_t Wl /* wave-uniform */, int T0, int lane,
                                          const tf_h8* __restrict__ actl /* team image + lane */, Il3Ring& ring,
                                          f32x16 (&acc)[2][IL3<TERMS>::RT]) {
  constexpr int PF = IL3_PF, RT = IL3<TERMS>::RT, XP = IL3<TERMS>::XP;
  {
  tf_h8 bq[2][RT][XP];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int p = 0; p < XP; ++p) bq[0][r][p] = actl[((0 * RT + r) * XP + p) * 64];
#pragma unroll
  for (int s = 0; s < K16; ++s) {
#if defined(IL3_ABLATE_WSTREAM) && IL3_ABLATE_WSTREAM != 3   // dev-only timing ablation (results are garbage): the weight fragments of a layer are NOT streamed from L2 -- the ring keeps
    // what the prefetch of its first k-steps brought.  Measured (round 6): 15.6 ms against 21.5 at the SAME cycles per pass -- 2.13 GHz
    // instead of 1.57: the stream costs power, not latency (control: tools/exp_il3.py IL_PERIODIC_W=1).
    if (false) {
#else
    if (s + PF < K16) {
#endif
      il3_gw_t b = il3_kstep_base(Wl, T0, s + PF);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) ring.a[(s + PF) % (PF + 1)][t][p] = b[lane + t * 128 + p * 64];
    }
    if (s + 1 < K16) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int p = 0; p < XP; ++p) bq[(s + 1) & 1][r][p] = actl[(((s + 1) * RT + r) * XP + p) * 64];
    }
#if IL3_INTERLEAVE
    // term-major: consecutive MFMAs go to DIFFERENT accumulators (the order of the three terms per accumulator is the same)
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[t][r] = tf_mfma_h(ring.a[s % (PF + 1)][t][term == 2 ? 1 : 0], bq[s & 1][r][term == 1 ? XP - 1 : 0], acc[t][r]);
#else
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const tf_h8 b_hi = bq[s & 1][r][0], b_lo = bq[s & 1][r][XP - 1];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const tf_h8 a_hi = ring.a[s % (PF + 1)][t][0];
        acc[t][r] = tf_mfma_h(a_hi, b_hi, acc[t][r]);
        acc[t][r] = tf_mfma_h(a_hi, b_lo, acc[t][r]);
        acc[t][r] = tf_mfma_h(ring.a[s % (PF + 1)][t][1], b_hi, acc[t][r]);
      }
    }
#endif
    // order inside a k-step: the activation fragments of k-step s + 1 and the weight fragments of k-step s + PF are REQUESTED before
    // the 12 MFMAs of k-step s (left to itself the scheduler sinks the LDS reads behind the last MFMA -- they reuse the registers of
    // the fragments in use -- and every k-step then waits out an LDS round trip with the matrix pipe idle)
#if IL3_INTERLEAVE
    // ... and ONE memory instruction behind each of the first MFMAs rather than all eight in front of the first: an instruction of
    // this wave issued between two MFMAs costs nothing (the matrix pipe is 32 cycles into the previous one), eight of them in a row
    // with the scalar address arithmetic leave it idle for ~50 cycles per k-step (cycle stamps of a team alone on its CU: 439 cycles
    // per k-step of 12 MFMAs = 384)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (s + 1 < K16 && i < RT * XP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (s + PF < K16) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * RT * TERMS - 8, 0);
#else
    __builtin_amdgcn_sched_group_barrier(0x100, RT * XP, 0);         // DS reads
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);               // VMEM reads
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * RT * TERMS, 0);  // MFMA
#endif
    __builtin_amdgcn_sched_barrier(0);      // bounds how far the loads of later k-steps are hoisted (registers)
  }
  }
}

template <int RT>
__device__ __forceinline__ void il3_bias(const float* __restrict__ lb /* layer's rows of this lane half */, int T0, f32x16 (&acc)[2][RT]) {
  {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const float4 v4 = *reinterpret_cast<const float4*>(lb + (T0 + t) * 16 + 4 * qd);
#pragma unroll
      for (int r = 0; r < RT; ++r) { acc[t][r][4 * qd] = v4.x; acc[t][r][4 * qd + 1] = v4.y; acc[t][r][4 * qd + 2] = v4.z; acc[t][r][4 * qd + 3] = v4.w; }
    }
  }
}

// ReLU + hi / lo split of this wave's 64 units x 64 rays into the next layer's B-fragments (k-steps 2 T0 .. 2 T0 + 3): the 64-ray form
// (the 128-ray form converts and publishes per unit-tile half: il4_pack / il4_publish)
template <int TERMS>
__device__ __forceinline__ void il3_publish(tf_h8* __restrict__ actl /* team image + lane */, int T0, const f32x16 (&acc)[2][IL3<TERMS>::RT]) {
  static_assert(TERMS == 3, "64-ray form");
  constexpr int RT = IL3<TERMS>::RT, XP = IL3<TERMS>::XP;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float x8[8];
        tf_h8* dst = actl + (((2 * (T0 + t) + u) * RT + r) * XP) * 64;
#pragma unroll
        for (int e = 0; e < 8; ++e) x8[e] = tf_relu(acc[t][r][8 * u + e]);
        tf_h8 hi, lo;
        tf_split8(x8, hi, lo);
        dst[0] = hi; dst[64] = lo;
      }
}


// ---- 128-ray form (TERMS = 2): a wave's 64 units x 128 rays are 128 accumulator registers -- with the weight ring, the activation
// fragments and what lives across a matrix step more than the register allocator places without spilling (16-register tuples in a
// file fragmented by 4-register ones: 110-170 spilled registers in every arrangement tried).  The wave therefore runs a layer as
// TWO HALVES, one unit tile (32 units x 128 rays = 64 registers) at a time: every weight fragment is still fetched once per 128 rays
// (two fragments -- hi | lo of one tile -- feed 8 MFMAs), the activation fragments are read twice (62 B / clock and CU: half of the
// LDS rate); the first half's output waits as 32 registers of packed f16 until the second half has read the image.
#ifndef IL4_PF
#define IL4_PF 2
#endif
#ifndef IL4_SETPRIO
#define IL4_SETPRIO 1
#endif
struct Il4Ring { tf_h8 a[IL4_PF + 1][2]; };
__device__ __forceinline__ void il4_prefetch(il3_gw_t Wl /* wave-uniform */, int T, int lane, Il4Ring& ring) {
#pragma unroll
  for (int s = 0; s < IL4_PF; ++s) {
    il3_gw_t b = il3_kstep_base(Wl, T, s);
#pragma unroll
    for (int p = 0; p < 2; ++p) ring.a[s][p] = b[lane + p * 64];
  }
}
// acc[r] = bias + W x for unit tile T and the team's four ray tiles (the first PF k-steps of weight fragments are in `ring`).
// NEXT: the last PF k-steps request the first PF k-steps of unit tile T + 1 (the wave's second half of the layer): one continuous
// weight stream over both halves, the ring's slots keep rotating (2 K16 k-steps).
template <int K16, bool NEXT, int S0 /* ring slot of k-step 0 */>
__device__ __forceinline__ void il4_half(il3_gw_t Wl /* wave-uniform */, int T, int lane, const float* __restrict__ lb /* layer's rows of this lane half */,
                                         const tf_h8* __restrict__ actl /* team image + lane */, Il4Ring& ring, f32x16 (&acc)[4]) {
  constexpr int PF = IL4_PF;
#if IL4_SETPRIO
  // the wave in a matrix step wins the SIMD's issue arbitration against its partner's vector step (4.03 -> 3.93 ms per 7.4 M rays on the
  // bench's access pattern; priority 3 measures the same)
  __builtin_amdgcn_s_setprio(IL4_SETPRIO);
#endif
  // the bias tile is the C operand of each ray tile's FIRST product (no copies into the four accumulators: a wave does not overlap
  // its own vector instructions with its own MFMAs, so every vector instruction of a matrix step is 4 idle cycles of the matrix pipe)
  f32x16 bv;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const float4 v4 = *reinterpret_cast<const float4*>(lb + T * 16 + 4 * qd);
    bv[4 * qd] = v4.x; bv[4 * qd + 1] = v4.y; bv[4 * qd + 2] = v4.z; bv[4 * qd + 3] = v4.w;
  }
  tf_h8 bq[2][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bq[0][r] = actl[r * 64];
#pragma unroll
  for (int s = 0; s < K16; ++s) {
    if (s + PF < K16 || NEXT) {
      il3_gw_t b = s + PF < K16 ? il3_kstep_base(Wl, T, s + PF) : il3_kstep_base(Wl, T + 1, s + PF - K16);
#pragma unroll
      for (int p = 0; p < 2; ++p) ring.a[(S0 + s + PF) % (PF + 1)][p] = b[lane + p * 64];
    }
    if (s + 1 < K16) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bq[(s + 1) & 1][r] = actl[((s + 1) * 4 + r) * 64];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[r] = tf_mfma_h(ring.a[(S0 + s) % (PF + 1)][0], bq[s & 1][r], s == 0 ? bv : acc[r]);
      acc[r] = tf_mfma_h(ring.a[(S0 + s) % (PF + 1)][1], bq[s & 1][r], acc[r]);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // DS reads (k-step s + 1)
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM reads (k-step s + PF)
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // MFMA
    __builtin_amdgcn_sched_barrier(0);
  }
#if IL4_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}
// One rounding to f16 of a half's 32 units x 128 rays: [ray tile][k-step half u] fragments of 8 halves.  The vector fptrunc
// (v_cvt_pk_f16_f32, round to nearest even): compiler-visible, so that hipcc pads the MFMA -> reader hazard (it does not for an asm
// statement reading MFMA results).  The ReLU follows at publish time (a vector step, not a matrix step).
__device__ __forceinline__ void il4_pack(const f32x16 (&acc)[4], tf_h8 (&h)[4][2]) {
  typedef float f32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x8 xv;
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = acc[r][8 * u + e];
      h[r][u] = __builtin_convertvector(xv, tf_h8);
    }
}
// ReLU + store into the next layer's B-fragments: unit tile T is k-steps 2 T, 2 T + 1.  ReLU on the packed halves AS SIGNED 16-BIT
// INTEGERS -- rounding to f16 is monotonic and keeps the sign, a negative half is a negative integer, a non-negative one its own bit
// pattern: one v_pk_max_i16 per two values (max(x, 0) on halves costs a canonicalising v_pk_max_f16 x, x in front of each)
__device__ __forceinline__ void il4_publish(tf_h8* __restrict__ actl /* team image + lane */, int T, const tf_h8 (&h)[4][2]) {
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  const i16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      actl[((2 * T + u) * 4 + r) * 64] = __builtin_bit_cast(tf_h8, __builtin_elementwise_max(__builtin_bit_cast(i16x8, h[r][u]), zero));
}
// a half's share of the 256 -> 3 layer (exact fp32 on the vector unit): sum[r][c] += relu(acc[r]) . w4 rows of unit tile T.
// Packed: the sums are kept as PAIRS (even | odd unit of a register pair), one v_pk_fma_f32 per two products -- 10 instead of 16 vector
// instructions per ray tile and four units, in a matrix step where every vector instruction is 4 idle cycles of the matrix pipe; the
// packed operands are ReLU results (v_max), not MFMA results (the hazard of DESIGN 'things the compiler got wrong' 6).
typedef float il4_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void il4_out3(const float* __restrict__ w4h /* w4a + lane half * 128 */, int T, const f32x16 (&acc)[4], il4_f2 (&sum)[4][3]) {
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    float4 wr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) wr[c] = *reinterpret_cast<const float4*>(w4h + c * 256 + T * 16 + 4 * qd);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const il4_f2 xa = {tf_relu(acc[r][4 * qd]), tf_relu(acc[r][4 * qd + 1])}, xb = {tf_relu(acc[r][4 * qd + 2]), tf_relu(acc[r][4 * qd + 3])};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const il4_f2 wa = {wr[c].x, wr[c].y}, wb = {wr[c].z, wr[c].w};
        sum[r][c] = __builtin_elementwise_fma(xb, wb, __builtin_elementwise_fma(xa, wa, sum[r][c]));
      }
    }
  }
}

__device__ __forceinline__ void il3_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS writes are done; weight / input loads stay in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// OUTER: the same network shape fed with the IDE of the ray DIRECTION itself (MCShadingNetwork.predict_outer_lights, 'direction':
// network/fields.py:913-916 -- sph_enc(directions, 0), no reflection, no normalisation, no positional columns: their weights are zero
// in the image and `pts` / `nrm` alias `view`)

// SAVE: ReLU(acc) of this wave's 64 units x 64 rays as fp32 rows acts_l[row][256] (row = pass * 64 + ray < m): accumulator register
// j of lane (ray, half h) is unit 32 tile + (j & 3) + 8 (j >> 2) + 4 h -- four consecutive units per group of four registers
__device__ __forceinline__ void il3_save_acts(float* __restrict__ acts_l, long long row0, long long m, int T0, int lane, const f32x16 (&acc)[2][2]) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const long long row = row0 + 32 * r + j;
    if (row >= m) continue;
    float* dst = acts_l + row * 256 + 4 * h;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst + 32 * (T0 + t) + 8 * g) =
            make_float4(tf_relu(acc[t][r][4 * g]), tf_relu(acc[t][r][4 * g + 1]), tf_relu(acc[t][r][4 * g + 2]), tf_relu(acc[t][r][4 * g + 3]));
  }
}

// SAVE (64-ray form, training): the three hidden layers' post-ReLU activations are also written, acts[layer][row][256] with row = the
// ray's position in the hit list (capacity rows per layer) -- what the backward pass of the net's dense layers (tf_linear_bwd) reads,
// instead of recomputing the three 256-wide layers with the dense-layer kernels (LightsFn.backward: 1.2 ms of a 12 ms training step).
// TEAMS = 1 (tf_set_launch_budget inner_teams = 1): ONE team per 256-thread workgroup -- one wave per SIMD, 88 KB of LDS, half of
// a CU's register file -- so that the traversal's (or another stream's) waves are resident beside it on every CU and issue in the
// slots this team leaves (its vector steps FE / P1 / P2 leave the matrix pipe idle, its matrix steps the vector unit).  The team runs
// the same six-step cycle; what the partner team hid is now hidden by a different kernel.
// (128-ray form: 32-bit source indices -- the launcher checks the capacity -- : six registers instead of twelve across the matrix steps)
template <int TERMS> using il3_src_t = std::conditional_t<TERMS == 3, long long, unsigned>;
template <bool OUTER, int TERMS, bool SAVE = false, int TEAMS = 2>
__global__ void __launch_bounds__(256 * TEAMS, TEAMS == 1 ? 2 : 1) __attribute__((amdgpu_num_vgpr(256)))
inner_light3_kernel(const float* __restrict__ ws_arg, const float* __restrict__ pts, const float* __restrict__ view,
                    const float* __restrict__ nrm, long long m_arg, const long long* __restrict__ idx,
                    const long long* __restrict__ count_dev, const float* __restrict__ depth, float near_eps, float exp_max,
                    float* __restrict__ out, float* __restrict__ acts = nullptr) {
  static_assert(!SAVE || TERMS == 3, "activations are saved by the 64-ray form");
  typedef IL3<TERMS> C;
  constexpr int RT = C::RT, NR = C::NR, RAYS = C::RAYS;
  long long m = m_arg;
  if (count_dev) m = min(m_arg, *count_dev);
  if (m <= 0) return;
  __shared__ __attribute__((aligned(16))) tf_h8 act[TEAMS * C::TEAM16];     // 128 KB: [team][k-step][ray tile][hi|lo][lane]
  __shared__ __attribute__((aligned(16))) float lbias[3 * 256];             // [layer][lane half][tile * 16 + reg] (128 per half)
  __shared__ __attribute__((aligned(16))) float w4a[3 * 256 + 4];           // [output][lane half][tile * 16 + reg]: rows of the 256 -> 3 layer; + its 3 biases
                                                                            // (as three registers held across the whole kernel they were the first to spill)
  // partial sums of the 256 -> 3 layer, [team][wave][ray tile * 3 + output][ray].  They live from step FE to the tail of step M1,
  // while k-steps 8..15 of the team's image are unused (the input row is 128 columns: k-steps 0..7; layer 3 has been read, layer 1 is
  // published one barrier later): the 128-ray form keeps them THERE (its stage is twice the size and the 160 KB are spent).
  __shared__ float part_own[TERMS == 3 ? TEAMS * C::PART : 4];
  __shared__ __attribute__((aligned(16))) float stage[TEAMS * C::STAGE];    // [team]: hit point | normal | direction rows of the NEXT pass's rays (16 bytes per
                                                                            // ray and array: where a 12-byte LDS-DMA lands) and their depths
  __shared__ il3_src_t<TERMS> stage_src[TEAMS * RAYS];                             // ... and their source indices
  __shared__ __attribute__((aligned(16))) float idem[36 * 20];              // IDE polynomial coefficients [column][power (17, padded to 20)]: read as
                                                                            // broadcast ds_read_b128 (through the scalar cache the 222 coefficients
                                                                            // of a ray went through v_mov copies into packed-FMA operands: 43 spills)
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6), team = TEAMS == 1 ? 0 : wave8 >> 2, w = wave8 & 3;
  for (int i = tid; i < 36 * 20; i += 256 * TEAMS) idem[i] = (i % 20) < 17 ? ws_arg[kIdeMat + (i % 20) * 36 + i / 20] : 0.f;
  for (int i = tid; i < 3 * 256; i += 256 * TEAMS) {
    const int layer = i / 256, r = i % 256;              // packed order (tf_pack_bias_kernel): r = n * 2 + half, n = tile * 16 + reg < 128
    lbias[layer * 256 + (r & 1) * 128 + (r >> 1)] = ws_arg[kIB1 + i];
    w4a[layer * 256 + (r & 1) * 128 + (r >> 1)] = ws_arg[kW4a + i];
  }
  const long long n_pass = (m + RAYS - 1) / RAYS;
  const int n_iter = (int)((n_pass + (long long)TEAMS * gridDim.x - 1) / ((long long)TEAMS * gridDim.x));
  const int T0 = 2 * w;
  tf_h8* actt = act + team * C::TEAM16;
  uint2* act8 = reinterpret_cast<uint2*>(actt);
  float* partt = TERMS == 3 ? part_own + team * C::PART : reinterpret_cast<float*>(actt + 8 * RT * C::XP * 64);
  auto pass_of = [&](int it) { return ((long long)it * gridDim.x + blockIdx.x) * TEAMS + team; };
  // lane l of every wave of a team stands for ray l (and, in the 128-ray form, ray 64 + l) of the team's pass (the four waves split a
  // ray's FEATURES).
  // Input rows: every wave needs all of its ray's inputs, but they are GATHERED once per team and never pass through registers: in the
  // gather step wave 0 / 1 / 2 sends the hit-point / normal / direction row of ray `lane` of the NEXT pass straight into the team's
  // LDS stage by LDS-DMA (one instruction per wave and 64 rays), wave 3 the depth and the source index; the encodings read their ray's
  // rows from there one pass later.  (Gathered by every wave into registers the rows cost four times the random requests and 19
  // registers held across the matrix steps.)
  float* staget = stage + team * C::STAGE;
  il3_src_t<TERMS>* stage_srct = stage_src + team * RAYS;
  typedef il3_src_t<TERMS> src_t;
  src_t src_nxt[NR];                           // source index of ray 64 e + lane of the next pass to gather
  src_t src_cur[NR], src_out[NR];              // ... of the pass in flight / of the pass being stored
  float dep_out[NR], dep_cur[NR];
#pragma unroll
  for (int e = 0; e < NR; ++e) { src_nxt[e] = 0; src_cur[e] = 0; src_out[e] = 0; dep_out[e] = 1.f; dep_cur[e] = 1.f; }
  auto row_src = [&](int it, int q, const long long* ix) {
    long long row = pass_of(it) * RAYS + q;
    if (row >= m) row = m - 1;
    return (src_t)(ix ? ix[row] : row);
  };
  // wave-uniform role: which array this wave sends (w < 3) -- or depth + index (w == 3)
  auto gather_dma = [&](const src_t (&src)[NR], int ln /* the lane index (the per-pass opaque copy inside the pass loop) */) {
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      if (w < 3) {
        // (three uniform branches: a select over the three pointers is lowered to a table in scratch memory)
        const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) float*)(staget + (w * RAYS + 64 * e) * 4));
        if (w == 0) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx3 %1, off" ::"s"(la), "v"(pts + 3 * (long long)src[e]) : "memory");     // lands at 16 bytes per lane
        else if (w == 1) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx3 %1, off" ::"s"(la), "v"(nrm + 3 * (long long)src[e]) : "memory");
        else asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx3 %1, off" ::"s"(la), "v"(view + 3 * (long long)src[e]) : "memory");
      } else {
        if (depth) {
          const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) float*)(staget + 3 * RAYS * 4 + 64 * e));
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" ::"s"(la), "v"(depth + src[e]) : "memory");       // 4 bytes per lane
        } else {
          staget[3 * RAYS * 4 + 64 * e + ln] = 1.f;
        }
        stage_srct[64 * e + ln] = src[e];
      }
    }
  };
#pragma unroll
  for (int e = 0; e < NR; ++e) src_nxt[e] = row_src(0, 64 * e + lane, idx);
  gather_dma(src_nxt, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int e = 0; e < NR; ++e) src_nxt[e] = n_iter > 1 ? row_src(1, 64 * e + lane, idx) : 0;
  std::conditional_t<TERMS == 3, Il3Ring, Il4Ring> ring;
  f32x16 acc[TERMS == 3 ? 2 : 1][RT];      // (128-ray form: one unit tile at a time)
  il4_f2 fsum[TERMS == 3 ? 1 : RT][3];     // 128-ray form: the wave's share of the 256 -> 3 layer (pairs: even | odd units), from step M3 to step FE
  if (tid < 3) w4a[3 * 256 + tid] = ws_arg[kIB4 + 2 * tid];                        // packed order: [n * 2 + half], unit n = reg for n < 4
  __syncthreads();
  // Both teams run the SAME straight-line program; team B passes three barriers before it starts and team A three after it has
  // finished, so that A is three steps ahead at every moment (s_barrier only counts arrivals).  Straight-line code instead of a
  // step machine keeps the register allocator's liveness exact: the accumulators are dead during the encodings, the weight ring
  // between a matrix phase and the next prefetch (as a step machine the kernel spilled 192 registers).
#ifdef TF_DEV   // dev-only: shader-clock stamps of one iteration of workgroup 0, per wave (phase ends and barrier releases), stored as taken
#ifndef IL3_STAMP_MASK
#define IL3_STAMP_MASK 0xffff
#endif
#define IL3_STAMP(i) do { if (((IL3_STAMP_MASK) >> (i)) & 1) { if (it == 40 && blockIdx.x == 0 && lane == 0) g_il3_stamps[wave8 * 16 + (i)] = __builtin_readcyclecounter(); } } while (0)
#else
#define IL3_STAMP(i) do {} while (0)
#endif
  // IL3_POS_EARLY (64-ray form): the positional block of the NEXT pass's encodings (6 sincos pairs per lane and their hi / lo split:
  // ~1.2 k of step FE's cycles) is evaluated in step P2 of the current pass, where this team waits for its partner's M1 anyway -- the
  // rows of the next pass are in the stage since step P1 -- and held as 12 registers of packed halves across M3; step FE, which bounds
  // the steps (FE | M2) by ~1.1 k cycles, only stores them.  Same values, same slots.
  constexpr bool POS_EARLY = IL3_POS_EARLY && TERMS == 3 && !SAVE;        // (the training forward holds the accumulators for il3_save_acts: 12 spills with it)
  uint2 pos_h[3], pos_l[3];
  auto pos_block = [&](int ln) {
    const float4 rp = *reinterpret_cast<const float4*>(staget + 4 * ln);
    const float p[3] = {rp.x, rp.y, rp.z};
    float e12[12];
    const bool small = __all(fabsf(p[0]) < 3.f && fabsf(p[1]) < 3.f && fabsf(p[2]) < 3.f);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float arg = p[i % 3] * (float)(1 << (i / 3)) * (w == 0 ? 1.f : w == 1 ? 4.f : w == 2 ? 16.f : 64.f);   // exact: powers of two
      if (small) tf_sincos_small(arg, e12[2 * i], e12[2 * i + 1]);
      else tf_sincos(arg, e12[2 * i], e12[2 * i + 1]);
    }
#pragma unroll
    for (int gq = 0; gq < 3; ++gq) il3_split4(e12[4 * gq], e12[4 * gq + 1], e12[4 * gq + 2], e12[4 * gq + 3], pos_h[gq], pos_l[gq]);
  };
  if constexpr (POS_EARLY) pos_block(lane);        // pass 0: its rows are in the stage (gathered and waited for above)
  if (TEAMS == 2 && team == 1) { il3_barrier(); il3_barrier(); il3_barrier(); }
  typedef const __attribute__((address_space(1))) tf_h8* gw_t;      // weights are read as GLOBAL loads (a generic pointer makes them flat loads,
                                                                     // which count against lgkmcnt as well and serialise with the LDS reads)
  for (int it = 0; it <= n_iter; ++it) {
    const float* ws = ws_arg;
    asm volatile("" : "+s"(ws));
    const tf_h8* W = reinterpret_cast<const tf_h8*>(ws);
    const bool live = it < n_iter;
    // per-pass opaque copies of the lane index and of the index-array pointer: everything the vector steps derive from them (LDS
    // addresses, shuffle lanes, the direction's sign, octave scales) is recomputed here in a few instructions.  As loop invariants of
    // the pass loop those values were computed once, SPILLED (the matrix phases leave no room) and reloaded in every pass -- and a
    // scratch reload waits for every older vector-memory operation (the gathers) in the in-order vmcnt.
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const long long* idx_o = idx;
    asm volatile("" : "+s"(idx_o));
    const int hh_o = lane_o >> 5;
    const float vsign_o = (idx_o && !OUTER) ? -1.f : 1.f;
    IL3_STAMP(0);
#ifdef TF_DEV
    if (it == 41 && blockIdx.x == 0 && lane == 0) g_il3_stamps[wave8 * 16 + 15] = __builtin_readcyclecounter();
#endif
    // ================= step FE
    // ---- F: the 256 -> 3 layer of the pass whose layer 3 this wave has just finished (64-ray form: from its accumulators; 128-ray form:
    // the sums were formed behind either half of step M3)
    if (it >= 1) {
      float sum[RT][3];
      if constexpr (TERMS == 2) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) sum[r][c] = fsum[r][c][0] + fsum[r][c][1];
      } else {
      if constexpr (SAVE) il3_save_acts(acts + 2 * m_arg * 256, pass_of(it - 1) * RAYS, m, T0, lane_o, acc);
      // (the packed form of the 128-ray path -- pairs of sums, v_pk_fma_f32: il4_out3 -- was measured here in round 6: step F 2.85 k -> 8.3 k
      // cycles, 22.3 -> 24.8 ms; the pairs are assembled with register moves beside 64 live accumulators.  Scalar FMAs stay.)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) sum[r][c] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          float4 wr[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) wr[c] = *reinterpret_cast<const float4*>(w4a + c * 256 + hh_o * 128 + (T0 + t) * 16 + 4 * qd);
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            const float x0 = tf_relu(acc[t][r][4 * qd]), x1 = tf_relu(acc[t][r][4 * qd + 1]);
            const float x2 = tf_relu(acc[t][r][4 * qd + 2]), x3 = tf_relu(acc[t][r][4 * qd + 3]);
#pragma unroll
            for (int c = 0; c < 3; ++c) sum[r][c] = fmaf(x3, wr[c].w, fmaf(x2, wr[c].z, fmaf(x1, wr[c].y, fmaf(x0, wr[c].x, sum[r][c]))));
          }
        }
      }
      if constexpr (TERMS == 2) {
        // the two lane halves hold different units of the same ray.  Two sums per v_permlane32_swap: afterwards one register holds
        // {a's lanes 0..31 | b's lanes 0..31}, the other {a's 32..63 | b's 32..63}; their sum is a's total in lanes 0..31 and b's in
        // lanes 32..63, and all 64 lanes store (rows k, k + 1 of `part` are contiguous) -- no cross-lane read through the LDS pipe,
        // half the stores, in the longest vector step of the cycle
#pragma unroll
        for (int k = 0; k < 3 * RT; k += 2) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum[k / 3][k % 3]), __float_as_uint(sum[(k + 1) / 3][(k + 1) % 3]), false, false);
          partt[(w * 3 * RT + k) * 32 + lane_o] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
      } else {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float tot = sum[r][c] + __shfl_xor(sum[r][c], 32);         // the two lane halves hold different units of the same ray
          if (hh_o == 0) partt[(w * 3 * RT + r * 3 + c) * 32 + (lane_o & 31)] = tot;
        }
      }
#pragma unroll
      for (int e = 0; e < NR; ++e) src_out[e] = src_cur[e];
    }
    IL3_STAMP(13);
    // ---- E: encodings of pass `it`.  A ray's 128 input columns are split over the team's four waves (il3_orig_col): wave w evaluates
    // the sincos pairs of octaves 2w, 2w + 1 (6 of 24) and a quarter of the IDE's polynomial work:
    //   wave 0: IDE columns 0..15      wave 1: 20..23, 32..35      wave 2: 16..19, 28..31 + the zero granule      wave 3: 24..27 + p
    // Every 4-column store granule has one owner.
    {
#pragma unroll
      for (int e = 0; e < NR; ++e) {
        dep_out[e] = dep_cur[e];
        dep_cur[e] = staget[3 * RAYS * 4 + 64 * e + lane_o];
        src_cur[e] = (src_t)stage_srct[64 * e + lane_o];
      }
      if (live) {
        // (128-ray form: the two rays of a lane one after the other in a loop that is NOT unrolled -- the store addresses of a ray are
        // its slot + compile-time constants either way, and the step's registers and code stay those of one ray)
#pragma unroll 1
        for (int e = 0; e < NR; ++e) {
        const int q_o = 64 * e + lane_o;
        // this pass's input rows of the lane's ray: from the team's stage (sent there by LDS-DMA during the previous pass)
        const float4 rp = *reinterpret_cast<const float4*>(staget + 4 * q_o), rn = *reinterpret_cast<const float4*>(staget + RAYS * 4 + 4 * q_o),
                     rv = *reinterpret_cast<const float4*>(staget + 2 * RAYS * 4 + 4 * q_o);
        const float p[3] = {rp.x, rp.y, rp.z};
        float n[3] = {rn.x, rn.y, rn.z};
        float v[3] = {vsign_o * rv.x, vsign_o * rv.y, vsign_o * rv.z};
        // this ray's slot in the team image, made opaque per pass: the 64 store addresses of a pass are this + compile-time constants
        // (immediate offsets); as loop invariants of the pass loop they were each materialised in a register and kept across the
        // matrix phases (93 registers sat unused through a matrix phase; 28-44 spills, whose reloads wait out the gathers)
        int ray_slot = il3_ray_slot8<TERMS>(q_o >> 5, q_o & 31);
        uint2* a8 = act8 + ray_slot;
        if constexpr (POS_EARLY) {
          // the positional block was evaluated and split one step earlier (step P2 of the previous pass / the prologue): only its stores
          // are left -- first, so that the 12 registers are free for the polynomial work below
#pragma unroll
          for (int gq = 0; gq < 3; ++gq) {
            a8[il3_col_slot8<TERMS>(72 + 12 * w + 4 * gq, 0)] = pos_h[gq];
            a8[il3_col_slot8<TERMS>(72 + 12 * w + 4 * gq, 1)] = pos_l[gq];
          }
        }
        float inv = 1.f / fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);
        n[0] *= inv; n[1] *= inv; n[2] *= inv;
        inv = 1.f / fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
        v[0] *= inv; v[1] *= inv; v[2] *= inv;
        const float vn = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
        const float rx = OUTER ? rv.x : vn * n[0] * 2.f - v[0], ry = OUTER ? rv.y : vn * n[1] * 2.f - v[1], rz = OUTER ? rv.z : vn * n[2] * 2.f - v[2];
        // IDE columns [C0, C1) (+ [D0, D1)): sph[col] = (rx + i ry)^mm * sum_q mat[q][col] rz^q; Re -> column col, Im -> 36 + col
        auto ide_cols = [&](auto c0_, auto c1_, auto d0_, auto d1_) {
          constexpr int C0 = decltype(c0_)::value, C1 = decltype(c1_)::value, D0 = decltype(d0_)::value, D1 = decltype(d1_)::value;
          float zp[17], cre[17], cim[17];               // powers the columns of this wave do not use are dead code
          zp[0] = 1.f; cre[0] = 1.f; cim[0] = 0.f;
#pragma unroll
          for (int q = 1; q <= 16; ++q) {
            zp[q] = zp[q - 1] * rz;
            cre[q] = cre[q - 1] * rx - cim[q - 1] * ry;
            cim[q] = cre[q - 1] * ry + cim[q - 1] * rx;
          }
          // granule by granule (4 columns -> 8 values -> two stores), a scheduling barrier behind each: left free the scheduler
          // interleaves all columns' polynomial chains for latency's sake and the step needs more than the 256 registers
#pragma unroll
          for (int gq = 0; gq < 9; ++gq) {
            if (!((4 * gq >= C0 && 4 * gq < C1) || (4 * gq >= D0 && 4 * gq < D1))) continue;
            float re[4], im[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const int col = 4 * gq + cc;
              // (d, mm) of column col: col = (2^d - 1) + d + mm
              const int d = col < 2 ? 0 : col < 5 ? 1 : col < 10 ? 2 : col < 19 ? 3 : 4;
              const int mm = col - ((1 << d) - 1 + d);
              float poly = 0.f;
#pragma unroll
              for (int q4 = 0; q4 <= (1 << d) - mm; q4 += 4) {
                const float4 m4 = *reinterpret_cast<const float4*>(idem + col * 20 + q4);
                poly += zp[q4] * m4.x;
                if (q4 + 1 <= (1 << d) - mm) poly += zp[q4 + 1] * m4.y;
                if (q4 + 2 <= (1 << d) - mm) poly += zp[q4 + 2] * m4.z;
                if (q4 + 3 <= (1 << d) - mm) poly += zp[q4 + 3] * m4.w;
              }
              re[cc] = cre[mm] * poly;
              im[cc] = cim[mm] * poly;
            }
            il3_store4<TERMS>(a8, 4 * gq, re[0], re[1], re[2], re[3]);
            il3_store4<TERMS>(a8, 36 + 4 * gq, im[0], im[1], im[2], im[3]);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        typedef std::integral_constant<int, 0> I0;
        if (w == 0) {
          ide_cols(I0{}, std::integral_constant<int, 16>{}, I0{}, I0{});
        } else if (w == 1) {
          ide_cols(std::integral_constant<int, 20>{}, std::integral_constant<int, 24>{}, std::integral_constant<int, 32>{}, std::integral_constant<int, 36>{});
        } else if (w == 2) {
          ide_cols(std::integral_constant<int, 16>{}, std::integral_constant<int, 20>{}, std::integral_constant<int, 28>{}, std::integral_constant<int, 32>{});
          il3_store4<TERMS>(a8, 124, 0.f, 0.f, 0.f, 0.f);
        } else {
          ide_cols(std::integral_constant<int, 24>{}, std::integral_constant<int, 28>{}, I0{}, I0{});
          il3_store4<TERMS>(a8, 120, p[0], p[1], p[2], 0.f);
        }
        // positional block of wave w: sincos of octaves 2w, 2w + 1, interleaved [sin, cos] per (octave, coordinate) pair
        if constexpr (!POS_EARLY) {
          float e12[12];
          const bool small = __all(fabsf(p[0]) < 3.f && fabsf(p[1]) < 3.f && fabsf(p[2]) < 3.f);
#pragma unroll
          for (int i = 0; i < (TERMS == 2 ? 3 : 6); ++i) {
            const float arg = p[i % 3] * (float)(1 << (i / 3)) * (w == 0 ? 1.f : w == 1 ? 4.f : w == 2 ? 16.f : 64.f);   // exact: powers of two
            if (small) tf_sincos_small(arg, e12[2 * i], e12[2 * i + 1]);
            else tf_sincos(arg, e12[2 * i], e12[2 * i + 1]);
          }
          if constexpr (TERMS == 2) {
            // the wave's second octave by angle doubling from the first (these operands are rounded to f16, 2^-12, on their way into
            // the matrix cores; the doubled pair is within 3e-7 of a range-reduced evaluation): 3 instead of 6 range reductions and
            // polynomial pairs per ray, in the longest vector step of the cycle
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const float sn = e12[2 * i], cs = e12[2 * i + 1];
              e12[2 * (i + 3)] = 2.f * sn * cs;
              e12[2 * (i + 3) + 1] = fmaf(-2.f * sn, sn, 1.f);
            }
          }
#pragma unroll
          for (int gq = 0; gq < 3; ++gq)
            il3_store4<TERMS>(a8, 72 + 12 * w + 4 * gq, e12[4 * gq], e12[4 * gq + 1], e12[4 * gq + 2], e12[4 * gq + 3]);
        }
        }
        IL3_STAMP(14);
        // layer 1's first weight fragments (the ring's registers are free for the encodings above)
        if constexpr (TERMS == 2) il4_prefetch((gw_t)(W + kQ1 / 4), T0, lane, ring);
        else il3_prefetch((gw_t)(W + kQ1 / 4), T0, lane, ring);
      }
    }
    IL3_STAMP(1);
    il3_barrier();
    IL3_STAMP(2);
    // ================= step M1; behind its matrix products, the radiance of the pass finished one step ago (fixed-order sum of the four
    // waves' partial sums).  Behind, not in front: the scattered stores would sit ahead of this step's weight loads in the in-order vmcnt.
    tf_h8 held[TERMS == 3 ? 1 : RT][2];      // 128-ray form: the first half's output (unit tile T0) as packed f16, until the image may be overwritten
    if (live) {
      if constexpr (TERMS == 2) {
        constexpr int K1 = OUTER ? 5 : 8;
        il4_half<K1, true, 0>((gw_t)(W + kQ1 / 4), T0, lane, lbias + hh * 128, actt + lane, ring, acc[0]);
        il4_pack(acc[0], held);
        il4_half<K1, false, K1 % (IL4_PF + 1)>((gw_t)(W + kQ1 / 4), T0 + 1, lane, lbias + hh * 128, actt + lane, ring, acc[0]);
      } else {
      il3_bias<RT>(lbias + hh * 128, T0, acc);
      // (the direction-encoded outer net has 72 input columns: five k-steps, the rest of its image is zero)
      il3_layer<OUTER ? 5 : 8, TERMS>((gw_t)(W + kQ1 / 4), T0, lane, actt + lane, ring, acc);
      }
    }
    // wave w stores ray tile w of the pass (128-ray form: all four waves; 64-ray form: waves 0, 1): the lanes of half w & 1 hold
    // that tile's source indices (ray 64 (w >> 1) + lane)
    auto emit_out = [&]() {
    if (it >= 1 && w < RT && hh_o == (w & 1)) {
      const int e = NR == 1 ? 0 : (w >> 1);
      const long long orow = pass_of(it - 1) * RAYS + 64 * e + lane_o;
      if (orow < m) {
        const float dpo = NR == 1 ? dep_out[0] : (e ? dep_out[NR - 1] : dep_out[0]);
        const long long so = NR == 1 ? (long long)src_out[0] : (long long)(e ? src_out[NR - 1] : src_out[0]);
        const float near = (depth && !(dpo > near_eps)) ? 0.f : 1.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int o = (w * 3 + c) * 32 + (lane_o & 31);
          const float x = ((partt[o] + partt[3 * RT * 32 + o]) + (partt[2 * 3 * RT * 32 + o] + partt[3 * 3 * RT * 32 + o])) + w4a[3 * 256 + c];
          out[3 * so + c] = expf(fminf(x, exp_max)) * near;
        }
      }
    }
    };
    // 64-ray form (IL3_OUT_LATE): not here but in step P1, behind the publish, where this team waits ~5 k cycles for its partner's
    // M3 anyway: waves 0 and 1 spent ~650 cycles on it at the end of M1, the step in which M1 is what the partner's barrier waits for
    // (cycle stamps, round 6).  The partial sums stay valid until step FE of the next pass; the stores sit ~5 k cycles ahead of M2's
    // counted waits
    constexpr bool OUT_LATE = IL3_OUT_LATE && TERMS == 3;
    if (!OUT_LATE || !live) emit_out();
    if (!live) break;
    IL3_STAMP(3);
    il3_barrier();
    IL3_STAMP(4);
    // ================= steps P1 M2 P2 M3
#pragma unroll
    for (int layer = 1; layer < 3; ++layer) {
      // the next layer's first weight fragments: they land while this wave publishes and waits for its partner
      if constexpr (TERMS == 2) il4_prefetch((gw_t)(W + (layer == 1 ? kH2 : kH3) / 4), T0, lane, ring);
      else il3_prefetch((gw_t)(W + (layer == 1 ? kH2 : kH3) / 4), T0, lane, ring);
      // (ahead of the gather: its LDS-DMA is an asm statement with a memory clobber, an LDS read behind it waits for vmcnt(0) --
      // measured: this block took 5 k cycles there, a random-row HBM round trip)
      if (OUT_LATE && layer == 1) emit_out();
      if (layer == 1 && it + 1 < n_iter) {
        // the gather step: BEHIND layer 2's weight prefetch and a publish + barrier wait ahead of the first wait that has to see it
        // retired (vmcnt retires in order: a random-row gather -- an HBM round trip -- in front of a matrix phase's weight loads stalls
        // that phase's first counted wait for the whole round trip)
        gather_dma(src_nxt, lane_o);
        if (it + 2 < n_iter) {
#pragma unroll
          for (int e = 0; e < NR; ++e) src_nxt[e] = row_src(it + 2, 64 * e + lane_o, idx_o);
        }
      }
      if constexpr (TERMS == 2) {
        il4_publish(actt + lane, T0, held);
        il4_pack(acc[0], held);
        il4_publish(actt + lane, T0 + 1, held);
      } else {
        il3_publish<TERMS>(actt + lane, T0, acc);
        if constexpr (SAVE) il3_save_acts(acts + (layer - 1) * m_arg * 256, pass_of(it) * RAYS, m, T0, lane, acc);
        if constexpr (POS_EARLY) { if (layer == 2 && it + 1 < n_iter) pos_block(lane_o); }
      }
      IL3_STAMP(4 * layer + 1);
      il3_barrier();
      IL3_STAMP(4 * layer + 2);
      if constexpr (TERMS == 2) {
        il3_gw_t Wl = (gw_t)(W + (layer == 1 ? kH2 : kH3) / 4);
        const float* lb = lbias + layer * 256 + hh * 128;
        const float* w4h = w4a + hh * 128;
        il4_half<16, true, 0>(Wl, T0, lane, lb, actt + lane, ring, acc[0]);
        if (layer == 1) {
          il4_pack(acc[0], held);
        } else {
#pragma unroll
          for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) fsum[r][c] = il4_f2{0.f, 0.f};
          il4_out3(w4h, T0, acc[0], fsum);
        }
        il4_half<16, false, 16 % (IL4_PF + 1)>(Wl, T0 + 1, lane, lb, actt + lane, ring, acc[0]);
        if (layer == 2) il4_out3(w4h, T0 + 1, acc[0], fsum);
      } else {
      il3_bias<RT>(lbias + layer * 256 + hh * 128, T0, acc);
      il3_layer<16, TERMS>((gw_t)(W + (layer == 1 ? kH2 : kH3) / 4), T0, lane, actt + lane, ring, acc);
      }
      IL3_STAMP(4 * layer + 3);
      il3_barrier();
      IL3_STAMP(4 * layer + 4);
    }
  }
  // `break` above leaves after the FE barrier of the step that has no pass: 6 n_iter + 1 barriers so far for either team
  if (TEAMS == 2 && team == 0) { il3_barrier(); il3_barrier(); il3_barrier(); }
}

// Layer-1 weights with the columns in the order of the column-owned kernel's input row: [IDE (72) | pos_enc8 (51)].
static __global__ void __launch_bounds__(256) inner_light_cols_kernel(const float* __restrict__ W /*[256,123]*/, float* __restrict__ Wp) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 256 * 123) return;
  const int row = e / 123, k = e % 123;
  Wp[e] = W[row * 123 + (k < 72 ? 51 + k : k - 72)];
}

static int inner_light_launch(const TfMlp4* net, const float* pts, const float* view, const float* nrm, int64_t m,
                              const int64_t* idx, const int64_t* count_dev, const float* depth, float near_eps, float exp_max,
                              int32_t precision, float* out, float* workspace, size_t workspace_floats, hipStream_t stream,
                              const char* who, bool outer = false, float* acts = nullptr) {
  TF_REQUIRE(m >= 0, TF_ESHAPE, "%s: m < 0", who);
  const bool packed = (precision & TF_WEIGHTS_PACKED) != 0;
  precision &= ~TF_WEIGHTS_PACKED;
  TF_REQUIRE(precision == TF_PREC_F32 || precision == TF_PREC_F16X3 || precision == TF_PREC_F16 || precision == TF_PREC_F16X2, TF_EINVAL,
             "%s: unknown precision %d", who, precision);
  if (m == 0) return TF_OK;
  TF_REQUIRE(!acts || (precision == TF_PREC_F16X3 && !outer), TF_EINVAL, "%s: activations are saved by the fp32-grade staggered "
             "kernel only (precision TF_PREC_F16X3)", who);
  TF_REQUIRE(!outer || precision == TF_PREC_F16X3 || precision == TF_PREC_F16X2, TF_EINVAL, "%s: the direction-encoded "
             "outer light runs on the staggered kernel only (precision TF_PREC_F16X3 or TF_PREC_F16X2)", who);
  TF_REQUIRE(net && pts && view && nrm && out && workspace, TF_EINVAL, "%s: null pointer", who);
  TF_REQUIRE(workspace_floats >= (size_t)kInnerWsFloats, TF_ESHAPE, "%s: workspace too small (%zu < %d floats)", who,
             workspace_floats, kInnerWsFloats);
  for (int l = 0; l < 4; ++l) TF_REQUIRE(net->w[l] && net->b[l], TF_EINVAL, "%s: null weight pointer (layer %d)", who, l);
  if (!packed) {
    if (precision == TF_PREC_F32) {
      tf_pack_wfrag_kernel<<<tf_blocks(8 * 64 * 64, 256), 256, 0, stream>>>(net->w[0], 256, 123, 0, 123, 8, 64, workspace + kI1, 1);
      tf_pack_wfrag_kernel<<<tf_blocks(8 * 128 * 64, 256), 256, 0, stream>>>(net->w[1], 256, 256, 0, 256, 8, 128, workspace + kI2, 1);
      tf_pack_wfrag_kernel<<<tf_blocks(8 * 128 * 64, 256), 256, 0, stream>>>(net->w[2], 256, 256, 0, 256, 8, 128, workspace + kI3, 1);
      tf_pack_wfrag_kernel<<<tf_blocks(128 * 64, 256), 256, 0, stream>>>(net->w[3], 3, 256, 0, 256, 1, 128, workspace + kI4, 1);
    } else {
      _Float16* hw = reinterpret_cast<_Float16*>(workspace);
      tf_pack_wfrag_h3_kernel<<<tf_blocks(8 * 16 * 64, 256), 256, 0, stream>>>(net->w[1], 256, 256, 0, 256, 8, 16, hw + 2 * (size_t)kH2);
      tf_pack_wfrag_h3_kernel<<<tf_blocks(8 * 16 * 64, 256), 256, 0, stream>>>(net->w[2], 256, 256, 0, 256, 8, 16, hw + 2 * (size_t)kH3);
      tf_pack_wfrag_h3_kernel<<<tf_blocks(1 * 16 * 64, 256), 256, 0, stream>>>(net->w[3], 3, 256, 0, 256, 1, 16, hw + 2 * (size_t)kH4);
      if (precision == TF_PREC_F16) {       // the column-owned kernel's first layer: IDE columns first (kP1)
        inner_light_cols_kernel<<<tf_blocks(256 * 123, 256), 256, 0, stream>>>(net->w[0], workspace + kWp);
        tf_pack_wfrag_h3_kernel<<<tf_blocks(8 * 8 * 64, 256), 256, 0, stream>>>(workspace + kWp, 256, 123, 0, 123, 8, 8, hw + 2 * (size_t)kP1);
      } else {                              // the staggered kernel's: its four waves' column order (kQ1; the outer net's matrix is [256,72])
        inner_light_cols3_kernel<<<tf_blocks(256 * 128, 256), 256, 0, stream>>>(net->w[0], outer ? 72 : 123, workspace + kWp);
        tf_pack_wfrag_h3_kernel<<<tf_blocks(8 * 8 * 64, 256), 256, 0, stream>>>(workspace + kWp, 256, 128, 0, 128, 8, 8, hw + 2 * (size_t)kQ1);
      }
    }
    tf_pack_bias_kernel<<<1, 256, 0, stream>>>(net->b[0], 256, 8, workspace + kIB1);
    tf_pack_bias_kernel<<<1, 256, 0, stream>>>(net->b[1], 256, 8, workspace + kIB2);
    tf_pack_bias_kernel<<<1, 256, 0, stream>>>(net->b[2], 256, 8, workspace + kIB3);
    tf_pack_bias_kernel<<<1, 256, 0, stream>>>(net->b[3], 3, 1, workspace + kIB4);
    for (int c = 0; c < 3; ++c)      // rows of the 256 -> 3 layer in accumulator order (staggered f16x3 kernel: that layer runs on the vector unit)
      tf_pack_bias_kernel<<<1, 256, 0, stream>>>(net->w[3] + c * 256, 256, 8, workspace + kW4a + c * 256);
    hipError_t e = hipMemcpyAsync(workspace + kIdeMat, ide_tables_cached(), 17 * 36 * sizeof(float), hipMemcpyHostToDevice, stream);
    TF_REQUIRE(e == hipSuccess, TF_EHIP, "%s: hipMemcpyAsync failed: %s", who, hipGetErrorString(e));
  }
#define IL_ARGS workspace, pts, view, nrm, m, (const long long*)idx, (const long long*)count_dev, depth, near_eps, exp_max, out
  // Which kernel serves which TfPrecision (this file's header comment): F16X3 / F16X2 the staggered kernel below, F32 / F16 the two
  // kernels of inner_light_modes.hip
  const bool one_team = tf_launch_budget().inner_teams == 1 && !outer && !acts;
  // one single-team workgroup per CU: two would fit (81.7 KB of LDS, 256 registers per wave) and leave nothing for the kernel this
  // budget makes room for, so the launch asks for 8 KB of (unused) dynamic LDS on top
  static const bool il1_nopad = getenv("TF_IL1_NOPAD") != nullptr;      // dev: two unsynchronised single-team workgroups per CU
  const size_t il1_pad = il1_nopad ? 0 : 8192;
  const long long team_cap = one_team && il1_nopad ? 512 : 256;
  if (precision == TF_PREC_F16X3) {
    // staggered two-team kernel: one 512-thread workgroup per CU, two 64-ray passes in flight
    long long blocks = one_team ? (m + 63) / 64 : ((m + 63) / 64 + 1) / 2;
    if (blocks > team_cap) blocks = team_cap;
    if (one_team) inner_light3_kernel<false, 3, false, 1><<<(unsigned)blocks, 256, il1_pad, stream>>>(IL_ARGS);
    else if (outer) inner_light3_kernel<true, 3><<<(unsigned)blocks, 512, 0, stream>>>(IL_ARGS);
    else if (acts) inner_light3_kernel<false, 3, true><<<(unsigned)blocks, 512, 0, stream>>>(IL_ARGS, acts);
    else inner_light3_kernel<false, 3><<<(unsigned)blocks, 512, 0, stream>>>(IL_ARGS);
  } else if (precision == TF_PREC_F16X2) {
    // ... its 128-ray form: two 128-ray passes in flight (source indices are carried as 32-bit values)
    TF_REQUIRE(m <= 0x7fffffffLL, TF_ESHAPE, "%s: more than 2^31 - 1 rays in one call", who);
    long long blocks = one_team ? (m + 127) / 128 : ((m + 127) / 128 + 1) / 2;
    if (blocks > team_cap) blocks = team_cap;
    if (one_team) inner_light3_kernel<false, 2, false, 1><<<(unsigned)blocks, 256, il1_pad, stream>>>(IL_ARGS);
    else if (outer) inner_light3_kernel<true, 2><<<(unsigned)blocks, 512, 0, stream>>>(IL_ARGS);
    else inner_light3_kernel<false, 2><<<(unsigned)blocks, 512, 0, stream>>>(IL_ARGS);
  } else if (precision == TF_PREC_F16) {
    tf_inner_light_launch_f16(IL_ARGS, stream);
  } else {
    tf_inner_light_launch_f32(IL_ARGS, stream);
  }
#undef IL_ARGS
  TF_LAUNCH_CHECK(who);
  return TF_OK;
}

extern "C" int tf_inner_light_fwd(const TfMlp4* net, const float* pts, const float* view, const float* nrm, int64_t m,
                                  float exp_max, int32_t precision, float* out, float* workspace, size_t workspace_floats,
                                  tf_stream_t stream) {
  return inner_light_launch(net, pts, view, nrm, m, nullptr, nullptr, nullptr, 0.f, exp_max, precision, out, workspace,
                            workspace_floats, (hipStream_t)stream, "tf_inner_light_fwd");
}

extern "C" int tf_inner_light_indexed_fwd(const TfMlp4* net, const float* pos, const float* dirs, const float* nrm,
                                          const int64_t* idx, const int64_t* count_dev, int64_t capacity, const float* depth,
                                          float near_eps, float exp_max, int32_t precision, float* lights, float* workspace,
                                          size_t workspace_floats, tf_stream_t stream) {
  TF_REQUIRE(capacity == 0 || (idx && count_dev), TF_EINVAL, "tf_inner_light_indexed_fwd: idx / count_dev is null");
  return inner_light_launch(net, pos, dirs, nrm, capacity, idx, count_dev, depth, near_eps, exp_max, precision, lights, workspace,
                            workspace_floats, (hipStream_t)stream, "tf_inner_light_indexed_fwd");
}

extern "C" int tf_inner_light_indexed_train_fwd(const TfMlp4* net, const float* pos, const float* dirs, const float* nrm,
                                                const int64_t* idx, const int64_t* count_dev, int64_t capacity, const float* depth,
                                                float near_eps, float exp_max, int32_t precision, float* lights, float* acts,
                                                float* workspace, size_t workspace_floats, tf_stream_t stream) {
  TF_REQUIRE(capacity == 0 || (idx && count_dev), TF_EINVAL, "tf_inner_light_indexed_train_fwd: idx / count_dev is null");
  TF_REQUIRE(acts, TF_EINVAL, "tf_inner_light_indexed_train_fwd: acts is null");
  return inner_light_launch(net, pos, dirs, nrm, capacity, idx, count_dev, depth, near_eps, exp_max, precision, lights, workspace,
                            workspace_floats, (hipStream_t)stream, "tf_inner_light_indexed_train_fwd", false, acts);
}

extern "C" int tf_outer_light_indexed_fwd(const TfMlp4* net, const float* dirs, const int64_t* idx, const int64_t* count_dev,
                                         int64_t capacity, float exp_max, int32_t precision, float* lights, float* workspace,
                                         size_t workspace_floats, tf_stream_t stream) {
  TF_REQUIRE(capacity == 0 || (idx && count_dev), TF_EINVAL, "tf_outer_light_indexed_fwd: idx / count_dev is null");
  return inner_light_launch(net, dirs, dirs, dirs, capacity, idx, count_dev, nullptr, 0.f, exp_max, precision, lights, workspace,
                            workspace_floats, (hipStream_t)stream, "tf_outer_light_indexed_fwd", true);
}

// ---------------------------------------------------------------- input encoding only (training: the weight-gradient
// GEMMs [rows x 256]^T [rows x 256] are plain library GEMMs on this [rows,123] matrix)
__global__ void __launch_bounds__(256) inner_light_encode_kernel(const float* __restrict__ mat, const float* __restrict__ pts,
                                                                 const float* __restrict__ view, const float* __restrict__ nrm,
                                                                 long long m_arg, const long long* __restrict__ idx,
                                                                 const long long* __restrict__ count_dev, float* __restrict__ X, int ld) {
  long long m = m_arg;
  if (count_dev) m = min(m_arg, *count_dev);
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= m) return;
  const long long src = idx ? idx[row] : row;
  const float vsign = idx ? -1.f : 1.f;
  float* x = X + row * ld;
  for (int k = 123; k < ld; ++k) x[k] = 0.f;
  const float p[3] = {pts[3 * src], pts[3 * src + 1], pts[3 * src + 2]};
  for (int k = 0; k < 3; ++k) x[k] = p[k];
  for (int f = 0; f < 8; ++f)
    for (int k = 0; k < 3; ++k) {
      const float a = p[k] * (float)(1 << f);
      x[3 + 6 * f + k] = sinf(a);
      x[3 + 6 * f + 3 + k] = cosf(a);
    }
  float n[3] = {nrm[3 * src], nrm[3 * src + 1], nrm[3 * src + 2]};
  float v[3] = {vsign * view[3 * src], vsign * view[3 * src + 1], vsign * view[3 * src + 2]};
  float inv = 1.f / fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);
  n[0] *= inv; n[1] *= inv; n[2] *= inv;
  inv = 1.f / fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
  const float vn = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
  const float rx = vn * n[0] * 2.f - v[0], ry = vn * n[1] * 2.f - v[1], rz = vn * n[2] * 2.f - v[2];
  float zp[17], cre[17], cim[17];
  zp[0] = 1.f; cre[0] = 1.f; cim[0] = 0.f;
  for (int k = 1; k < 17; ++k) {
    zp[k] = zp[k - 1] * rz;
    cre[k] = cre[k - 1] * rx - cim[k - 1] * ry;
    cim[k] = cre[k - 1] * ry + cim[k - 1] * rx;
  }
  int col = 0;
  for (int d = 0; d < 5; ++d) {
    const int l = 1 << d;
    for (int mm = 0; mm <= l; ++mm, ++col) {
      float poly = 0.f;
      for (int k = 0; k <= l - mm; ++k) poly += zp[k] * mat[k * 36 + col];
      x[51 + col] = cre[mm] * poly;
      x[51 + 36 + col] = cim[mm] * poly;
    }
  }
}

extern "C" int tf_inner_light_encode(const float* pos, const float* dirs, const float* nrm, const int64_t* idx,
                                     const int64_t* count_dev, int64_t capacity, float* X, int32_t ld, float* workspace,
                                     size_t workspace_floats, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(capacity >= 0 && ld >= 123, TF_ESHAPE, "tf_inner_light_encode: capacity < 0 or ld < 123");
  if (capacity == 0) return TF_OK;
  TF_REQUIRE(pos && dirs && nrm && X && workspace, TF_EINVAL, "tf_inner_light_encode: null pointer");
  TF_REQUIRE(workspace_floats >= (size_t)kInnerWsFloats, TF_ESHAPE, "tf_inner_light_encode: workspace too small");
  hipError_t e = hipMemcpyAsync(workspace + kIdeMat, ide_tables_cached(), 17 * 36 * sizeof(float), hipMemcpyHostToDevice, stream);
  TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_inner_light_encode: hipMemcpyAsync failed: %s", hipGetErrorString(e));
  inner_light_encode_kernel<<<tf_blocks(capacity, 256), 256, 0, stream>>>(workspace + kIdeMat, pos, dirs, nrm, capacity,
                                                                         (const long long*)idx, (const long long*)count_dev, X, ld);
  TF_LAUNCH_CHECK("tf_inner_light_encode");
  return TF_OK;
}

// ---------------------------------------------------------------- stream compaction of a byte mask (hit rays)
#define COMPACT_ITEMS 8192   // elements per workgroup: one global atomic per 8192 rays instead of one per wave
// One pass over the mask: a thread owns 32 consecutive bytes (two 16-byte loads, kept in registers), the workgroup's offsets come
// from a wave scan + 4 LDS words, the indices of a thread are written as one run.  (The first version read single bytes twice:
// 64-byte wave loads, 0.45 ms per 201 M rays.)  The order of `idx` across workgroups follows the atomic, as before: consumers
// scatter through it, no result depends on it.
__global__ void __launch_bounds__(256) compact_mask_kernel(const unsigned char* __restrict__ mask, long long m,
                                                           long long* __restrict__ idx, unsigned long long* __restrict__ count) {
  __shared__ unsigned int s_wave[4];
  __shared__ unsigned long long s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long first = (long long)blockIdx.x * COMPACT_ITEMS + 32LL * threadIdx.x;
  unsigned int w[8];
  if (first + 32 <= m && (((uintptr_t)mask) & 15) == 0) {
    const uint4 a = *reinterpret_cast<const uint4*>(mask + first), b = *reinterpret_cast<const uint4*>(mask + first + 16);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      unsigned int v = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long long i = first + 4 * q + e;
        if (i < m) v |= (unsigned int)mask[i] << (8 * e);
      }
      w[q] = v;
    }
  }
  // bit j of `bits` = byte j of the 32 is non-zero
  unsigned int bits = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) bits |= ((w[q] >> (8 * e)) & 0xffu) ? (1u << (4 * q + e)) : 0u;
  const unsigned int mine = __popc(bits);
  unsigned int incl = mine;                       // inclusive scan over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = total ? atomicAdd(count, (unsigned long long)total) : 0ULL;
  }
  __syncthreads();
  unsigned long long at = s_base + (incl - mine);
  for (int k = 0; k < wave; ++k) at += s_wave[k];
  while (bits) {
    const int j = __ffs(bits) - 1;
    bits &= bits - 1;
    idx[at++] = first + j;
  }
}

// The same compaction with the mask taken from a float array: element i is kept iff v[i] < thr (a ray hit iff its depth is below the
// miss value: the traversal then need not store a separate flag byte per ray -- 0.6 ms of scattered one-byte stores per 201 M rays).
__global__ void __launch_bounds__(256) compact_below_kernel(const float* __restrict__ v, float thr, long long m,
                                                            long long* __restrict__ idx, unsigned long long* __restrict__ count) {
  __shared__ unsigned int s_wave[4];
  __shared__ unsigned long long s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a wave reads 4 x 2 KB runs: lane l owns elements [first + 128 q + 4 l', ...): keep it simple -- 32 CONSECUTIVE floats per thread,
  // fetched as eight 16-byte loads (a wave's loads cover 8 KB contiguous)
  const long long first = (long long)blockIdx.x * COMPACT_ITEMS + 32LL * threadIdx.x;
  unsigned int bits = 0;
  if (first + 32 <= m && (((uintptr_t)v) & 15) == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(v + first + 4 * q);
      bits |= (a.x < thr ? 1u : 0u) << (4 * q) | (a.y < thr ? 2u : 0u) << (4 * q) | (a.z < thr ? 4u : 0u) << (4 * q) | (a.w < thr ? 8u : 0u) << (4 * q);
    }
  } else {
    for (int j = 0; j < 32; ++j)
      if (first + j < m && v[first + j] < thr) bits |= 1u << j;
  }
  const unsigned int mine = __popc(bits);
  unsigned int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = total ? atomicAdd(count, (unsigned long long)total) : 0ULL;
  }
  __syncthreads();
  unsigned long long at = s_base + (incl - mine);
  for (int k = 0; k < wave; ++k) at += s_wave[k];
  while (bits) {
    const int j = __ffs(bits) - 1;
    bits &= bits - 1;
    idx[at++] = first + j;
  }
}

extern "C" int tf_compact_below(const float* v, float thr, int64_t m, int64_t* idx, int64_t* count, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(m >= 0, TF_ESHAPE, "tf_compact_below: m < 0");
  TF_REQUIRE(count, TF_EINVAL, "tf_compact_below: count is null");
  hipError_t e = hipMemsetAsync(count, 0, sizeof(int64_t), stream);
  TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_compact_below: hipMemsetAsync failed: %s", hipGetErrorString(e));
  if (m == 0) return TF_OK;
  TF_REQUIRE(v && idx, TF_EINVAL, "tf_compact_below: null pointer");
  compact_below_kernel<<<tf_blocks(m, COMPACT_ITEMS), 256, 0, stream>>>(v, thr, m, (long long*)idx, (unsigned long long*)count);
  TF_LAUNCH_CHECK("tf_compact_below");
  return TF_OK;
}

extern "C" int tf_compact_mask(const uint8_t* mask, int64_t m, int64_t* idx, int64_t* count, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(m >= 0, TF_ESHAPE, "tf_compact_mask: m < 0");
  TF_REQUIRE(count, TF_EINVAL, "tf_compact_mask: count is null");
  hipError_t e = hipMemsetAsync(count, 0, sizeof(int64_t), stream);
  TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_compact_mask: hipMemsetAsync failed: %s", hipGetErrorString(e));
  if (m == 0) return TF_OK;
  TF_REQUIRE(mask && idx, TF_EINVAL, "tf_compact_mask: null pointer");
  compact_mask_kernel<<<tf_blocks(m, COMPACT_ITEMS), 256, 0, stream>>>(mask, m, (long long*)idx, (unsigned long long*)count);
  TF_LAUNCH_CHECK("tf_compact_mask");
  return TF_OK;
}
