// Backward of tf_flow_logq_fwd: gradient of sum_r g_logq[r] * logq[r] with respect to the 16 tensors of the two coupling
// nets and to the per-point condition vectors.  This is the training direction of TensoFlow (the NIS loss,
// network/fields.py:1257-1333 -> TensoFlow.forward, network/flow.py:801-831); the reference gets it from autograd through
// ~200 tiny kernels.  One launch here, per 32-row tile and wave:
//   1. forward recompute (fp32 MFMA, weights in LDS) keeping the three hidden activations of each net in registers;
//   2. closed-form reverse pass through the two piecewise-quadratic splines and the prior term;
//   3. delta propagation W^T * delta on the MFMA (transposed fragments streamed from L2);
//   4. weight gradients delta * h^T on the MFMA: both tiles are transposed through a per-wave LDS scratch so that the rows of
//      the tile become the MFMA k dimension; every wave OWNS gradient blocks of a net (two whole 32x32 blocks of dW3 / dW2 and half
//      the rows of one block each of dW4 / dW1), accumulates them over the four tiles of its workgroup (128 rows per step, two
//      workgroup barriers per layer) in registers across the whole kernel and stores them once (round 3; per-tile float atomics
//      into the workgroup's slice cost a third of the kernel, LDS float atomics a fifth: 1.99 -> 1.15 ms per 262 k rows);
//   5. bias gradients reduced over the tile's rows in LDS and kept in registers like the blocks; per-point (hoisted layer-1)
//      gradients reduced the same way, one atomic per unit and tile, issued where no fragment fetch queues behind it.
// Round 5 (cycle stamps of one tile: 197 k cycles for 90 k cycles of matrix steps, ONE wave per SIMD): the transposed fragments are
// global_load (the flat_load the compiler chose counts against the LDS counter too: every LDS wait also waited for the L2), every
// product prefetches its next group of fragments (tf_layer_pf) and its first group ahead of the stage's barriers, the forward's own
// z[:,0] replaces the first of three net evaluations (tf_flow_logq_bwd's `z`), per-tile bias atomics are gone (a fragment fetch
// issued after a float atomic waits for it: the memory counter is in order) and the two half-empty stages (dW4: two blocks, dW1:
// two blocks, four waves) split their blocks' rows over all four waves: 1.01 -> see DESIGN.md ms per 262 k rows.
#include "mfma_mlp.h"
#include "flow_image.h"
#include "tf_common.h"

#define FLOW_NB 10
static constexpr float kEps32 = 1.1920928955078125e-07f;
static constexpr float kHalfPi = 1.5707963267948966f;

// transposed fragments per net (global): T4 = W4^T [64 x 21->32], T3, T2 = W^T [64 x 64], T1 = W1s^T [8->32 x 64]
static constexpr int kT4 = 0, kT3 = kT4 + 2 * 16 * 64, kT2 = kT3 + 2 * 32 * 64, kT1 = kT2 + 2 * 32 * 64, kTNet = kT1 + 1 * 32 * 64;
static constexpr int kFwdWs = 2 * hNetFloats;               // f16x3 forward images of both nets
static constexpr int kBwdWs = kFwdWs + 2 * kTNet;            // + P [2][pn][64]
static constexpr int kTileLds = 64 * 33;                     // one transposed [64 units][32 rows (+1 pad)] tile

// Weight / bias gradients are accumulated per WORKGROUP in a private slice of the workspace and folded into the caller's
// tensors by a second launch: every tile adds all ~25 k gradient values, and with all workgroups adding into the same
// 100 KB the float atomics ran at a third of their rate (memory-side atomics serialise on an address).
//   slice (per net, floats): w0 [64,44] | w1 [64,64] | w2 [64,64] | w3 [21,64] | b1 64 | b2 64 | b3 32
static constexpr int kGW0 = 0, kGW1 = kGW0 + 64 * 44, kGW2 = kGW1 + 64 * 64, kGW3 = kGW2 + 64 * 64, kGB1 = kGW3 + 21 * 64,
                     kGB2 = kGB1 + 64, kGB3 = kGB2 + 64, kGNet = kGB3 + 32, kGradFloats = 2 * kGNet;
static constexpr int kBwdBlocks = 256;   // persistent workgroups (one per CU)

extern "C" size_t tf_flow_bwd_workspace_floats(int64_t pn) {
  return (size_t)kBwdWs + (size_t)2 * 64 * (size_t)(pn > 0 ? pn : 0) + (size_t)kBwdBlocks * kGradFloats;
}

struct FlowGrads {
  float* w[2][4];
  float* b[2][4];
  float* gP;       // [2][pn][64]
  float* slices;   // [kBwdBlocks][kGradFloats]
};

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : 0.01f * x; }
__device__ __forceinline__ float dleaky(float h) { return h > 0.f ? 1.f : 0.01f; }   // sign(h) == sign(pre-activation)

#pragma clang fp contract(off)
struct PwT {
  float e[FLOW_NB], w[FLOW_NB], wn[FLOW_NB], wss[FLOW_NB + 1], C[FLOW_NB + 1], S;
  float ev[FLOW_NB + 1], vn[FLOW_NB + 1], v[FLOW_NB + 1], vw[FLOW_NB + 1], den;
};

__device__ __forceinline__ void pw_tables_fwd(const float (&wv)[32], PwT& T) {
  float run = 0.f;
  T.C[0] = 0.f;
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) {
    T.e[i] = fmaxf(expf(wv[11 + i]), 1e-6f);
    run += T.e[i];
    T.C[i + 1] = run;
  }
  T.S = run;
  T.wss[0] = 0.f;
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) {
    T.wn[i] = T.e[i] / T.S;
    T.w[i] = fmaxf(T.wn[i], 1e-6f);
    T.wss[i + 1] = T.C[i + 1] / T.S;
  }
#pragma unroll
  for (int i = 0; i <= FLOW_NB; ++i) T.ev[i] = expf(wv[i]);
  float den = 0.f;
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) den += (T.ev[i] + T.ev[i + 1]) / 2.f * T.w[i];
  T.den = den;
  T.vw[0] = 0.f;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i <= FLOW_NB; ++i) { T.vn[i] = T.ev[i] / den; T.v[i] = fmaxf(T.vn[i], 1e-6f); }
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) { acc += (T.v[i] + T.v[i + 1]) / 2.f * T.w[i]; T.vw[i + 1] = acc; }
}

#define PICK(arr, n, idx, dst) { dst = arr[0]; _Pragma("unroll") for (int q_ = 1; q_ < (n); ++q_) dst = (idx == q_) ? arr[q_] : dst; }

// forward spline (density direction) + its reverse pass: returns out, logj; given upstream (g_out, g_logj) adds d/d wv.
__device__ __forceinline__ void pw_forward_fb(float xin, const float (&wv)[32], float& out, float& logj) {
  PwT T;
  pw_tables_fwd(wv, T);
  int cnt = 0;
  {
    float best = kEps32;
#pragma unroll
    for (int i = 1; i <= FLOW_NB; ++i) { float val = (T.wss[i] > xin) ? 0.f : T.wss[i]; if (val > best) { best = val; cnt = i; } }
  }
  const int m = min(max(cnt, 0), FLOW_NB - 1), m1 = m + 1;
  float vm, vm1, wm, vwm, wssm;
  PICK(T.v, FLOW_NB + 1, m, vm) PICK(T.v, FLOW_NB + 1, m1, vm1) PICK(T.w, FLOW_NB, m, wm) PICK(T.vw, FLOW_NB + 1, m, vwm)
  PICK(T.wss, FLOW_NB + 1, m, wssm)
  const float al = fminf(fmaxf((xin - wssm) / wm, 0.f), 1.f);
  const float o = (al * al) / 2.f * ((vm1 - vm) * wm) + al * vm * wm + vwm;
  out = fminf(fmaxf(o, kEps32), 1.f - kEps32);
  const float lerp = al < 0.5f ? vm + al * (vm1 - vm) : vm1 - (vm1 - vm) * (1.f - al);
  logj = logf(lerp);
}

// -> gradient wrt the transformed coordinate xin itself (d out / d xin = lerp = exp(logj), d logj / d xin = dv / (w lerp): both through al)
__device__ __forceinline__ float pw_forward_bwd(float xin, const float (&wv)[32], float g_out, float g_logj, float (&g_wv)[32]) {
  PwT T;
  pw_tables_fwd(wv, T);
  int cnt = 0;
  {
    float best = kEps32;
#pragma unroll
    for (int i = 1; i <= FLOW_NB; ++i) { float val = (T.wss[i] > xin) ? 0.f : T.wss[i]; if (val > best) { best = val; cnt = i; } }
  }
  const int m = min(max(cnt, 0), FLOW_NB - 1), m1 = m + 1;
  float vm, vm1, wm, vwm, wssm;
  PICK(T.v, FLOW_NB + 1, m, vm) PICK(T.v, FLOW_NB + 1, m1, vm1) PICK(T.w, FLOW_NB, m, wm) PICK(T.vw, FLOW_NB + 1, m, vwm)
  PICK(T.wss, FLOW_NB + 1, m, wssm)
  const float araw = (xin - wssm) / wm;
  const float al = fminf(fmaxf(araw, 0.f), 1.f);
  const float dv = vm1 - vm;
  const float o = (al * al) / 2.f * (dv * wm) + al * vm * wm + vwm;
  const float go = (o >= kEps32 && o <= 1.f - kEps32) ? g_out : 0.f;   // clamp(eps, 1-eps) passes gradient inside only
  const float lerp = vm + al * dv;
  const float g_lerp = g_logj / lerp;
  float g_vm = g_lerp * (1.f - al) + go * al * wm - go * (al * al) / 2.f * wm;
  float g_vm1 = g_lerp * al + go * (al * al) / 2.f * wm;
  float g_al = g_lerp * dv + go * (al * dv * wm + vm * wm);
  float g_wm = go * ((al * al) / 2.f * dv + al * vm);
  const float g_vwm = go;
  if (!(araw >= 0.f && araw <= 1.f)) g_al = 0.f;
  const float g_wssm = -g_al / wm;
  const float g_xin = g_al / wm;
  g_wm += -g_al * araw / wm;
  // scatter the picked gradients back into per-index arrays
  float g_v[FLOW_NB + 1], g_w[FLOW_NB];
#pragma unroll
  for (int i = 0; i <= FLOW_NB; ++i) g_v[i] = (i == m ? g_vm : 0.f) + (i == m1 ? g_vm1 : 0.f);
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) g_w[i] = (i == m ? g_wm : 0.f);
  // vw[m] = sum_{i<m} (v_i + v_{i+1})/2 * w_i
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) {
    const float on = i < m ? g_vwm : 0.f;
    g_v[i] += on * T.w[i] / 2.f;
    g_v[i + 1] += on * T.w[i] / 2.f;
    g_w[i] += on * (T.v[i] + T.v[i + 1]) / 2.f;
  }
  // v = max(ev/den, 1e-6)
  float g_ev[FLOW_NB + 1];
  float g_den = 0.f;
#pragma unroll
  for (int i = 0; i <= FLOW_NB; ++i) {
    const float g_vn = T.vn[i] > 1e-6f ? g_v[i] : 0.f;
    g_ev[i] = g_vn / T.den;
    g_den -= g_vn * T.vn[i] / T.den;
  }
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) {
    g_ev[i] += g_den * T.w[i] / 2.f;
    g_ev[i + 1] += g_den * T.w[i] / 2.f;
    g_w[i] += g_den * (T.ev[i] + T.ev[i + 1]) / 2.f;
  }
#pragma unroll
  for (int i = 0; i <= FLOW_NB; ++i) g_wv[i] += g_ev[i] * T.ev[i];
  // w = max(e/S, 1e-6); wss[k] = C[k]/S
  float g_e[FLOW_NB];
  float g_S = 0.f;
  float Cm;
  PICK(T.C, FLOW_NB + 1, m, Cm)
  const float g_Cm = g_wssm / T.S;
  g_S -= g_wssm * Cm / (T.S * T.S);
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) {
    const float g_wn = T.wn[i] > 1e-6f ? g_w[i] : 0.f;
    g_e[i] = g_wn / T.S + (i < m ? g_Cm : 0.f);
    g_S -= g_wn * T.e[i] / (T.S * T.S);
  }
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) {
    const float ge = g_e[i] + g_S;
    g_wv[11 + i] += (T.e[i] > 1e-6f) ? ge * T.e[i] : 0.f;
  }
  return g_xin;
}
#pragma clang fp contract(fast)

// forward net keeping the hidden activations; P-row + sample embed -> wv[32]
// `net`: the f16x3 fragment image in LDS (flow_image.h).  The re-evaluation runs with split f16 operands like the product's forward
// (fp32-grade: 22 significant bits per operand, fp32 accumulation; 3 x 32-cycle matrix steps per 16 inputs instead of 8 x 64):
// round 5, 15.1 k -> see DESIGN.md cycles per 32-row tile and net.  The reverse pass below stays on exact-fp32 products: its
// operands are gradients, whose range (the caller's loss scale) f16 does not cover.
__device__ __forceinline__ void net_fwd_keep(const float* __restrict__ net, const float* __restrict__ Prow, const float (&in8)[8],
                                             int lane, f32x16 (&in1)[1], f32x16 (&h1)[2], f32x16 (&h2)[2], f32x16 (&h3)[2],
                                             float (&wv)[32]) {
  const int h = lane >> 5;
  const tf_h8* nh = reinterpret_cast<const tf_h8*>(net) + lane;
#pragma unroll
  for (int j = 0; j < 16; ++j) in1[0][j] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) in1[0][j] = h ? in8[4 + j] : in8[j];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) h1[t][j] = Prow[32 * t + tf_rho(j, h)];
  tf_layer_h3<1, 2, 1>(nh + hL1 / 4, in1, h1);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) { h1[t][j] = leaky(h1[t][j]); h2[t][j] = net[hB2 + (t * 16 + j) * 2 + h]; }
  tf_layer_h3<4, 2, 2>(nh + hL2 / 4, h1, h2);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) { h2[t][j] = leaky(h2[t][j]); h3[t][j] = net[hB3 + (t * 16 + j) * 2 + h]; }
  tf_layer_h3<4, 2, 2>(nh + hL3 / 4, h2, h3);
  f32x16 o[1];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) h3[t][j] = leaky(h3[t][j]);
#pragma unroll
  for (int j = 0; j < 16; ++j) o[0][j] = net[hB4 + j * 2 + h];
  tf_layer_h3<4, 1, 2>(nh + hL4 / 4, h3, o);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float mine = o[0][j], other = __shfl_xor(mine, 32);
    const int r0 = (j & 3) + 8 * (j >> 2);
    wv[r0] = h ? other : mine;
    wv[r0 + 4] = h ? mine : other;
  }
}

// tile (accumulator layout, TT unit tiles) -> LDS [unit][row] with a 33-float row stride
template <int TT>
__device__ __forceinline__ void tile_to_lds(const f32x16 (&x)[TT], float* __restrict__ l, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) l[(32 * t + tf_rho(j, h)) * 33 + r] = x[t][j];
}

// One OWNED 32 x 32 block of delta * hin^T over the four tiles of the workgroup (128 rows): the operands come from every wave's LDS
// transposes, the block stays in this wave's registers across the whole tile loop and is stored once at the end of the kernel.
// (Per-tile adds of every block -- float atomics into the workgroup's global slice, then ds_add_f32 into LDS accumulators -- cost
// 0.67 resp. 0.42 ms of a 2.0 ms call: an LDS float atomic retires about one lane per clock.)
template <int NW = 4>
__device__ __forceinline__ void accum_block(const float* __restrict__ scr0 /* wave 0's delta transpose */, int to, int ti, int lane,
                                            f32x16& acc, int wv0 = 0 /* first of the NW waves whose tiles are summed */) {
  const int i = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int wq = 0; wq < NW; ++wq) {
    const int wv = wv0 + wq;
    const float* ld_ = scr0 + wv * 2 * kTileLds + (32 * to + i) * 33 + kh;
    const float* lh_ = scr0 + wv * 2 * kTileLds + kTileLds + (32 * ti + i) * 33 + kh;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = tf_mfma(ld_[2 * s], lh_[2 * s], acc);
  }
}

// sum over the tile's 32 rows of an LDS [unit][row] tile: lanes 0..31 take unit 32*t + lane
__device__ __forceinline__ float row_sum(const float* __restrict__ l, int unit) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) s += l[unit * 33 + r];
  return s;
}

// backward through one net given delta4 (accumulator layout, 21 valid units); returns g of the 8 sample inputs (in8 order)
typedef const __attribute__((address_space(1))) float* tf_gfloat_p;
__device__ __forceinline__ void net_bwd(const float* __restrict__ tfrag_, const f32x16 (&in1)[1], const f32x16 (&h1)[2],
                                        const f32x16 (&h2)[2], const f32x16 (&h3)[2], const float (&g_wv)[32],
                                        float* __restrict__ lds_d, float* __restrict__ lds_h, const float* __restrict__ scr0, int wave,
                                        f32x16 (&accW)[4] /* this wave's blocks: dW3 (wave>>1, wave&1) | dW2 (same) | dW4 (0, wave&1) and dW1 (wave&1, 0) over the tiles of waves 2*(wave>>1), +1 */,
                                        float (&accB)[5] /* lanes 0..31: db4[lane] | db3[lane], [32+lane] | db2[lane], [32+lane] */,
                                        float* __restrict__ gP /* [pn][64] of this net */, int pt /* of this lane's row */, int lane,
                                        float (&g_in8)[8]) {
  const int h = lane >> 5;
  // global_load, not flat_load (the address space is lost behind the kernel's LICM fence)
  tf_gfloat_p tfrag = (tf_gfloat_p)tfrag_;
  constexpr int G = 8;
  float pf2[G * 2], pf1[G];
  f32x16 d4[1], d3[2], d2[2], d1[2], d0[1];
#pragma unroll
  for (int j = 0; j < 16; ++j) { const int r0 = (j & 3) + 8 * (j >> 2); d4[0][j] = h ? g_wv[r0 + 4] : g_wv[r0]; }
  // ---- layer 4: dW4 = d4 * h3^T, db4, d3 = (W4^T d4) * lrelu'(h3)
  // (two workgroup barriers per layer: every wave has read the transposes of the previous layer before they are overwritten; every
  // wave's transposes are in LDS before any wave accumulates its block over all four tiles)
  tf_layer_pf_first<16, 2, G>(tfrag + kT4 + lane, pf2);
  __syncthreads();
  tile_to_lds<1>(d4, lds_d, lane);
  tile_to_lds<2>(h3, lds_h, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  accum_block<2>(scr0, 0, wave & 1, lane, accW[2], 2 * (wave >> 1));
  if (lane < 21) accB[0] += row_sum(lds_d, lane);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) d3[t][j] = 0.f;
  tf_layer_pf_rest<16, 2, 1, G>(tfrag + kT4 + lane, pf2, d4, d3);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) d3[t][j] *= dleaky(h3[t][j]);
  // ---- layer 3
  tf_layer_pf_first<32, 2, G>(tfrag + kT3 + lane, pf2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  tile_to_lds<2>(d3, lds_d, lane);
  tile_to_lds<2>(h2, lds_h, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  accum_block(scr0, wave >> 1, wave & 1, lane, accW[0]);
  if (lane < 32) { accB[1] += row_sum(lds_d, lane); accB[2] += row_sum(lds_d, 32 + lane); }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) d2[t][j] = 0.f;
  tf_layer_pf_rest<32, 2, 2, G>(tfrag + kT3 + lane, pf2, d3, d2);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) d2[t][j] *= dleaky(h2[t][j]);
  // ---- layer 2
  tf_layer_pf_first<32, 2, G>(tfrag + kT2 + lane, pf2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  tile_to_lds<2>(d2, lds_d, lane);
  tile_to_lds<2>(h1, lds_h, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  accum_block(scr0, wave >> 1, wave & 1, lane, accW[1]);
  if (lane < 32) { accB[3] += row_sum(lds_d, lane); accB[4] += row_sum(lds_d, 32 + lane); }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) d1[t][j] = 0.f;
  tf_layer_pf_rest<32, 2, 2, G>(tfrag + kT2 + lane, pf2, d2, d1);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) d1[t][j] *= dleaky(h1[t][j]);
  // ---- layer 1 (sample part 64 x 8) + hoisted per-point part
  tf_layer_pf_first<32, 1, G>(tfrag + kT1 + lane, pf1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  tile_to_lds<2>(d1, lds_d, lane);
  tile_to_lds<1>(in1, lds_h, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  accum_block<2>(scr0, wave & 1, 0, lane, accW[3], 2 * (wave >> 1));
  // per-point sums of delta1 over the tile's rows, lanes 0..31 taking units lane and 32 + lane: the rows of a point are contiguous in
  // every caller's order (sn rows each, or what a mask keeps of them), so a tile holds one to a few runs; a run that ends inside the
  // tile is added at once, the last one after the product below.  Any order of rays_id is summed correctly (a run per change).
  // (Round 3 added every lane's 32 values with float atomics whenever a tile held two points: the rays_id form -- the specular
  // lobe's -- ran 1.16 ms for 158 k rows beside 1.01 ms for 262 k rows of the dense form.)
  float gp0 = 0.f, gp1 = 0.f;
  int gp_pt = __shfl(pt, 0);
  if (__all(pt == gp_pt)) {                          // one point (the dense form's every tile when sn is a multiple of 32)
    if (lane < 32) { gp0 = row_sum(lds_d, lane); gp1 = row_sum(lds_d, 32 + lane); }
  } else {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const int p_r = __shfl(pt, r);                // constant lane index: a scalar read
      if (p_r != gp_pt) {
        if (lane < 32) { atomicAdd(gP + (size_t)gp_pt * 64 + lane, gp0); atomicAdd(gP + (size_t)gp_pt * 64 + 32 + lane, gp1); }
        gp_pt = p_r;
        gp0 = gp1 = 0.f;
      }
      if (lane < 32) { gp0 += lds_d[lane * 33 + r]; gp1 += lds_d[(32 + lane) * 33 + r]; }
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) d0[0][j] = 0.f;
  tf_layer_pf_rest<32, 1, 2, G>(tfrag + kT1 + lane, pf1, d1, d0);
  // the per-point sums go out LAST: the fragment fetches of the next product would wait for these atomics (in-order memory counter)
  if (lane < 32) { atomicAdd(gP + (size_t)gp_pt * 64 + lane, gp0); atomicAdd(gP + (size_t)gp_pt * 64 + 32 + lane, gp1); }
  // d0 rows 0..7 = gradient of the 8 sample inputs; row rho(j,h): collect both halves
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float mine = d0[0][j], other = __shfl_xor(mine, 32);
    g_in8[j] = h ? other : mine;
    g_in8[4 + j] = h ? mine : other;
  }
}

__device__ __forceinline__ void embed8(float y, float (&in8)[8]) {
  in8[0] = y * 2.f - 1.f;
  in8[1] = sinf(y) * 2.f - 1.f; in8[2] = cosf(y) * 2.f - 1.f;
  in8[3] = sinf(y * 2.f) * 2.f - 1.f; in8[4] = cosf(y * 2.f) * 2.f - 1.f;
  in8[5] = sinf(y * 4.f) * 2.f - 1.f; in8[6] = cosf(y * 4.f) * 2.f - 1.f;
  in8[7] = 0.f;
}

__global__ void __launch_bounds__(256) flow_logq_bwd_kernel(const float* __restrict__ ws_arg, const float* __restrict__ P,
                                                            const float* __restrict__ xin, const long long* __restrict__ rays_id,
                                                            long long m, int sn, long long pn, const float* __restrict__ g_logq,
                                                            FlowGrads G, float* __restrict__ g_x, const float* __restrict__ z_saved) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // LDS image: the forward fragments and biases of both nets.
  for (int i = threadIdx.x; i < kFwdWs; i += 256) lds[i] = ws_arg[i];
  __syncthreads();
  // this wave's weight-gradient blocks of the two nets (net_bwd): in registers for the whole kernel
  f32x16 accW[2][4];
  float accB[2][5];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) accW[a][b][j] = 0.f;
#pragma unroll
    for (int b = 0; b < 5; ++b) accB[a][b] = 0.f;
  }
  const float* scr0 = lds + kFwdWs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* lds_d = lds + kFwdWs + wave * 2 * kTileLds;
  float* lds_h = lds_d + kTileLds;
  const long long n_tiles = (m + 31) / 32;
  // the four waves of the workgroup walk the tiles in lock step (net_bwd holds workgroup barriers): a wave past the last tile runs an
  // all-invalid tile (g = 0: it adds exact zeros)
  for (long long tile0 = (long long)blockIdx.x * 4; tile0 < n_tiles; tile0 += (long long)gridDim.x * 4) {
    const long long tile = tile0 + wave;
    const float* ws = ws_arg;
    asm volatile("" : "+s"(ws));                 // keep the transposed-fragment addresses out of LICM's reach
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const float* net_lds = lds + opaque;         // ... and the LDS weight operands
    long long row = tile * 32 + (lane & 31);
    const bool valid = row < m;
    if (!valid) row = m - 1;
    const long long pt = rays_id ? rays_id[row] : row / sn;
    const float x0 = fminf(fmaxf(xin[2 * row], 1e-6f), 1.f - 1e-6f);
    const float x1 = fminf(fmaxf(xin[2 * row + 1], 1e-6f), 1.f - 1e-6f);
    const float g = valid ? g_logq[row] : 0.f;
    // ---- forward recompute.  Net 1's activations are NOT kept across net 0's forward and backward (kept, the kernel needed 512
    // registers and still spilled 517 values into its hot loops): net 1 runs once for z0 and a second time just before its own
    // backward -- one more small forward per tile instead of the scratch traffic.
    float in8a[8], in8b[8], wv0[32];
    float z0, lj1;
    if (z_saved) {
      z0 = z_saved[2 * row];                       // the forward's own value: one evaluation of net 1 less per tile
    } else {
      float wv1[32];
      f32x16 in1a[1], a1[2], a2[2], a3[2];
      embed8(x1, in8a);
      net_fwd_keep(net_lds + hNetFloats, P + (pn + pt) * 64, in8a, lane, in1a, a1, a2, a3, wv1);   // net 1 keeps x1, moves x0
      pw_forward_fb(x0, wv1, z0, lj1);
    }
    asm volatile("" : "+v"(z0));                   // the first evaluation ends here: nothing of it but z0 stays live
    f32x16 in1b[1], b1[2], b2[2], b3[2];
    embed8(z0, in8b);
    net_fwd_keep(net_lds, P + pt * 64, in8b, lane, in1b, b1, b2, b3, wv0);                    // net 0 keeps z0, moves x1
    float z1, lj0;
    pw_forward_fb(x1, wv0, z1, lj0);
    // ---- reverse
    const float g_z1 = g * (-tanf(z1 * kHalfPi) * kHalfPi);
    float g_wv0[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) g_wv0[k] = 0.f;
    float g_x1 = pw_forward_bwd(x1, wv0, g_z1, g, g_wv0);        // x1 as the coordinate block 0 moves
    float g_in8[8];
    net_bwd(ws + kFwdWs, in1b, b1, b2, b3, g_wv0, lds_d, lds_h, scr0, wave, accW[0], accB[0], G.gP, (int)pt, lane, g_in8);
    // d(2*emb(z0) - 1)/dz0
    float g_z0 = 2.f * (g_in8[0] + g_in8[1] * cosf(z0) - g_in8[2] * sinf(z0) + 2.f * g_in8[3] * cosf(2.f * z0) -
                        2.f * g_in8[4] * sinf(2.f * z0) + 4.f * g_in8[5] * cosf(4.f * z0) - 4.f * g_in8[6] * sinf(4.f * z0));
    asm volatile("" : "+v"(g_z0));
    // net 1 again, this time keeping what its backward needs
    float wv1[32];
    f32x16 in1a[1], a1[2], a2[2], a3[2];
    float x1_again = x1;
    asm volatile("" : "+v"(x1_again));             // a value of its own: otherwise the two evaluations are merged and kept live after all
    embed8(x1_again, in8a);
    net_fwd_keep(net_lds + hNetFloats, P + (pn + pt) * 64, in8a, lane, in1a, a1, a2, a3, wv1);
    float g_wv1[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) g_wv1[k] = 0.f;
    const float g_x0 = pw_forward_bwd(x0, wv1, g_z0, g, g_wv1);
    net_bwd(ws + kFwdWs + kTNet, in1a, a1, a2, a3, g_wv1, lds_d, lds_h, scr0, wave, accW[1], accB[1], G.gP + pn * 64, (int)pt,
            lane, g_in8);
    if (g_x) {
      // ... and x1 as the coordinate block 1 keeps: through its embedding into net 1 (d(2 emb(x1) - 1) / d x1, as for z0 above)
      g_x1 += 2.f * (g_in8[0] + g_in8[1] * cosf(x1) - g_in8[2] * sinf(x1) + 2.f * g_in8[3] * cosf(2.f * x1) -
                     2.f * g_in8[4] * sinf(2.f * x1) + 4.f * g_in8[5] * cosf(4.f * x1) - 4.f * g_in8[6] * sinf(4.f * x1));
      if (valid && lane < 32) { g_x[2 * row] = g_x0; g_x[2 * row + 1] = g_x1; }
    }
  }
  // every wave stores the blocks it owns into the workgroup's (zero-filled) slice: plain stores where a block has one owner, adds
  // for the dW4 / dW1 blocks two waves hold half the rows of and for the bias sums every wave holds a quarter of
  float* const sl = G.slices + (size_t)blockIdx.x * kGradFloats;
  auto store_block = [&](float* dst, int ld, int nout, int nin, int to, int ti, const f32x16& a, bool shared) {
    const int col = 32 * ti + (lane & 31);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = 32 * to + tf_rho(j, lane >> 5);
      if (row < nout && col < nin) {
        if (shared) atomicAdd(dst + row * ld + col, a[j]);
        else dst[row * ld + col] = a[j];
      }
    }
  };
#pragma unroll
  for (int net = 0; net < 2; ++net) {
    float* dst = sl + net * kGNet;
    store_block(dst + kGW2, 64, 64, 64, wave >> 1, wave & 1, accW[net][0], false);
    store_block(dst + kGW1, 64, 64, 64, wave >> 1, wave & 1, accW[net][1], false);
    store_block(dst + kGW3, 64, 21, 64, 0, wave & 1, accW[net][2], true);
    store_block(dst + kGW0, 44, 64, 7, wave & 1, 0, accW[net][3], true);
    if (lane < 32) {
      if (lane < 21) atomicAdd(dst + kGB3 + lane, accB[net][0]);
      atomicAdd(dst + kGB2 + lane, accB[net][1]);
      atomicAdd(dst + kGB2 + 32 + lane, accB[net][2]);
      atomicAdd(dst + kGB1 + lane, accB[net][3]);
      atomicAdd(dst + kGB1 + 32 + lane, accB[net][4]);
    }
  }
}

// fold the per-workgroup slices into the caller's gradient tensors (+=): one thread per gradient element
// (eight groups of slices per element, joined by an atomic: one thread per element walking all 256 slices was 100 workgroups of
// dependent-latency loads, 62 us for 26 MB)
__global__ void __launch_bounds__(256) flow_grad_fold_kernel(FlowGrads G, int n_blocks) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= kGradFloats) return;
  const int per = (n_blocks + (int)gridDim.y - 1) / (int)gridDim.y, b0 = blockIdx.y * per, b1 = min(b0 + per, n_blocks);
  if (b0 >= b1) return;
  float s = 0.f;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) s += G.slices[(size_t)b * kGradFloats + e];
  const int net = e / kGNet, r = e % kGNet;
  float* dst;
  if (r < kGW1) dst = G.w[net][0] + r;
  else if (r < kGW2) dst = G.w[net][1] + (r - kGW1);
  else if (r < kGW3) dst = G.w[net][2] + (r - kGW2);
  else if (r < kGB1) dst = G.w[net][3] + (r - kGW3);
  else if (r < kGB2) dst = G.b[net][1] + (r - kGB1);
  else if (r < kGB3) dst = G.b[net][2] + (r - kGB2);
  else if (r < kGB3 + 21) dst = G.b[net][3] + (r - kGB3);
  else return;
  atomicAdd(dst, s);
}

// The hoisted layer-1 point part, folded back (tf_flow_logq_bwd with g_cond): c = 2 cond - 1 [pn,37], P_k = c W1_k[:, 7:44]^T + b1_k
//   g_cond[pt][j]       = 2 sum_k sum_u gP[k][pt][u] W1_k[u][7 + j]
//   gW1_k[u][7 + j]    += sum_pt gP[k][pt][u] c[pt][j],      gb1_k[u] += sum_pt gP[k][pt][u]
// (round 5: the caller's three small dense-layer products per net were ~20 launches per call)
__global__ void __launch_bounds__(256) flow_point_fold_cond_kernel(const float* __restrict__ gP, const float* __restrict__ w1a,
                                                                   const float* __restrict__ w1b, long long pn, float* __restrict__ g_cond) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= pn * 37) return;
  const long long pt = e / 37;
  const int j = (int)(e - pt * 37);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float* w = (k ? w1b : w1a) + 7 + j;
    const float* g = gP + ((long long)k * pn + pt) * 64;
#pragma unroll 8
    for (int u = 0; u < 64; ++u) acc += g[u] * w[u * 44];
  }
  g_cond[e] = 2.f * acc;
}
__global__ void __launch_bounds__(256) flow_point_fold_w_kernel(const float* __restrict__ gP, const float* __restrict__ cond, long long pn,
                                                                float* __restrict__ gw1a, float* __restrict__ gb1a,
                                                                float* __restrict__ gw1b, float* __restrict__ gb1b) {
  const int e = blockIdx.y * 256 + threadIdx.x;            // (k, u, j'), j' fastest; j' == 37: the bias column
  if (e >= 2 * 64 * 38) return;
  const int j = e % 38, u = (e / 38) % 64, k = e / (38 * 64);
  const long long p0 = (long long)blockIdx.x * 32, p1 = p0 + 32 < pn ? p0 + 32 : pn;   // 19 x pn / 32 workgroups: the sum over points is the long axis
  const float* g = gP + (long long)k * pn * 64 + u;
  float acc = 0.f;
  if (j < 37) {
    for (long long pt = p0; pt < p1; ++pt) acc += g[pt * 64] * (2.f * cond[pt * 37 + j] - 1.f);
    atomicAdd((k ? gw1b : gw1a) + u * 44 + 7 + j, acc);
  } else {
    for (long long pt = p0; pt < p1; ++pt) acc += g[pt * 64];
    atomicAdd((k ? gb1b : gb1a) + u, acc);
  }
}

// shared with flow.hip
__global__ void __launch_bounds__(256) flow_point_part_kernel2(const float* __restrict__ w1a, const float* __restrict__ b1a,
                                                               const float* __restrict__ w1b, const float* __restrict__ b1b,
                                                               const float* __restrict__ cond, long long pn, float* __restrict__ P) {
  long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= 2 * pn * 64) return;
  int u = (int)(e & 63);
  long long pt = (e >> 6) % pn;
  int blk = (int)((e >> 6) / pn);
  const float* w = (blk ? w1b : w1a) + u * 44 + 7;
  float acc = (blk ? b1b : b1a)[u];
  const float* c = cond + pt * 37;
#pragma unroll
  for (int k = 0; k < 37; ++k) acc += w[k] * (c[k] * 2.f - 1.f);
  P[e] = acc;
}

extern "C" int tf_flow_logq_bwd(const TfCouplingNet nets[2], const float* cond, const float* x, const float* z_saved,
                                const int64_t* rays_id, int64_t m, int32_t sn, int64_t pn, const float* g_logq,
                                const TfCouplingNetGrad gnets[2], float* g_point, float* g_cond, float* g_x, float* workspace,
                                size_t workspace_floats, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "tf_flow_logq_bwd";
  TF_REQUIRE(m >= 0 && pn >= 0 && sn > 0, TF_ESHAPE, "%s: negative size / sn <= 0", who);
  if (m == 0) return TF_OK;
  TF_REQUIRE(nets && gnets && cond && x && g_logq && g_point && workspace, TF_EINVAL, "%s: null pointer", who);
  TF_REQUIRE(pn > 0 && (rays_id || m == pn * (int64_t)sn), TF_ESHAPE, "%s: without rays_id m must equal pn*sn", who);
  TF_REQUIRE(pn < (1LL << 31), TF_ESHAPE, "%s: pn must be < 2^31", who);
  TF_REQUIRE(workspace_floats >= tf_flow_bwd_workspace_floats(pn), TF_ESHAPE, "%s: workspace too small", who);
  FlowGrads G;
  for (int b = 0; b < 2; ++b)
    for (int l = 0; l < 4; ++l) {
      TF_REQUIRE(nets[b].w[l] && nets[b].b[l] && gnets[b].w[l] && gnets[b].b[l], TF_EINVAL, "%s: null weight / gradient pointer", who);
      G.w[b][l] = gnets[b].w[l];
      G.b[b][l] = gnets[b].b[l];
    }
  G.gP = g_point;
  TfPackBatch PB(stream);           // fragments and transposed fragments of both coupling nets: ONE launch (was 22)
  for (int b = 0; b < 2; ++b) {
    float* base = workspace + (size_t)b * hNetFloats;
    _Float16* hb = reinterpret_cast<_Float16*>(base);
    PB.wfrag_h3(nets[b].w[0], 64, 44, 0, 7, 2, 1, hb + 2 * (size_t)hL1);
    PB.wfrag_h3(nets[b].w[1], 64, 64, 0, 64, 2, 4, hb + 2 * (size_t)hL2);
    PB.wfrag_h3(nets[b].w[2], 64, 64, 0, 64, 2, 4, hb + 2 * (size_t)hL3);
    PB.wfrag_h3(nets[b].w[3], 21, 64, 0, 64, 1, 4, hb + 2 * (size_t)hL4);
    PB.bias(nets[b].b[1], 64, 2, base + hB2);
    PB.bias(nets[b].b[2], 64, 2, base + hB3);
    PB.bias(nets[b].b[3], 21, 1, base + hB4);
    float* tb = workspace + kFwdWs + (size_t)b * kTNet;
    // transposed fragments: logical matrix = W^T  (rows = layer inputs, cols = layer outputs)
    PB.wfrag(nets[b].w[3], 64, 64, 0, 21, 2, 16, tb + kT4, 0, 1);
    PB.wfrag(nets[b].w[2], 64, 64, 0, 64, 2, 32, tb + kT3, 0, 1);
    PB.wfrag(nets[b].w[1], 64, 64, 0, 64, 2, 32, tb + kT2, 0, 1);
    PB.wfrag(nets[b].w[0], 7, 44, 0, 64, 1, 32, tb + kT1, 0, 1);
  }
  PB.flush();
  float* P = workspace + kBwdWs;
  flow_point_part_kernel2<<<tf_blocks(2 * pn * 64, 256), 256, 0, stream>>>(nets[0].w[0], nets[0].b[0], nets[1].w[0], nets[1].b[0],
                                                                          cond, pn, P);
  const size_t lds = (size_t)(kFwdWs + 4 * 2 * kTileLds) * sizeof(float);
  static std::atomic<unsigned long long> attr_set{0};
  int attr_dev;
  if (tf_once_needed(attr_set, &attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)flow_logq_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    TF_REQUIRE(e == hipSuccess, TF_EHIP, "%s: hipFuncSetAttribute failed: %s", who, hipGetErrorString(e));
    tf_once_done(attr_set, attr_dev);
  }
  const long long tiles = (m + 31) / 32;
  long long blocks = (tiles + 3) / 4;
  if (blocks > kBwdBlocks) blocks = kBwdBlocks;
  G.slices = workspace + kBwdWs + (size_t)2 * 64 * (size_t)pn;
  hipError_t e2 = hipMemsetAsync(G.slices, 0, (size_t)blocks * kGradFloats * sizeof(float), stream);
  TF_REQUIRE(e2 == hipSuccess, TF_EHIP, "%s: hipMemsetAsync failed: %s", who, hipGetErrorString(e2));
  flow_logq_bwd_kernel<<<(unsigned)blocks, 256, lds, stream>>>(workspace, P, x, (const long long*)rays_id, m, sn, pn, g_logq, G, g_x, z_saved);
  flow_grad_fold_kernel<<<dim3(tf_blocks(kGradFloats, 256), 8), 256, 0, stream>>>(G, (int)blocks);
  if (g_cond) {
    flow_point_fold_cond_kernel<<<tf_blocks(pn * 37, 256), 256, 0, stream>>>(g_point, nets[0].w[0], nets[1].w[0], pn, g_cond);
    flow_point_fold_w_kernel<<<dim3((unsigned)((pn + 31) / 32), (2 * 64 * 38 + 255) / 256), 256, 0, stream>>>(
        g_point, cond, pn, gnets[0].w[0], gnets[0].b[0], gnets[1].w[0], gnets[1].b[0]);
  }
  TF_LAUNCH_CHECK(who);
  return TF_OK;
}
