// TensoFlow coupling flow ('pwquad'): fused sampler and density kernels.
// Replaces TensoFlow.flow / flow_inv + Block.flow / flow_inv + ElementWisePWQuadraticTransform
// (network/flow.py:314-525, :549-641, :766-799, :801-855) -- ~30 tiny PyTorch kernels per block and
// a [pn*sn, 44] activation round trip per layer in the reference -- by ONE launch:
//   * rows (point, sample) sit on MFMA columns; the two 44-64-64-64-21 nets run as fp32 MFMA
//     (mfma_mlp.h) with all fragment-ordered weights resident in LDS (86 KB per workgroup);
//   * the 37 per-point condition inputs of layer 1 are hoisted into a per-point vector
//     P[pn,64] (one small pre-pass), so layer 1 costs 8 instead of 44 k-steps per sample;
//   * the spline (normalisation, bin search, quadratic solve / evaluation, log-det) is evaluated
//     in registers in the order of operations of the reference so bin indices agree.
// Bound: fp32 MFMA (157 TF/s); HBM traffic is 16-20 B per sample.
#include "mfma_mlp.h"
#include "flow_image.h"
#include "tf_common.h"
#include "tf_internal.h"

#define FLOW_NB 10
static constexpr float kEps32 = 1.1920928955078125e-07f;  // torch.finfo(float32).eps
static constexpr float kHalfPi = 1.5707963267948966f;

// ------------------------------------------------------------------------------- spline
// The reference evaluates the spline with separate fp32 multiplies and adds (PyTorch eager); its closed-form
// root (-b +- sqrt(b^2 - 2ac)) / a is ill-conditioned for nearly flat bins, so fused multiply-adds here would
// change the rounding sequence and move samples by up to ~1e-4.  Contraction is therefore off for the spline.
#pragma clang fp contract(off)
struct PwTables {
  float w[FLOW_NB], wss[FLOW_NB + 1], v[FLOW_NB + 1], vw[FLOW_NB + 1];
};

// x / d for many x and one d: reciprocal refined by one Newton step, quotient corrected by its fma residual (Markstein):
// 3 instructions per quotient instead of the ~10 of the IEEE sequence (v_div_scale / v_div_fmas / v_div_fixup); the result
// is the correctly rounded quotient except for rare 1-ulp cases -- far below the 1e-6 the MFMA summation order already
// moves the spline parameters by.  Divisors here are sums of exponentials (positive, normal range).
struct SharedDiv {
  float d, r;
  __device__ __forceinline__ explicit SharedDiv(float d_) : d(d_) {
    const float r0 = __builtin_amdgcn_rcpf(d_);
    r = fmaf(fmaf(-d_, r0, 1.f), r0, r0);
  }
  __device__ __forceinline__ float operator()(float x) const {
    const float q = x * r;
    return fmaf(fmaf(-d, q, x), r, q);
  }
};

// exp / log on the transcendental unit: v_exp_f32(x * log2 e) and v_log_f32(x) * ln 2 (2 instructions each) instead of the
// ~10 / ~15 of the libm sequences (range scaling for denormal results, extra-precision argument) -- 42 exponentials per row made
// them 8 % of the kernel's vector instructions.  Relative error <= |x| * 6e-8 + 1 ulp; arguments here are net outputs of
// magnitude O(1) and knot heights >= 1e-6 (normal range), and the spline already moves by ~1e-6 with the MFMA summation order.
#ifdef TF_FLOW_LIBM   // dev-only switch
__device__ __forceinline__ float tf_exp(float x) { return expf(x); }
__device__ __forceinline__ float tf_log(float x) { return logf(x); }
#else
__device__ __forceinline__ float tf_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float tf_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
#endif

template <bool CLAMP_W>
__device__ __forceinline__ void pw_tables(const float (&wv)[32], PwTables& T) {
  // wv[0..10] = v_tilde, wv[11..20] = w_tilde   (flow.py:337-350 / :420-434)
  float wsum[FLOW_NB];
  float run = 0.f;
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) {
    float e = tf_exp(wv[11 + i]);
    if (CLAMP_W) e = fmaxf(e, 1e-6f);
    T.w[i] = e;
    run += e;
    wsum[i] = run;
  }
  const SharedDiv by_wn(run);
  T.wss[0] = 0.f;
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) {
    float wi = by_wn(T.w[i]);
    if (CLAMP_W) wi = fmaxf(wi, 1e-6f);
    T.w[i] = wi;
    T.wss[i + 1] = by_wn(wsum[i]);
  }
  float ev[FLOW_NB + 1];
#pragma unroll
  for (int i = 0; i <= FLOW_NB; ++i) ev[i] = tf_exp(wv[i]);
  float den = 0.f;
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) den += (ev[i] + ev[i + 1]) / 2.f * T.w[i];
  const SharedDiv by_den(den);
#pragma unroll
  for (int i = 0; i <= FLOW_NB; ++i) T.v[i] = fmaxf(by_den(ev[i]), 1e-6f);
  T.vw[0] = 0.f;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < FLOW_NB; ++i) {
    acc += (T.v[i] + T.v[i + 1]) / 2.f * T.w[i];
    T.vw[i + 1] = acc;
  }
}

#define PW_PICK(arr, n, idx, dst)            \
  {                                          \
    dst = arr[0];                            \
    _Pragma("unroll") for (int q_ = 1; q_ < (n); ++q_) dst = (idx == q_) ? arr[q_] : dst; \
  }

// sampling direction (inverse spline), flow.py:415-525
__device__ __forceinline__ void pw_inverse(float y, const float (&wv)[32], float& x, float& logj, int& bin) {
  PwTables T;
  pw_tables<false>(wv, T);
  // arg-max trick of flow.py:443-457, emulated exactly (first occurrence of the maximum of
  // [eps, finder*(vw+1)]), so ties from rounding pick the same bin as the reference.
#ifdef TF_FLOW_PICK_SEPARATE   // dev-only switch: rounds 1-5's form (search, then five select chains over the tables: 152 instructions)
  int cnt = 0;
  {
    float best = kEps32;
#pragma unroll
    for (int i = 0; i <= FLOW_NB; ++i) {
      float val = (T.vw[i] > y) ? 0.f : T.vw[i] + 1.f;
      if (val > best) { best = val; cnt = i + 1; }
    }
  }
  int e = min(max(cnt - 1, 0), FLOW_NB - 1);
  float ve, ve1, we, vwe, wsse;
  PW_PICK(T.v, FLOW_NB + 1, e, ve)
  {
    int e1 = e + 1;
    PW_PICK(T.v, FLOW_NB + 1, e1, ve1)
  }
  PW_PICK(T.w, FLOW_NB, e, we)
  PW_PICK(T.vw, FLOW_NB + 1, e, vwe)
  PW_PICK(T.wss, FLOW_NB + 1, e, wsse)
#else
  // The same search with the bin's five table entries picked up ON THE WAY (round 6): whenever the running maximum moves to knot i the
  // entries of bin e = min(i, NB - 1) are taken along -- one compare + six selects per knot instead of a search and five select chains
  // over the finished index (110 against 152 vector instructions per spline; no arithmetic changes, bins and values bit-identical).
  int e = 0;
  float ve = T.v[0], ve1 = T.v[1], we = T.w[0], vwe = T.vw[0], wsse = T.wss[0];
  {
    float best = kEps32;
#pragma unroll
    for (int i = 0; i <= FLOW_NB; ++i) {
      const float val = (T.vw[i] > y) ? 0.f : T.vw[i] + 1.f;
      const bool up = val > best;
      constexpr int NBm1 = FLOW_NB - 1;
      const int b_ = i < NBm1 ? i : NBm1;       // cnt = i + 1 -> e = clamp(cnt - 1, 0, NB - 1)
      best = up ? val : best;
      e = up ? b_ : e;
      ve = up ? T.v[b_] : ve; ve1 = up ? T.v[b_ + 1] : ve1; we = up ? T.w[b_] : we; vwe = up ? T.vw[b_] : vwe; wsse = up ? T.wss[b_] : wsse;
    }
  }
#endif
  float a = (ve1 - ve) * we;
  float b = ve * we;
  float c = vwe - y;
  a = fabsf(a) < kEps32 ? kEps32 : a;
  float d = fmaxf(b * b - 2.f * a * c, 0.f);
  float sq = sqrtf(d);
  const SharedDiv by_a(a);
  float s1 = by_a(-b - sq), s2 = by_a(-b + sq);
  float sol = (s1 >= 0.f && s1 < 1.f) ? s1 : s2;
  sol = fminf(fmaxf(sol, kEps32), 1.f - kEps32);
  x = fminf(fmaxf(we * sol + wsse, kEps32), 1.f - kEps32);
  // torch.lerp(start, end, w) = start + w*(end-start) for w < 0.5 and end - (end-start)*(1-w) otherwise
  const float lerp = sol < 0.5f ? ve + sol * (ve1 - ve) : ve1 - (ve1 - ve) * (1.f - sol);
  logj = -tf_log(lerp);
  bin = e;
}

// density direction (forward spline), flow.py:332-413
__device__ __forceinline__ void pw_forward(float xin, const float (&wv)[32], float& out, float& logj, int& bin) {
  PwTables T;
  pw_tables<true>(wv, T);
#ifdef TF_FLOW_PICK_SEPARATE
  int cnt = 0;   // flow.py:355-366: argmax of [eps, finder*wsum], first occurrence
  {
    float best = kEps32;
#pragma unroll
    for (int i = 1; i <= FLOW_NB; ++i) {
      float val = (T.wss[i] > xin) ? 0.f : T.wss[i];
      if (val > best) { best = val; cnt = i; }
    }
  }
  int m = min(max(cnt, 0), FLOW_NB - 1);
  float vm, vm1, wm, vwm, wssm;
  PW_PICK(T.v, FLOW_NB + 1, m, vm)
  {
    int m1 = m + 1;
    PW_PICK(T.v, FLOW_NB + 1, m1, vm1)
  }
  PW_PICK(T.w, FLOW_NB, m, wm)
  PW_PICK(T.vw, FLOW_NB + 1, m, vwm)
  PW_PICK(T.wss, FLOW_NB + 1, m, wssm)
#else
  // flow.py:355-366: argmax of [eps, finder*wsum], first occurrence -- with the bin's table entries picked up on the way (see pw_inverse)
  int m = 0;
  float vm = T.v[0], vm1 = T.v[1], wm = T.w[0], vwm = T.vw[0], wssm = T.wss[0];
  {
    float best = kEps32;
#pragma unroll
    for (int i = 1; i <= FLOW_NB; ++i) {
      const float val = (T.wss[i] > xin) ? 0.f : T.wss[i];
      const bool up = val > best;
      constexpr int NBm1 = FLOW_NB - 1;
      const int b_ = i < NBm1 ? i : NBm1;       // cnt = i -> m = clamp(cnt, 0, NB - 1)
      best = up ? val : best;
      m = up ? b_ : m;
      vm = up ? T.v[b_] : vm; vm1 = up ? T.v[b_ + 1] : vm1; wm = up ? T.w[b_] : wm; vwm = up ? T.vw[b_] : vwm; wssm = up ? T.wss[b_] : wssm;
    }
  }
#endif
  float al = fminf(fmaxf((xin - wssm) / wm, 0.f), 1.f);
  float o = (al * al) / 2.f * ((vm1 - vm) * wm) + al * vm * wm + vwm;
  out = fminf(fmaxf(o, kEps32), 1.f - kEps32);
  float lerp = al < 0.5f ? vm + al * (vm1 - vm) : vm1 - (vm1 - vm) * (1.f - al);
  logj = tf_log(lerp);
  bin = m;
}

// ------------------------------------------------------------------------------- kernel
// LDS images per block net: flow_image.h

// LeakyReLU(0.01) as max(x, 0.01 x): two instructions instead of compare + multiply + select, same value for every x
// One v_mul + one v_med3: median(x, 0.01 x, +inf) = max(x, 0.01 x).  Written as fmaxf() the compiler first canonicalises the
// accumulator value (v_max x, x: the matrix-core result is not a known-canonical float to it) -- three instructions per
// activation, 1 150 per 64 rows.  (A packed v_pk_mul_f32 for the products was tried and is NOT safe here: results changed
// from run to run on 9 % of the samples -- a packed-FP32 read of a matrix-core result that hipcc's hazard recogniser does not
// cover; the check was tools/exp_flow_determinism.py of the round-5 tree, git e89d4f5: the same launch twice, bit-compared.)
__device__ __forceinline__ float leaky(float x) {
#ifdef TF_FLOW_FMAX_LEAKY   // dev-only switch: the previous form
  return fmaxf(x, 0.01f * x);
#else
  float big = 3.0e38f;             // opaque to the optimiser (med3 with a literal +inf is folded back into the canonicalising max)
  asm("" : "+s"(big));
  return __builtin_amdgcn_fmed3f(x, 0.01f * x, big);
#endif
}
__device__ __forceinline__ void leaky16(f32x16& v) {
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = leaky(v[j]);
}

// net eval for one 32-row tile: y_keep / P row of the row on this lane's MFMA column (lane & 31) -> o: the 21 (+pad)
// outputs of the tile in accumulator layout (lane half h holds units rho(j, h) of its column's row)
// MODE: 0 = exact fp32 MFMA, 3 = f16x3, 1 = plain f16 operands (TF_PREC_F16; same fragment image as f16x3, lo halves unused)
template <int MODE>
__device__ __forceinline__ void coupling_net(const float* __restrict__ net /*LDS*/, const float* __restrict__ Prow,
                                             float y_keep, int lane, f32x16 (&o)[1]) {
  const int h = lane >> 5;
  constexpr bool H3 = MODE != 0;
  constexpr int TERMS = MODE == 1 ? 1 : 3;
  constexpr int B2 = H3 ? hB2 : kB2, B3 = H3 ? hB3 : kB3, B4 = H3 ? hB4 : kB4;
  const tf_h8* nh = reinterpret_cast<const tf_h8*>(net) + lane;
  // layer-1 sample part: embed3(y) (7 values, Reshift 2x-1), k = rho(j,h), j = 0..3
  float emb[8];
  emb[0] = y_keep;
  tf_sincos_small(y_keep, emb[1], emb[2]);          // y_keep in (0,1): arguments <= 4 (libm sinf/cosf: ~100 instructions each)
  tf_sincos_small(y_keep * 2.f, emb[3], emb[4]);
  tf_sincos_small(y_keep * 4.f, emb[5], emb[6]);
  emb[7] = 0.5f;  // padded input: 2*0.5-1 = 0 (its weight column is zero anyway)
  f32x16 in1[1];
#pragma unroll
  for (int j = 0; j < 16; ++j) in1[0][j] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) in1[0][j] = (h ? emb[4 + j] : emb[j]) * 2.f - 1.f;
  // (f16x3: k-step 0 takes registers 0..7; 4..7 stay 0 -> inputs 8..15 are padding)
  f32x16 a[2], b[2];
  // init accumulators with the hoisted per-point part (already includes b1)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) a[t][j] = Prow[32 * t + tf_rho(j, h)];
  if (H3) tf_layer_h3<1, 2, 1, TERMS>(nh + hL1 / 4, in1, a);
  else tf_layer<4, 2, 1>(net + kL1 + lane, in1, a);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) b[t][j] = net[B2 + (t * 16 + j) * 2 + h];
#pragma unroll
  for (int t = 0; t < 2; ++t) leaky16(a[t]);
  if (H3) tf_layer_h3<4, 2, 2, TERMS>(nh + hL2 / 4, a, b);
  else tf_layer<32, 2, 2>(net + kL2 + lane, a, b);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) a[t][j] = net[B3 + (t * 16 + j) * 2 + h];
#pragma unroll
  for (int t = 0; t < 2; ++t) leaky16(b[t]);
  if (H3) tf_layer_h3<4, 2, 2, TERMS>(nh + hL3 / 4, b, a);
  else tf_layer<32, 2, 2>(net + kL3 + lane, b, a);
#pragma unroll
  for (int t = 0; t < 2; ++t) leaky16(a[t]);
#pragma unroll
  for (int j = 0; j < 16; ++j) o[0][j] = net[B4 + j * 2 + h];
  if (H3) tf_layer_h3<4, 1, 2, TERMS>(nh + hL4 / 4, a, o);
  else tf_layer<32, 1, 2>(net + kL4 + lane, a, o);
}

// ---- The two tiles of a 64-row group, software-pipelined by half a layer (f16 operand modes).
// Measured on the kernel above (coupling_net twice per block): one wave per SIMD 12.4 ms, two 10.8, three 10.4 for the bench's two
// launches -- a SIMD's time is the SUM of its matrix-core cycles (264 MFMAs x 32) and its vector-issue cycles (~2 750 x 4) whatever
// the number of waves, because a wave's own stream alternates a block of vector instructions (LeakyReLU + operand split of a whole
// layer) with a block of dependent MFMAs.  A vector instruction issued BETWEEN two MFMAs of the same wave costs nothing while at most
// ~6 of them follow each MFMA (tools/mfma_valu_overlap.hip), so the two tiles are interleaved in ONE instruction stream: while the
// matrix cores run layer k of tile A, the vector unit activates and splits tile B's layer k-1 output, pair by pair, one pair behind
// each MFMA; scheduling barriers pin that order.  The arithmetic per accumulator is unchanged (same MFMA sequence, same split), so
// results are bit-identical to coupling_net.
struct FlowS16 {          // pre-split operands of one tile's 32 layer inputs: pair p = (value 2p, value 2p + 1) as packed halves
  unsigned hi[16], lo[16];
};

template <int TERMS, bool LEAKY>
__device__ __forceinline__ void flow_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  if (LEAKY) { x0 = leaky(x0); x1 = leaky(x1); }
  // the consumer is an MFMA of a LATER stage (>= 6 MFMAs away): no hazard nop needed behind the asm statement
  if (TERMS == 3)
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hi), "=&v"(lo) : "v"(x0), "v"(x1));
  else
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
}

// One layer of one tile on the matrix cores; `piece(i)` is called behind MFMA i (0 .. K16 * TOUT * TERMS - 1): the other tile's vector work.
template <int K16, int TOUT, int TERMS, typename Piece>
__device__ __forceinline__ void flow_mfma_stage(const tf_h8* __restrict__ wf /* + lane */, const FlowS16& S, f32x16 (&out)[TOUT], Piece&& piece) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  tf_h8 fa[TOUT][2], fn[TOUT][2];
#pragma unroll
  for (int t = 0; t < TOUT; ++t) { fa[t][0] = wf[t * 128]; if (TERMS == 3) fa[t][1] = wf[t * 128 + 64]; }
  int slot = 0;
#pragma unroll
  for (int s16 = 0; s16 < K16; ++s16) {
    if (s16 + 1 < K16) {      // next k-step's weight fragments: requested one k-step ahead of their MFMAs
#pragma unroll
      for (int t = 0; t < TOUT; ++t) { fn[t][0] = wf[((s16 + 1) * TOUT + t) * 128]; if (TERMS == 3) fn[t][1] = wf[((s16 + 1) * TOUT + t) * 128 + 64]; }
    }
    u32x4 bh, bl;
#pragma unroll
    for (int q = 0; q < 4; ++q) { bh[q] = S.hi[4 * s16 + q]; bl[q] = S.lo[4 * s16 + q]; }
    const tf_h8 b_hi = __builtin_bit_cast(tf_h8, bh), b_lo = __builtin_bit_cast(tf_h8, bl);
#pragma unroll
    for (int t = 0; t < TOUT; ++t) {
      out[t] = tf_mfma_h(fa[t][0], b_hi, out[t]);
      piece(slot++); __builtin_amdgcn_sched_barrier(0);
      if (TERMS == 3) {
        out[t] = tf_mfma_h(fa[t][0], b_lo, out[t]);
        piece(slot++); __builtin_amdgcn_sched_barrier(0);
        out[t] = tf_mfma_h(fa[t][1], b_hi, out[t]);
        piece(slot++); __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (s16 + 1 < K16) {
#pragma unroll
      for (int t = 0; t < TOUT; ++t) { fa[t][0] = fn[t][0]; if (TERMS == 3) fa[t][1] = fn[t][1]; }
    }
  }
}

// pair p of a finished layer: activate, split into S, and re-initialise the two accumulator registers with the NEXT layer's bias
template <int TERMS>
__device__ __forceinline__ void flow_post_pair(int p, f32x16 (&acc)[2], FlowS16& S, const float* __restrict__ bias_next /* LDS, or null */, int h) {
  const int t = p >> 3, j = 2 * (p & 7);
  flow_pair<TERMS, true>(acc[t][j], acc[t][j + 1], S.hi[p], S.lo[p]);
  if (bias_next) { acc[t][j] = bias_next[(t * 16 + j) * 2 + h]; acc[t][j + 1] = bias_next[(t * 16 + j + 1) * 2 + h]; }
}

template <int MODE>
__device__ __forceinline__ void coupling_pair(const float* __restrict__ net /*LDS*/, const float* __restrict__ ProwA, const float* __restrict__ ProwB,
                                              float yA, float yB, int lane, f32x16 (&oA)[1], f32x16 (&oB)[1]) {
  static_assert(MODE != 0, "f16 operand modes only");
  constexpr int TERMS = MODE == 1 ? 1 : 3;
  const int h = lane >> 5;
  const tf_h8* nh = reinterpret_cast<const tf_h8*>(net) + lane;
  f32x16 aA[2], aB[2];
  FlowS16 SA, SB;
  auto embed = [&](float y, FlowS16& S) {     // layer-1 sample part: embed3(y) (7 values + pad, Reshift 2x-1), k = rho(j, h), j = 0..3
    float emb[8];
    emb[0] = y;
    tf_sincos_small(y, emb[1], emb[2]);
    tf_sincos_small(y * 2.f, emb[3], emb[4]);
    tf_sincos_small(y * 4.f, emb[5], emb[6]);
    emb[7] = 0.5f;
    float in4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) in4[j] = (h ? emb[4 + j] : emb[j]) * 2.f - 1.f;
    flow_pair<TERMS, false>(in4[0], in4[1], S.hi[0], S.lo[0]);
    flow_pair<TERMS, false>(in4[2], in4[3], S.hi[1], S.lo[1]);
    S.hi[2] = S.hi[3] = 0u; S.lo[2] = S.lo[3] = 0u;          // inputs 8..15 of the k-step are padding
  };
  auto init_p = [&](const float* Prow, f32x16 (&a)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) a[t][j] = Prow[32 * t + tf_rho(j, h)];
  };
  auto none = [](int) {};
  // (an asm-written operand that the very next MFMA reads needs its two wait states spelled out: the hazard recogniser does not see
  // vector writes made inside an asm string)
  // stage 0: tile A's layer-1 input
  init_p(ProwA, aA);
  embed(yA, SA);
  asm volatile("s_nop 1");
  // stage 1: L1(A) on the matrix cores | tile B's layer-1 input
  init_p(ProwB, aB);
  flow_mfma_stage<1, 2, TERMS>(nh + hL1 / 4, SA, aA, none);
  embed(yB, SB);
  asm volatile("s_nop 1");
  // stage 2: L1(B) | post(A, layer 1) -> SA, aA <- b2
  flow_mfma_stage<1, 2, TERMS>(nh + hL1 / 4, SB, aB, [&](int i) { flow_post_pair<TERMS>(i, aA, SA, net + hB2, h); });
#pragma unroll
  for (int p = 2 * TERMS; p < 16; ++p) flow_post_pair<TERMS>(p, aA, SA, net + hB2, h);
  asm volatile("s_nop 1");
  // pieces behind the MFMAs of a 4-k-step layer: TERMS == 3 -> 24 slots, pairs behind two of every three; TERMS == 1 -> 8 slots, two pairs each
  auto spread = [&](int i, f32x16 (&acc)[2], FlowS16& S, const float* bias) {
    if (TERMS == 3) { if (i % 3 != 2) flow_post_pair<TERMS>(2 * (i / 3) + i % 3, acc, S, bias, h); }
    else { flow_post_pair<TERMS>(2 * i, acc, S, bias, h); flow_post_pair<TERMS>(2 * i + 1, acc, S, bias, h); }
  };
  // stage 3: L2(A) | post(B, layer 1) -> SB, aB <- b2
  flow_mfma_stage<4, 2, TERMS>(nh + hL2 / 4, SA, aA, [&](int i) { spread(i, aB, SB, net + hB2); });
  // stage 4: L2(B) | post(A, layer 2) -> SA, aA <- b3
  flow_mfma_stage<4, 2, TERMS>(nh + hL2 / 4, SB, aB, [&](int i) { spread(i, aA, SA, net + hB3); });
  // stage 5: L3(A) | post(B, layer 2) -> SB, aB <- b3
  flow_mfma_stage<4, 2, TERMS>(nh + hL3 / 4, SA, aA, [&](int i) { spread(i, aB, SB, net + hB3); });
  // stage 6: L3(B) | post(A, layer 3) -> SA ; oA <- b4
  flow_mfma_stage<4, 2, TERMS>(nh + hL3 / 4, SB, aB, [&](int i) { spread(i, aA, SA, nullptr); });
#pragma unroll
  for (int j = 0; j < 16; ++j) oA[0][j] = net[hB4 + j * 2 + h];
  // stage 7: L4(A) (4 k-steps x 1 tile) | post(B, layer 3) -> SB ; oB <- b4
  flow_mfma_stage<4, 1, TERMS>(nh + hL4 / 4, SA, oA, [&](int i) {
    if (TERMS == 3) { flow_post_pair<TERMS>(i, aB, SB, nullptr, h); }
    else { flow_post_pair<TERMS>(4 * i, aB, SB, nullptr, h); flow_post_pair<TERMS>(4 * i + 1, aB, SB, nullptr, h);
           flow_post_pair<TERMS>(4 * i + 2, aB, SB, nullptr, h); flow_post_pair<TERMS>(4 * i + 3, aB, SB, nullptr, h); }
  });
  if (TERMS == 3) {
#pragma unroll
    for (int p = 12; p < 16; ++p) flow_post_pair<TERMS>(p, aB, SB, nullptr, h);
  }
  asm volatile("s_nop 1");
#pragma unroll
  for (int j = 0; j < 16; ++j) oB[0][j] = net[hB4 + j * 2 + h];
  // stage 8: L4(B)
  flow_mfma_stage<4, 1, TERMS>(nh + hL4 / 4, SB, oB, none);
}

// Two tiles (A: rows 0..31, B: rows 32..63 of a 64-row group) share one spline pass: lane l evaluates the spline of row l,
// i.e. lanes 0..31 need all 32 outputs of tile A's column (lane & 31) and lanes 32..63 those of tile B's.  Each lane keeps
// the 16 outputs it holds of ITS tile and receives the other 16 from its partner lane (l ^ 32), which holds them in the
// accumulators of the tile it does not need itself.
__device__ __forceinline__ void gather_outputs(const f32x16 (&oA)[1], const f32x16 (&oB)[1], int h, float (&wv)[32]) {
  // v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second: afterwards the first holds
  // {A's lanes 0..31 | B's lanes 0..31} -- for every lane the outputs rho(j, 0) of ITS tile -- and the second {A's 32..63 | B's 32..63},
  // the outputs rho(j, 1).  One instruction per register instead of four selects and a cross-lane read.
  (void)h;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(oA[0][j]), __float_as_uint(oB[0][j]), false, false);
    const int r0 = (j & 3) + 8 * (j >> 2);
    wv[r0] = __uint_as_float(r[0]);
    wv[r0 + 4] = __uint_as_float(r[1]);
  }
}

// Occupancy: the kernel alternates matrix-core phases (the nets) and long vector phases (the spline); neither unit is busy more than
// ~37 % of the time with two waves per SIMD, so the f16 variants are compiled for THREE (768 threads per workgroup, <= 168 VGPRs).
#ifndef FLOW_WPB_H3
#define FLOW_WPB_H3 12
#endif
#ifndef FLOW_PIPELINED
#define FLOW_PIPELINED 0   // 1: coupling_pair (the two tiles of a group interleaved in one instruction stream).  Measured, bench's two launches:
                           // one wave per SIMD 12.4 -> 11.7 ms, two 10.8 -> 10.6, three (default) 10.5 -> 10.7: what the interleave hides
                           // inside a wave, a second and third wave already hide across waves.  tools/mfma_chain.hip: one wave alone runs a
                           // 32x32x16 MFMA every 32.8 cycles, 40.7 with 7 vector instructions behind two of every three (18.7 cycles of
                           // issue, 8 exposed), and waves that share a SIMD take turns by age rather than filling each other's gaps.
#endif
template <bool SAMPLE, int MODE>
__global__ void __launch_bounds__(MODE == 0 ? 512 : 64 * FLOW_WPB_H3) flow_kernel(const float* __restrict__ netfrag /*[2][kNetFloats]*/,
                                                   const float* __restrict__ P /*[2][pn][64]*/,
                                                   const float* __restrict__ latent, const float* __restrict__ jitter,
                                                   const float* __restrict__ xin, const long long* __restrict__ rays_id,
                                                   long long m, int sn, long long pn, float* __restrict__ out_xy,
                                                   float* __restrict__ out_lj, int* __restrict__ bins) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool H3 = MODE != 0;
  constexpr int NF = H3 ? hNetFloats : kNetFloats;
  for (int i = threadIdx.x; i < 2 * NF; i += blockDim.x) lds[i] = netfrag[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  // one iteration = 64 rows: lane l owns row 64*group + l (state, spline, output); the nets run per 32-row tile with row
  // (lane & 31) resp. 32 + (lane & 31) on this lane's MFMA column, so every spline is evaluated once (not once per half)
  const long long n_groups = (m + 63) / 64;
  const int h = lane >> 5, col = lane & 31;
#ifdef FLOW_STAGGER   // dev-only experiment: waves that share a SIMD (wave, wave + 4, wave + 8) start a third of a group apart
  for (int d = 0; d < (wave >> 2); ++d) __builtin_amdgcn_s_sleep(FLOW_STAGGER);
#endif
  for (long long grp = (long long)blockIdx.x * waves_per_block + wave; grp < n_groups;
       grp += (long long)gridDim.x * waves_per_block) {
    asm volatile("" ::: "memory");  // keep the LDS weight fragments out of registers across tiles (LICM)
    long long row = grp * 64 + lane;
    const bool valid = row < m;
    if (!valid) row = m - 1;
    long long pt;
    float x0, x1, lj;
    if (SAMPLE) {
      pt = row / sn;
      int s = (int)(row % sn);
      x0 = latent[2 * s];
      x1 = latent[2 * s + 1];
      if (jitter) { x0 = x0 + jitter[row]; x0 = x0 - floorf(x0); }   // (x + u) % 1
      x0 = fminf(fmaxf(x0, 1e-6f), 1.f - 1e-6f);
      x1 = fminf(fmaxf(x1, 1e-6f), 1.f - 1e-6f);
      { float sn_, cs_; tf_sincos_small(x1 * kHalfPi, sn_, cs_); lj = -tf_log(cs_); }   // x1 in (0,1): argument < pi/2
    } else {
      pt = rays_id ? rays_id[row] : row / sn;
      x0 = fminf(fmaxf(xin[2 * row], 1e-6f), 1.f - 1e-6f);
      x1 = fminf(fmaxf(xin[2 * row + 1], 1e-6f), 1.f - 1e-6f);
      lj = 0.f;
    }
#ifdef FLOW_STAMPS   // dev-only: shader-clock stamps of one 64-row group of workgroup 0 (its 9th), per wave
    unsigned long long st[12]; int n_st = 0;
    const bool st_on = blockIdx.x == 0 && grp == (long long)wave + 8LL * gridDim.x * waves_per_block;
#define FLOW_STAMP() do { if (st_on && n_st < 12) st[n_st++] = __builtin_readcyclecounter(); } while (0)
#else
#define FLOW_STAMP() do {} while (0)
#endif
    FLOW_STAMP();
    const int ptA = __shfl((int)pt, col), ptB = __shfl((int)pt, 32 + col);   // pn < 2^31 (checked by the launcher)
    float wv[32];
    f32x16 oA[1], oB[1];
    int bin0, bin1;
    float t, l;
    // one coupling block for both tiles: `keep` is the conditioning coordinate of this lane's own row
    auto run_block = [&](const float* net, long long pbase, float keep) {
      const float kA = __shfl(keep, col), kB = __shfl(keep, 32 + col);
#if FLOW_PIPELINED
      if (MODE != 0) {
        coupling_pair<MODE == 0 ? 3 : MODE>(net, P + (pbase + ptA) * 64, P + (pbase + ptB) * 64, kA, kB, lane, oA, oB);
        FLOW_STAMP(); FLOW_STAMP();
      } else
#endif
      {
      coupling_net<MODE>(net, P + (pbase + ptA) * 64, kA, lane, oA);
      FLOW_STAMP();
      coupling_net<MODE>(net, P + (pbase + ptB) * 64, kB, lane, oB);
      FLOW_STAMP();
      }
      gather_outputs(oA, oB, h, wv);
    };
    if (SAMPLE) {
      run_block(lds, 0, x0);                                             // block 0 keeps x0, moves x1
      pw_inverse(x1, wv, t, l, bin0); x1 = t; lj += l;
      FLOW_STAMP();
      run_block(lds + NF, pn, x1);                                       // block 1 keeps x1, moves x0
      pw_inverse(x0, wv, t, l, bin1); x0 = t; lj += l;
      FLOW_STAMP();
    } else {
      run_block(lds + NF, pn, x1);
      pw_forward(x0, wv, t, l, bin1); x0 = t; lj += l;
      run_block(lds, 0, x0);
      pw_forward(x1, wv, t, l, bin0); x1 = t; lj += l;
      { float sn_, cs_; tf_sincos_small(x1 * kHalfPi, sn_, cs_); lj += tf_log(cs_); }   // + latent_prior.log_prob(z)
    }
    if (valid) {
      reinterpret_cast<float2*>(out_xy)[row] = make_float2(x0, x1);
      out_lj[row] = lj;
      if (bins) reinterpret_cast<int2*>(bins)[row] = make_int2(bin0, bin1);
    }
#ifdef FLOW_STAMPS
    FLOW_STAMP();
    if (SAMPLE && st_on && lane == 0 && n_st == 8)
      printf("flow<%d> wave %d: net A %llu | net B %llu | gather + spline %llu | net A %llu | net B %llu | gather + spline %llu | store %llu | group %llu ticks\n", MODE, wave,
             st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6], st[7] - st[0]);
#endif
  }
}

// per-point hoisted layer-1 part: P[b][pt][u] = b1[u] + sum_k W1[u][7+k] * (2*cond[pt][k] - 1)
// A lane owns unit u of one coupling net and keeps its 37 weights in registers; its wave walks FLOW_PP_PTS points, whose condition rows
// are wave-uniform (scalar loads), one coalesced 256-byte store per point.  (Rounds 1-5: one thread per output re-read its weight row --
// 64 different rows per wave-wide load, 37 of them per output: 0.54 ms per 262 144 points, 8 launches per bench step; same sum order.)
#define FLOW_PP_PTS 32
__global__ void __launch_bounds__(256) flow_point_part_kernel(const float* __restrict__ w1a, const float* __restrict__ b1a,
                                                              const float* __restrict__ w1b, const float* __restrict__ b1b,
                                                              const float* __restrict__ cond, long long pn,
                                                              float* __restrict__ P) {
  const int u = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int blk = blockIdx.y;
  const float* w = (blk ? w1b : w1a) + u * 44 + 7;
  float wr[37];
#pragma unroll
  for (int k = 0; k < 37; ++k) wr[k] = w[k];
  const float b = (blk ? b1b : b1a)[u];
  const long long p0 = ((long long)blockIdx.x * 4 + wave) * FLOW_PP_PTS;
  for (int i = 0; i < FLOW_PP_PTS; ++i) {
    const long long pt = p0 + i;
    if (pt >= pn) break;                                     // wave-uniform
    const float* c = cond + pt * 37;
    float acc = b;
#pragma unroll
    for (int k = 0; k < 37; ++k) acc += wr[k] * (c[k] * 2.f - 1.f);
    P[((long long)blk * pn + pt) * 64 + u] = acc;
  }
}

static int pack_nets_h3(const TfCouplingNet nets[2], float* netfrag, hipStream_t stream) {
  TfPackBatch PB(stream);           // both coupling nets in ONE launch (was 14)
  for (int b = 0; b < 2; ++b) {
    float* base = netfrag + (size_t)b * hNetFloats;
    for (int l = 0; l < 4; ++l)
      TF_REQUIRE(nets[b].w[l] && nets[b].b[l], TF_EINVAL, "tf_flow: null weight pointer (block %d layer %d)", b, l);
    _Float16* hb = reinterpret_cast<_Float16*>(base);
    PB.wfrag_h3(nets[b].w[0], 64, 44, 0, 7, 2, 1, hb + 2 * (size_t)hL1);
    PB.wfrag_h3(nets[b].w[1], 64, 64, 0, 64, 2, 4, hb + 2 * (size_t)hL2);
    PB.wfrag_h3(nets[b].w[2], 64, 64, 0, 64, 2, 4, hb + 2 * (size_t)hL3);
    PB.wfrag_h3(nets[b].w[3], 21, 64, 0, 64, 1, 4, hb + 2 * (size_t)hL4);
    PB.bias(nets[b].b[1], 64, 2, base + hB2);
    PB.bias(nets[b].b[2], 64, 2, base + hB3);
    PB.bias(nets[b].b[3], 21, 1, base + hB4);
  }
  PB.flush();
  return TF_OK;
}

static int pack_nets(const TfCouplingNet nets[2], float* netfrag, hipStream_t stream) {
  TfPackBatch PB(stream);
  for (int b = 0; b < 2; ++b) {
    float* base = netfrag + (size_t)b * kNetFloats;
    for (int l = 0; l < 4; ++l)
      TF_REQUIRE(nets[b].w[l] && nets[b].b[l], TF_EINVAL, "tf_flow: null weight pointer (block %d layer %d)", b, l);
    PB.wfrag(nets[b].w[0], 64, 44, 0, 7, 2, 4, base + kL1);
    PB.wfrag(nets[b].w[1], 64, 64, 0, 64, 2, 32, base + kL2);
    PB.wfrag(nets[b].w[2], 64, 64, 0, 64, 2, 32, base + kL3);
    PB.wfrag(nets[b].w[3], 21, 64, 0, 64, 1, 32, base + kL4);
    PB.bias(nets[b].b[1], 64, 2, base + kB2);
    PB.bias(nets[b].b[2], 64, 2, base + kB3);
    PB.bias(nets[b].b[3], 21, 1, base + kB4);
  }
  PB.flush();
  return TF_OK;
}

static constexpr int kWsNet = 2 * (hNetFloats > kNetFloats ? hNetFloats : kNetFloats);
extern "C" size_t tf_flow_workspace_floats(int64_t pn) { return (size_t)kWsNet + (size_t)2 * 64 * (size_t)(pn > 0 ? pn : 0); }

template <bool SAMPLE>
static int flow_launch(const TfCouplingNet nets[2], const float* cond, const float* latent, const float* jitter,
                       const float* x, const int64_t* rays_id, int64_t m, int32_t sn, int64_t pn, float* out_xy,
                       float* out_lj, int32_t* bins, int32_t precision, float* workspace, size_t workspace_floats,
                       hipStream_t stream, const char* who) {
  const bool packed = (precision & TF_WEIGHTS_PACKED) != 0;
  precision &= ~TF_WEIGHTS_PACKED;
  TF_REQUIRE(precision == TF_PREC_F32 || precision == TF_PREC_F16X3 || precision == TF_PREC_F16, TF_EINVAL,
             "%s: unknown precision %d", who, precision);
  TF_REQUIRE(m >= 0 && pn >= 0 && sn > 0, TF_ESHAPE, "%s: negative size / sn <= 0", who);
  if (m == 0) return TF_OK;
  TF_REQUIRE(nets && cond && out_xy && out_lj && workspace, TF_EINVAL, "%s: null pointer", who);
  TF_REQUIRE(pn > 0, TF_ESHAPE, "%s: pn == 0 with m > 0", who);
  TF_REQUIRE(workspace_floats >= tf_flow_workspace_floats(pn), TF_ESHAPE, "%s: workspace too small (%zu < %zu floats)", who,
             workspace_floats, tf_flow_workspace_floats(pn));
  if (SAMPLE) {
    TF_REQUIRE(latent, TF_EINVAL, "%s: latent is null", who);
    TF_REQUIRE(m == pn * (int64_t)sn, TF_ESHAPE, "%s: m != pn*sn", who);
  } else {
    TF_REQUIRE(x, TF_EINVAL, "%s: x is null", who);
    TF_REQUIRE(rays_id || m == pn * (int64_t)sn, TF_ESHAPE, "%s: without rays_id m must equal pn*sn", who);
  }
  float* netfrag = workspace;
  float* P = workspace + kWsNet;
  const bool h3 = precision != TF_PREC_F32;      // TF_PREC_F16 runs on the f16x3 fragment image (hi halves only)
  if (!packed)
    if (int rc = h3 ? pack_nets_h3(nets, netfrag, stream) : pack_nets(nets, netfrag, stream)) return rc;
  flow_point_part_kernel<<<dim3((unsigned)tf_blocks(pn, 4 * FLOW_PP_PTS), 2), 256, 0, stream>>>(nets[0].w[0], nets[0].b[0], nets[1].w[0],
                                                                                                nets[1].b[0], cond, pn, P);
  const size_t lds = (size_t)kWsNet * sizeof(float);
  static std::atomic<unsigned long long> attr_set{0};
  int attr_dev;
  if (tf_once_needed(attr_set, &attr_dev)) {
    hipFuncSetAttribute((const void*)flow_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)flow_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)flow_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)flow_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)flow_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)flow_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    tf_once_done(attr_set, attr_dev);
  }
  TF_REQUIRE(pn < (1LL << 31), TF_ESHAPE, "%s: pn must be < 2^31", who);
  const long long tiles = (m + 63) / 64;   // 64-row groups
  // tf_set_launch_budget: 4 or 8 waves (one / two per SIMD) leave two thirds / one third of a CU's registers to another stream's kernel
  const int wpb_budget = tf_launch_budget().flow_waves_per_block;
  const int waves_per_block = h3 ? (wpb_budget > 0 && wpb_budget < FLOW_WPB_H3 ? wpb_budget : FLOW_WPB_H3) : (wpb_budget == 4 ? 4 : 8);
  long long blocks = (tiles + waves_per_block - 1) / waves_per_block;
  if (blocks > 256) blocks = 256;  // one resident 8-wave workgroup per CU; waves loop over tiles
  if (precision == TF_PREC_F16)
    flow_kernel<SAMPLE, 1><<<(unsigned)blocks, 64 * waves_per_block, lds, stream>>>(
        netfrag, P, latent, jitter, x, (const long long*)rays_id, m, sn, pn, out_xy, out_lj, bins);
  else if (h3)
    flow_kernel<SAMPLE, 3><<<(unsigned)blocks, 64 * waves_per_block, lds, stream>>>(
        netfrag, P, latent, jitter, x, (const long long*)rays_id, m, sn, pn, out_xy, out_lj, bins);
  else
    flow_kernel<SAMPLE, 0><<<(unsigned)blocks, 64 * waves_per_block, lds, stream>>>(
        netfrag, P, latent, jitter, x, (const long long*)rays_id, m, sn, pn, out_xy, out_lj, bins);
  TF_LAUNCH_CHECK(who);
  return TF_OK;
}

extern "C" int tf_flow_sample_fwd(const TfCouplingNet nets[2], const float* cond, const float* latent, const float* jitter,
                                  int64_t pn, int32_t sn, float* angles, float* logj, int32_t* bins, int32_t precision,
                                  float* workspace, size_t workspace_floats, tf_stream_t stream) {
  return flow_launch<true>(nets, cond, latent, jitter, nullptr, nullptr, pn * (int64_t)sn, sn, pn, angles, logj, bins,
                           precision, workspace, workspace_floats, (hipStream_t)stream, "tf_flow_sample_fwd");
}

extern "C" int tf_flow_logq_fwd(const TfCouplingNet nets[2], const float* cond, const float* x, const int64_t* rays_id,
                                int64_t m, int32_t sn, int64_t pn, float* z, float* logq, int32_t* bins, int32_t precision,
                                float* workspace, size_t workspace_floats, tf_stream_t stream) {
  return flow_launch<false>(nets, cond, nullptr, nullptr, x, rays_id, m, sn, pn, z, logq, bins, precision, workspace,
                            workspace_floats, (hipStream_t)stream, "tf_flow_logq_fwd");
}

// ---------------------------------------------------------------- the spline alone (the kernels above call pw_inverse / pw_forward
// on the coupling nets' outputs; this entry runs the SAME device functions on caller-supplied parameter rows, so that the
// reference's spline vectors -- tests/golden/pwquad.npz, incl. its degenerate rows -- are checked on the GPU too)
__global__ void __launch_bounds__(256) pwquad_kernel(const float* __restrict__ wv, const float* __restrict__ y, long long m, int inverse,
                                                     float* __restrict__ out, float* __restrict__ logj, int* __restrict__ bins) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  float row[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) row[k] = k < 21 ? wv[i * 21 + k] : 0.f;
  float o, lj;
  int bin;
  if (inverse) pw_inverse(y[i], row, o, lj, bin);
  else pw_forward(y[i], row, o, lj, bin);
  out[i] = o; logj[i] = lj;
  if (bins) bins[i] = bin;
}

extern "C" int tf_pwquad_eval(const float* wv, const float* y, int64_t m, int32_t inverse, float* out, float* logj, int32_t* bins,
                              tf_stream_t stream_) {
  TF_REQUIRE(m >= 0, TF_ESHAPE, "tf_pwquad_eval: m < 0");
  if (m == 0) return TF_OK;
  TF_REQUIRE(wv && y && out && logj, TF_EINVAL, "tf_pwquad_eval: null pointer");
  pwquad_kernel<<<tf_blocks(m, 256), 256, 0, (hipStream_t)stream_>>>(wv, y, m, inverse, out, logj, bins);
  TF_LAUNCH_CHECK("tf_pwquad_eval");
  return TF_OK;
}
