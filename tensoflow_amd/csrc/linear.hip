// Dense layers of the training direction: Y = act(X W^T + b) and its three backward products, on the matrix cores: exact fp32
// (v_mfma_f32_32x32x2_f32, the default) or the f16x3 operand split (three v_mfma_f32_32x32x16_f16 per product term, fp32 accumulate)
// for operands inside the f16 range -- NOT for raw gradients (1e-7 .. 1e-5 per element under a mean-reduced loss: they flush to zero),
// and measured no faster on these tall-skinny products (the tile traffic through LDS, not the matrix rate, holds them).
//
// The reference trains its small MLPs -- TensoSDF's decoder (network/fields.py:78-81), the material predictors and the inner-light
// net (network/other_field.py:50-119, fields.py:905-911,1010-1017), ShapeShadingNetwork's three 128-wide nets (fields.py:448-567)
// -- through torch.nn.Linear, i.e. library GEMMs.  Here every such product in a training step is one launch of ONE tiled kernel
// on v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain per output element; 157 TF/s peak):
//     forward        Y  [n,N] = act(X [n,K] . W^T + b)
//     backward data  gX [n,K] = gZ [n,N] . W [N,K]            gZ = gY * act'(Y)      (tf_linear_bwd computes gZ in place of gY's copy)
//     backward weight gW [N,K] = gZ^T . X   (reduction over the n rows split across workgroups, fp32 atomics), gb = column sums
// n is 1e3 .. 1e6 rows, K and N <= 256: "tall-skinny" products.  One workgroup = 4 waves = a 128 x 64 tile of C, K walked in
// chunks of 16 through LDS ([k][m] / [k][n] images: a lane's MFMA operand is one dword, consecutive lanes consecutive addresses).
#include "mfma_mlp.h"
#include <cstdlib>
#include "tf_common.h"
#include "tf_internal.h"

#ifndef TF_GEMM2
#define TF_GEMM2 1      // 0 (dev switch): every product on the register-staged kernel
#endif
#include <stdint.h>

namespace {
constexpr int BM = 128, BN = 64, KC = 16;

struct GemmArgs {
  const float* A; long long sAm, sAk;      // A(m, k) = A[m * sAm + k * sAk]
  const float* B; long long sBk, sBn;      // B(k, n) = B[k * sBk + n * sBn]
  float* C; long long sCm;                 // C(m, n) = C[m * sCm + n]
  const float* bias;                       // [N] or null (forward epilogue)
  long long M, N, K;                       // K = reduction length
  long long k_split;                       // reduction elements per blockIdx.z slice (== K when gridDim.z == 1)
  int act; float act_param;
  int atomic;                              // 1: C += partial (split reduction)
  const long long* n_dev;                  // device-side row count (null: none): clamps M (rows_are_m) or the reduction length
  int rows_are_m;
};

__device__ __forceinline__ float act_fwd(float z, int act, float p) {
  switch (act) {
    case TF_ACT_RELU: return fmaxf(z, 0.f);
    case TF_ACT_SOFTPLUS: {                 // torch.nn.Softplus(beta = p, threshold = 20)
      const float bz = p * z;
      return bz > 20.f ? z : log1pf(expf(bz)) / p;
    }
    case TF_ACT_SIGMOID: return 1.f / (1.f + expf(-z));
    case TF_ACT_EXP_CLAMP: return expf(fminf(z, p));
    default: return z;
  }
}

// d act / d z expressed through the OUTPUT y = act(z) (what the forward pass keeps)
__device__ __forceinline__ float act_bwd_from_y(float y, int act, float p) {
  switch (act) {
    case TF_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case TF_ACT_SOFTPLUS: return 1.f - expf(-p * y);          // sigmoid(p z) = 1 - exp(-p softplus(z))
    case TF_ACT_SIGMOID: return y * (1.f - y);
    case TF_ACT_EXP_CLAMP: return y < expf(p) ? y : 0.f;      // clamp(z, max = p): zero slope above the clamp
    default: return 1.f;
  }
}

// A_KFAST / B_NFAST: which index of the operand is contiguous in memory (decides the coalescing of the tile loads)
template <bool A_KFAST, bool B_NFAST, bool H3>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs G) {
  __shared__ float As[KC][BM + 4];
  __shared__ float Bs[KC][BN + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (G.n_dev) {       // the caller sized the launch for a capacity; the number of valid rows is only known on the device
    const long long nv = max(0LL, *G.n_dev);
    if (G.rows_are_m) G.M = min(G.M, nv); else G.K = min(G.K, nv);
  }
  const long long m0 = (long long)blockIdx.x * BM, n0 = (long long)blockIdx.y * BN;
  const long long kb = (long long)blockIdx.z * G.k_split, ke = min(kb + G.k_split, G.K);
  if (m0 >= G.M || kb >= ke) return;                       // wave-uniform
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  // Tile (A: [KC][BM], B: [KC][BN]) of reduction chunk k0: fetched global -> registers, then registers -> LDS, so that the loads of
  // chunk c + 1 are in flight while the MFMAs of chunk c run (a tall-skinny product has few workgroups per CU to hide them behind).
  float ra[8], rb[4];
  auto fetch = [&](long long k0) {
    if (A_KFAST) {          // 4 lanes cover the 16 k of one row (64 contiguous bytes), 64 rows per pass
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const long long m = m0 + p * 64 + (tid >> 2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const long long k = k0 + (tid & 3) * 4 + c;
          ra[4 * p + c] = (m < G.M && k < ke) ? G.A[m * G.sAm + k * G.sAk] : 0.f;
        }
      }
    } else {                // m contiguous: consecutive lanes consecutive m
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int e = p * 256 + tid;
        const long long m = m0 + e % BM, k = k0 + e / BM;
        ra[p] = (m < G.M && k < ke) ? G.A[m * G.sAm + k * G.sAk] : 0.f;
      }
    }
    if (B_NFAST) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int e = p * 256 + tid;
        const long long n = n0 + e % BN, k = k0 + e / BN;
        rb[p] = (n < G.N && k < ke) ? G.B[k * G.sBk + n * G.sBn] : 0.f;
      }
    } else {                // k contiguous (W [N,K] read as B(k, n) = W[n K + k])
      const long long n = n0 + (tid >> 2);
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const long long k = k0 + (tid & 3) * 4 + cc;
        rb[cc] = (n < G.N && k < ke) ? G.B[k * G.sBk + n * G.sBn] : 0.f;
      }
    }
  };
  auto stash = [&]() {
    if (A_KFAST) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int c = 0; c < 4; ++c) As[(tid & 3) * 4 + c][p * 64 + (tid >> 2)] = ra[4 * p + c];
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) { const int e = p * 256 + tid; As[e / BM][e % BM] = ra[p]; }
    }
    if (B_NFAST) {
#pragma unroll
      for (int p = 0; p < 4; ++p) { const int e = p * 256 + tid; Bs[e / BN][e % BN] = rb[p]; }
    } else {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) Bs[(tid & 3) * 4 + cc][tid >> 2] = rb[cc];
    }
  };
  fetch(kb);
  stash();
  __syncthreads();
  for (long long k0 = kb; k0 < ke; k0 += KC) {
    const bool more = k0 + KC < ke;
    if (more) fetch(k0 + KC);
    // ---- wave `wave`: rows [32 wave, 32 wave + 32) x all 64 columns
    if (H3) {
      // the chunk is ONE k-step of the f16 instruction: lane half h supplies k = 8 h .. 8 h + 7 of its row (A) / column (B), split on
      // the fly into hi + lo halves (tf_split8: 12 vector instructions per 8 values); hi.hi + hi.lo + lo.hi, the lo.lo term (2^-22) dropped
      const int h = lane >> 5, i = lane & 31;
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = As[8 * h + e][32 * wave + i];
      tf_h8 a_hi, a_lo;
      tf_split8(v8, a_hi, a_lo);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = Bs[8 * h + e][32 * t + i];
        tf_h8 b_hi, b_lo;
        tf_split8(v8, b_hi, b_lo);
        acc[t] = tf_mfma_h(a_hi, b_hi, acc[t]);
        acc[t] = tf_mfma_h(a_hi, b_lo, acc[t]);
        acc[t] = tf_mfma_h(a_lo, b_hi, acc[t]);
      }
    } else {
      // exact fp32: lane supplies A[i = lane & 31][k = lane >> 5] of each pair of k
#pragma unroll
      for (int s = 0; s < KC / 2; ++s) {
        const float a = As[2 * s + (lane >> 5)][32 * wave + (lane & 31)];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = tf_mfma(a, Bs[2 * s + (lane >> 5)][32 * t + (lane & 31)], acc[t]);
      }
    }
    __syncthreads();
    if (more) {
      stash();
      __syncthreads();
    }
  }
  // ---- epilogue: lane holds D[i = rho(reg, lane >> 5)][j = lane & 31] of each 32 x 32 tile
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long long n = n0 + 32 * t + (lane & 31);
    if (n >= G.N) continue;
    const float bv = G.bias ? G.bias[n] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const long long m = m0 + 32 * wave + tf_rho(reg, lane >> 5);
      if (m >= G.M) continue;
      float* dst = G.C + m * G.sCm + n;
      if (G.atomic) atomicAdd(dst, acc[t][reg]);
      else *dst = act_fwd(acc[t][reg] + bv, G.act, G.act_param);
    }
  }
}

template <bool A_KFAST, bool B_NFAST>
int launch(const GemmArgs& G, int splits, bool h3, hipStream_t stream, const char* who) {
  dim3 grid((unsigned)((G.M + BM - 1) / BM), (unsigned)((G.N + BN - 1) / BN), (unsigned)splits);
  if (h3) gemm_kernel<A_KFAST, B_NFAST, true><<<grid, 256, 0, stream>>>(G);
  else gemm_kernel<A_KFAST, B_NFAST, false><<<grid, 256, 0, stream>>>(G);
  TF_LAUNCH_CHECK(who);
  return TF_OK;
}

// ----------------------------------------------------------------------------------------------------------------------------
// The same three products for ALIGNED operands (every leading stride a multiple of four floats: all 128- and 256-wide layers), as a
// software pipeline that keeps the matrix cores fed.  What held the kernel above at 44 TF/s of 157 (measured, 236 k x 256 x 256): its
// tile loads pass through registers one 16-deep chunk ahead, i.e. 1 024 cycles of MFMAs per wave stand against an HBM / L2 round trip of
// 2-4 k cycles, and its 88-126 registers leave 3 waves per SIMD to cover the difference.  Here:
//   * tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), double buffered: the loads of
//     chunk c + 1 are issued before the MFMAs of chunk c, one s_barrier per chunk, four workgroups resident per CU;
//   * a workgroup owns 128 x 128 of C, a wave 64 x 64 (2 x 2 MFMA tiles): every operand fragment read from LDS feeds two MFMAs, 32 MFMAs
//     (2 048 cycles) per wave and chunk;
//   * operand images in LDS follow the operand's contiguous index.  "Reduction-fast" (X [m][k], W [n][k], gZ [m][n] as the A operand
//     of the data gradient): a row's 16 reduction elements are four 16-byte quads, stored at slot 4 r + (q ^ ((r >> 1) & 3)) of its
//     16-row block, so that a lane's `ds_read_b128` of quad q (rows r = 0..31 across the lanes) is conflict free; lane half h takes quads
//     2h, 2h + 1 = reduction elements 8h .. 8h + 7, element t of them in MFMA t (the pairing of the two k of v_mfma_f32_32x32x2_f32 is
//     free as long as A and B agree).  "Output-fast" (W [n][k] as B of the data gradient, gZ and X in the weight gradient): image
//     [16 reduction rows][128 outputs], conflict-free `ds_read_b32`, the same element -> MFMA pairing.
//   * out-of-range quads (reduction tail, rows past a device-side count) are fetched from a 16-byte zero word instead of being
//     predicated: every lane always issues its DMA, the vmcnt arithmetic stays uniform.
#ifndef TF_G2_NST
#define TF_G2_NST 2     // LDS stages.  Measured (236 k x 256 x 256 forward / 128 x 128): 2 stages 78.9 / 68.8 TF/s, 3 stages 75.4 / 57.9,
                        // 4 stages 63.9 / 47.0 (two waves per SIMD) -- more resident workgroups beat a deeper prefetch; reading the next
                        // chunk's fragments under the current chunk's MFMAs (a second register set) changed nothing (73-75): a workgroup's
                        // life is 16 chunks, what is left is its prologue round trip and its epilogue's scattered stores
#endif
constexpr int G2_NST = TF_G2_NST;
__device__ __attribute__((aligned(16))) float g2_zero[4];

struct Gemm2Args {
  const float* A; long long sA;            // reduction-fast: A(row, red) = A[row sA + red]; output-fast: A(red, row) = A[red sA + row]
  const float* B; long long sB;            // likewise with (col, red)
  float* C; long long sC;
  const float* bias;
  long long M, N, K, k_split;
  int act; float act_param; int atomic;
  const long long* n_dev; int rows_are_m;
  // data-gradient product of a layer CHAIN (tf_linear_bwd_fused): C(m, n) *= act'(mulY(m, n)) with mulY [M][sC] the post-activation
  // output of the layer below (= this layer's input), and colsum[n] += sum_m C(m, n) -- that layer's bias gradient
  const float* mulY = nullptr; int mul_act = 0; float mul_param = 0.f; float* colsum = nullptr;
};

// SPLIT (round 5; the default, see launch2): every operand element as three bf16 (x = hi + mid + lo, 24 significant bits, fp32's exponent
// range -- the operands are gradients as often as activations, whose scale f16 does not cover), the product as six
// v_mfma_f32_32x32x16_bf16 (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid: everything down to 2^-24 of |a||b|) instead of eight
// v_mfma_f32_32x32x2_f32: 192 instead of 512 matrix cycles per 16 reduction elements, against ~11 vector instructions per pair of
// operand values for the split (a lane's eight values of a fragment ARE the bf16 instruction's operand: k = 8 h + e).
typedef __bf16 tf_b8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned g2_cvt2(float a, float b) {
  typedef float f2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
  f2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_t));
}
__device__ __forceinline__ void g2_split8(const float (&x)[8], tf_b8& hi, tf_b8& mid, tf_b8& lo) {
  typedef unsigned u4_t __attribute__((ext_vector_type(4)));
  u4_t h, m, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = x[2 * p], x1 = x[2 * p + 1];
    const unsigned hh = g2_cvt2(x0, x1);
    const float r0 = x0 - __uint_as_float(hh << 16), r1 = x1 - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = g2_cvt2(r0, r1);
    const float s0 = r0 - __uint_as_float(mm << 16), s1 = r1 - __uint_as_float(mm & 0xffff0000u);
    h[p] = hh; m[p] = mm; l[p] = g2_cvt2(s0, s1);
  }
  // (Non-finite operands: an Inf -- or a value that rounds up to the bf16 Inf, |x| > 3.3895e38 -- has hi = Inf and NaN residuals, and even
  // with those zeroed its products with the OTHER operand's signed residual planes are +Inf and -Inf: the split cannot propagate an Inf
  // as the fp32 instruction does.  The row such an operand reaches comes out non-finite (NaN), every other row is untouched; a caller
  // that needs IEEE Inf semantics asks for TF_PREC_F32.  tests/test_gpu_linear.py pins this.)
  hi = __builtin_bit_cast(tf_b8, h); mid = __builtin_bit_cast(tf_b8, m); lo = __builtin_bit_cast(tf_b8, l);
}
__device__ __forceinline__ f32x16 g2_mfma_b(tf_b8 a, tf_b8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

template <bool A_RF, bool B_RF, bool SPLIT = false>
__global__ void __launch_bounds__(256, SPLIT ? 3 : 4) gemm2_kernel(Gemm2Args G) {
  __shared__ __attribute__((aligned(16))) float lds[G2_NST][2][128 * 16];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  if (G.n_dev) {
    const long long nv = max(0LL, *G.n_dev);
    if (G.rows_are_m) G.M = min(G.M, nv); else G.K = min(G.K, nv);
  }
  const long long m0 = (long long)blockIdx.x * 128, n0 = (long long)blockIdx.y * 128;
  const long long kb = (long long)blockIdx.z * G.k_split, ke = min(kb + G.k_split, G.K);
  if (m0 >= G.M || kb >= ke) return;                       // workgroup-uniform
  const int nchunk = (int)((ke - kb + 15) / 16);
  typedef const __attribute__((address_space(1))) float* gp_t;
  const gp_t zero = (gp_t)g2_zero;
  // ---- this lane's two DMA sources per operand and chunk (blocks 2 wave, 2 wave + 1 of the operand's eight 1 KB blocks)
  auto issue = [&](int c, int st) {
    const long long k0 = kb + 16LL * c;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const bool rf = op == 0 ? A_RF : B_RF;
      const float* base = op == 0 ? G.A : G.B;
      const long long stride = op == 0 ? G.sA : G.sB, o0 = op == 0 ? m0 : n0, bound = op == 0 ? G.M : G.N;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int blk = 2 * wave + j;
        gp_t src;
        if (rf) {
          const int rl = lane >> 2, q = (lane & 3) ^ ((rl >> 1) & 3);
          const long long row = min(o0 + 16 * blk + rl, bound - 1), red = k0 + 4 * q;
          src = red < ke ? (gp_t)(base + row * stride + red) : zero;
        } else {
          const long long red = k0 + 2 * blk + (lane >> 5), out = o0 + 4 * (lane & 31);
          src = (red < ke && out < bound) ? (gp_t)(base + red * stride + out) : zero;
        }
        const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) float*)(&lds[st][op][blk * 256]));
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(src) : "memory");
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
  for (int p = 0; p < G2_NST - 1; ++p)
    if (p < nchunk) issue(p, p);
  for (int c = 0; c < nchunk; ++c) {
    const int st = c % G2_NST;
    // chunk c has landed (the G2_NST - 2 chunks behind it may still be in flight: four DMAs per wave and chunk)
    const int behind = min(nchunk - 1 - c, G2_NST - 2);
    if (behind >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                                // ... for every wave; and every wave is done reading chunk c - 1
    asm volatile("" ::: "memory");
    if (c + G2_NST - 1 < nchunk) issue(c + G2_NST - 1, (c + G2_NST - 1) % G2_NST);
    float fa[2][8], fb[2][8];
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const bool rf = op == 0 ? A_RF : B_RF;
      const float* img = lds[st][op];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int r = (op == 0 ? 64 * wm : 64 * wn) + 32 * t2 + i;
        float* dst = op == 0 ? fa[t2] : fb[t2];
        if (rf) {
          const int blk = r >> 4, rl = r & 15;
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const float4 v = *reinterpret_cast<const float4*>(img + blk * 256 + 4 * (4 * rl + ((2 * h + qq) ^ ((rl >> 1) & 3))));
            dst[4 * qq] = v.x; dst[4 * qq + 1] = v.y; dst[4 * qq + 2] = v.z; dst[4 * qq + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t) dst[t] = img[(8 * h + t) * 128 + r];
        }
      }
    }
    if (SPLIT) {
      tf_b8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) { g2_split8(fa[t2], ah[t2], am[t2], al[t2]); g2_split8(fb[t2], bh[t2], bm[t2], bl[t2]); }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          f32x16 c = acc[a][b];
          c = g2_mfma_b(am[a], bm[b], c);
          c = g2_mfma_b(al[a], bh[b], c);
          c = g2_mfma_b(ah[a], bl[b], c);
          c = g2_mfma_b(am[a], bh[b], c);
          c = g2_mfma_b(ah[a], bm[b], c);
          acc[a][b] = g2_mfma_b(ah[a], bh[b], c);
        }
    } else {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = tf_mfma(fa[a][t], fb[b][t], acc[a][b]);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const long long n = n0 + 64 * wn + 32 * b + i;
      if (n >= G.N) continue;
      const float bv = G.bias ? G.bias[n] : 0.f;
      float cs = 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const long long m = m0 + 64 * wm + 32 * a + tf_rho(reg, h);
        if (m >= G.M) continue;
        float* dst = G.C + m * G.sC + n;
        if (G.atomic) atomicAdd(dst, acc[a][b][reg]);
        else if (G.mulY) {
          const float v = acc[a][b][reg] * act_bwd_from_y(G.mulY[m * G.sC + n], G.mul_act, G.mul_param);
          *dst = v;
          cs += v;
        }
        else *dst = act_fwd(acc[a][b][reg] + bv, G.act, G.act_param);
      }
      if (G.colsum) {                       // the two lane halves hold the other 16 rows of this column: one atomic per wave and column
        cs += __shfl_xor(cs, 32);
        if (h == 0) atomicAdd(G.colsum + n, cs);
      }
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// Layers with at most four outputs (the 256 -> 3 radiance layers, the 128 -> 1 / 128 -> 3 material heads): 61 of 64 tile columns of a
// matrix-core kernel would be padding (measured: 0.77 ms for the backward of 236 k x 256 -> 3, 25 x the time its bytes need).  Three
// streaming kernels instead, one row per wave / per thread group, fp32 fma:
//   forward      y[r][j]  = act(sum_k x[r][k] w[j][k] + b[j])    a lane owns four k of every 256, wave reduction
//   data         gx[r][k] = sum_j gz[r][j] w[j][k]               a lane owns four k
//   weight       gw[j][k] += sum_r gz[r][j] x[r][k]              a thread owns four k, a workgroup 1 024 rows, one atomic per (j, k) and workgroup
template <int NOUT>
__global__ void __launch_bounds__(256) thin_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
                                                       long long n, int K, int act, float p, float* __restrict__ Y,
                                                       const long long* __restrict__ n_dev) {
  if (n_dev) n = min(n, max(0LL, *n_dev));
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
  float bj[NOUT];
#pragma unroll
  for (int j = 0; j < NOUT; ++j) bj[j] = b ? b[j] : 0.f;
  for (long long r = wave; r < n; r += n_waves) {
    float s[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) s[j] = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
      const float4 x = *reinterpret_cast<const float4*>(X + r * K + k);
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(W + j * K + k);       // 1-4 KB in all: served by the vector L1
        s[j] = fmaf(x.w, w.w, fmaf(x.z, w.z, fmaf(x.y, w.y, fmaf(x.x, w.x, s[j]))));
      }
    }
#pragma unroll
    for (int j = 0; j < NOUT; ++j)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_xor(s[j], o);
    if (lane < NOUT) {
      float v = s[0];
#pragma unroll
      for (int j = 1; j < NOUT; ++j) v = lane == j ? s[j] : v;
      float bb = bj[0];
#pragma unroll
      for (int j = 1; j < NOUT; ++j) bb = lane == j ? bj[j] : bb;
      Y[r * NOUT + lane] = act_fwd(v + bb, act, p);
    }
  }
}

template <int NOUT>
__global__ void __launch_bounds__(256) thin_data_kernel(const float* __restrict__ gZ, const float* __restrict__ W, long long n, int K,
                                                        float* __restrict__ gX, const long long* __restrict__ n_dev) {
  if (n_dev) n = min(n, max(0LL, *n_dev));
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;       // one quad of four k
  const int kq = K / 4;
  const long long r = q / kq;
  if (r >= n) return;
  const int k = 4 * (int)(q - r * kq);
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NOUT; ++j) {
    const float g = gZ[r * NOUT + j];
    const float4 w = *reinterpret_cast<const float4*>(W + j * K + k);
    o.x = fmaf(g, w.x, o.x); o.y = fmaf(g, w.y, o.y); o.z = fmaf(g, w.z, o.z); o.w = fmaf(g, w.w, o.w);
  }
  *reinterpret_cast<float4*>(gX + r * K + k) = o;
}

// ... the same followed by the activation backward of the layer BELOW (tf_linear_bwd_fused): gZx[r][k] = gx[r][k] * act'(X[r][k]) and
// gbx[k] += sum_r gZx[r][k].  A thread keeps its four k and walks the rows of the workgroup's 1 024-row slab, so that the column sums
// leave the workgroup as one atomic per column.
template <int NOUT>
__global__ void __launch_bounds__(256) thin_data_mask_kernel(const float* __restrict__ gZ, const float* __restrict__ W, const float* __restrict__ X,
                                                             long long n, int K, int xact, float xp, float* __restrict__ gZx,
                                                             float* __restrict__ gbx, const long long* __restrict__ n_dev) {
  if (n_dev) n = min(n, max(0LL, *n_dev));
  const int kq = K / 4, rp = 256 / kq;
  const int t = threadIdx.x, sub = t / kq, k = 4 * (t - sub * kq);
  const long long r0 = (long long)blockIdx.x * 1024, r1 = min(r0 + 1024, n);
  if (r0 >= n) return;                                 // workgroup-uniform
  __shared__ float part[1024];
  for (int e = t; e < K; e += 256) part[e] = 0.f;
  __syncthreads();
  if (sub < rp) {
    float4 w[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) w[j] = *reinterpret_cast<const float4*>(W + j * K + k);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long r = r0 + sub; r < r1; r += rp) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const float g = gZ[r * NOUT + j];
        o.x = fmaf(g, w[j].x, o.x); o.y = fmaf(g, w[j].y, o.y); o.z = fmaf(g, w[j].z, o.z); o.w = fmaf(g, w[j].w, o.w);
      }
      const float4 x = *reinterpret_cast<const float4*>(X + r * K + k);
      o.x *= act_bwd_from_y(x.x, xact, xp); o.y *= act_bwd_from_y(x.y, xact, xp);
      o.z *= act_bwd_from_y(x.z, xact, xp); o.w *= act_bwd_from_y(x.w, xact, xp);
      *reinterpret_cast<float4*>(gZx + r * K + k) = o;
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    if (gbx) { atomicAdd(&part[k], s.x); atomicAdd(&part[k + 1], s.y); atomicAdd(&part[k + 2], s.z); atomicAdd(&part[k + 3], s.w); }
  }
  __syncthreads();
  if (gbx)
    for (int e = t; e < K; e += 256) atomicAdd(gbx + e, part[e]);
}

template <int NOUT>
__global__ void __launch_bounds__(256) thin_weight_kernel(const float* __restrict__ gZ, const float* __restrict__ X, long long n, int K,
                                                          float* __restrict__ gW, const long long* __restrict__ n_dev) {
  if (n_dev) n = min(n, max(0LL, *n_dev));
  const int kq = K / 4, rp = 256 / kq;                 // rp rows in flight per workgroup pass (K <= 1024)
  const int t = threadIdx.x, sub = t / kq, k = 4 * (t - sub * kq);
  const long long r0 = (long long)blockIdx.x * 1024, r1 = min(r0 + 1024, n);
  if (r0 >= n) return;                                 // workgroup-uniform
  __shared__ float part[NOUT * 1024];                  // [j][k]: the row groups of the workgroup meet here, then one atomic per (j, k)
  for (int e = t; e < NOUT * K; e += 256) part[e] = 0.f;
  __syncthreads();
  if (sub < rp) {
    float4 a[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long r = r0 + sub; r < r1; r += rp) {
      const float4 x = *reinterpret_cast<const float4*>(X + r * K + k);
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const float g = gZ[r * NOUT + j];
        a[j].x = fmaf(g, x.x, a[j].x); a[j].y = fmaf(g, x.y, a[j].y); a[j].z = fmaf(g, x.z, a[j].z); a[j].w = fmaf(g, x.w, a[j].w);
      }
    }
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
      atomicAdd(&part[j * K + k], a[j].x); atomicAdd(&part[j * K + k + 1], a[j].y);
      atomicAdd(&part[j * K + k + 2], a[j].z); atomicAdd(&part[j * K + k + 3], a[j].w);
    }
  }
  __syncthreads();
  for (int e = t; e < NOUT * K; e += 256) atomicAdd(gW + e, part[e]);
}
inline bool thin_ok(int K, int N) { return TF_GEMM2 && N >= 1 && N <= 4 && K % 4 == 0 && K >= 4 && K <= 1024; }

template <bool A_RF, bool B_RF>
int launch2(const Gemm2Args& G, int splits, bool split, hipStream_t stream, const char* who) {
  dim3 grid((unsigned)((G.M + 127) / 128), (unsigned)((G.N + 127) / 128), (unsigned)splits);
  // split = TF_PREC_BF16X3: the bf16 triple split (round 5: 76.9 -> 89.3 TF/s effective on 236 k x 256 x 256, errors against fp64 the same
  // 3-7e-7 of the largest element as the exact-fp32 instruction's, on activations and on 1e-9-scale gradients alike); TF_PREC_F32 is
  // the v_mfma_f32_32x32x2_f32 form, always (round 6: the caller names the arithmetic; no environment switch)
  if (split) gemm2_kernel<A_RF, B_RF, true><<<grid, 256, 0, stream>>>(G);
  else gemm2_kernel<A_RF, B_RF, false><<<grid, 256, 0, stream>>>(G);
  TF_LAUNCH_CHECK(who);
  return TF_OK;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// 128-wide column blocks against the 64-wide ones of the register-staged kernel: not when the last block would be mostly padding
// (TensoSDF's 129-wide output layer: 256 column slots instead of 192)
inline bool wide_cols_pay(long long n_cols) { return ((n_cols + 127) / 128) * 128 * 4 <= ((n_cols + 63) / 64) * 64 * 5; }

// gZ = gY * act'(Y); gb[j] += sum over rows (one partial per workgroup, atomics)
__global__ void __launch_bounds__(256) act_bwd_kernel(const float* __restrict__ gY, const float* __restrict__ Y, long long n, int N, int act,
                                                      float p, float* __restrict__ gZ, float* __restrict__ gb, const long long* __restrict__ n_dev,
                                                      int slab) {
  if (n_dev) n = min(n, max(0LL, *n_dev));
  // thread = column j (strided), workgroup = a slab of `slab` rows (256 for long matrices; a 2 048-row layer cut into 256-row slabs
  // was 8 workgroups walking 256 dependent rows each: 64 us for 2 MB)
  const long long r0 = (long long)blockIdx.x * slab, r1 = min(r0 + slab, n);
  if (r0 >= n) return;      // launched for a capacity: a slab past the device-side row count adds nothing (and used to add 256 atomic zeros)
  for (int j = threadIdx.x; j < N; j += 256) {
    float s = 0.f;
    for (long long r = r0; r < r1; ++r) {
      const float g = gY[r * N + j] * act_bwd_from_y(Y[r * N + j], act, p);
      gZ[r * N + j] = g;
      s += g;
    }
    if (gb) atomicAdd(gb + j, s);
  }
}
// N < 64 columns: a workgroup owns 4 096 consecutive elements; the bias gradient is summed in LDS and leaves the workgroup as ONE
// atomic per column (one atomic per WAVE and column -- 33 k atomics on three words for 236 k x 3 -- cost 0.4 ms: a single word
// sustains ~88 M atomics/s)
__global__ void __launch_bounds__(256) act_bwd_small_kernel(const float* __restrict__ gY, const float* __restrict__ Y, long long total, int N,
                                                            int act, float p, float* __restrict__ gZ, float* __restrict__ gb,
                                                            const long long* __restrict__ n_dev) {
  if (n_dev) total = min(total, max(0LL, *n_dev) * N);
  __shared__ float sgb[64];
  const long long e0 = (long long)blockIdx.x * 4096;
  if (e0 >= total) return;                      // workgroup-uniform
  if (threadIdx.x < 64) sgb[threadIdx.x] = 0.f;
  __syncthreads();
  if (N <= 4) {
    // the 1- and 3-output heads (every ray of a training step passes through one): a thread keeps one partial sum per column over its
    // 16 elements and the wave is reduced ONCE per column at the end (a shuffle tree per column and ITERATION was 40 us for 2.8 MB)
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const long long e = e0 + it * 256 + threadIdx.x;
      if (e >= total) break;
      const float g = gY[e] * act_bwd_from_y(Y[e], act, p);
      gZ[e] = g;
      const int col = (int)(e % N);
#pragma unroll
      for (int c = 0; c < 4; ++c) part[c] += col == c ? g : 0.f;
    }
    if (gb) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c >= N) break;
        float sc = part[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
        if ((threadIdx.x & 63) == 0 && sc != 0.f) atomicAdd(&sgb[c], sc);
      }
    }
    __syncthreads();
    if (gb && threadIdx.x < N) atomicAdd(gb + threadIdx.x, sgb[threadIdx.x]);
    return;
  }
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const long long e = e0 + it * 256 + threadIdx.x;
    if ((e & ~63LL) >= total) break;              // the whole wave is past the valid elements
    float g = 0.f;
    if (e < total) {
      g = gY[e] * act_bwd_from_y(Y[e], act, p);
      gZ[e] = g;
    }
    if (gb) {
      const int col = (int)(e % N);               // lanes of a wave cycle through the N columns: a shuffle tree per column, then LDS
      for (int c = 0; c < N; ++c) {
        float sc = (col == c && e < total) ? g : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
        if ((threadIdx.x & 63) == 0 && sc != 0.f) atomicAdd(&sgb[c], sc);
      }
    }
  }
  __syncthreads();
  if (gb && threadIdx.x < N) atomicAdd(gb + threadIdx.x, sgb[threadIdx.x]);
}
}  // namespace

// The two backward products of a dense layer given gZ = d loss / d (pre-activation) [n,N]: gX [n,K] = gZ . W (overwritten) and
// gW [N,K] += gZ^T . X (ATOMIC accumulation into whatever gW holds: tf_linear_bwd zeroes it first; tf_sdf_alpha_bwd accumulates over
// sample chunks).  Internal to the library (tf_internal.h): no activation pass, no bias gradient.
// xact != TF_ACT_NONE (tf_linear_bwd_fused): gX receives (gZ . W) * xact'(X) -- the gradient wrt the PRE-activation of the layer below,
// whose post-activation output X is -- and gbx [K] (or null) accumulates its column sums; shapes that neither the aligned matrix-core
// kernel nor the thin kernels take run the plain product followed by the activation pass.
static int linear_products_x(const float* X, const float* W, const float* gZ, long long n, int K, int N, int precision, float* gX, float* gW,
                             const long long* n_dev, hipStream_t stream, int xact, float xact_param, float* gbx);
int tf_linear_products(const float* X, const float* W, const float* gZ, long long n, int K, int N, int precision, float* gX, float* gW,
                       const long long* n_dev, hipStream_t stream) {
  return linear_products_x(X, W, gZ, n, K, N, precision, gX, gW, n_dev, stream, TF_ACT_NONE, 0.f, nullptr);
}
static int linear_products_x(const float* X, const float* W, const float* gZ, long long n, int K, int N, int precision, float* gX, float* gW,
                             const long long* n_dev, hipStream_t stream, int xact, float xact_param, float* gbx) {
  const bool h3 = precision == TF_PREC_F16X3, b3 = precision == TF_PREC_BF16X3;
  const bool fuse = (xact != TF_ACT_NONE || gbx) && gX;      // (a layer below without an activation still gets its bias gradient here)
  if (n == 0) return TF_OK;
  if (!h3 && thin_ok(K, N) && aligned16(X) && aligned16(W) && (!gX || aligned16(gX)) && (!gW || aligned16(gW))) {
    const long long* nd = n_dev;
#define TF_THIN(NO)                                                                                                                  \
    do {                                                                                                                              \
      if (gX && fuse) thin_data_mask_kernel<NO><<<tf_blocks(n, 1024), 256, 0, stream>>>(gZ, W, X, n, K, xact, xact_param, gX, gbx, nd); \
      else if (gX) thin_data_kernel<NO><<<tf_blocks(n * (K / 4), 256), 256, 0, stream>>>(gZ, W, n, K, gX, nd);                        \
      if (gW) thin_weight_kernel<NO><<<tf_blocks(n, 1024), 256, 0, stream>>>(gZ, X, n, K, gW, nd);                                     \
    } while (0)
    if (N == 1) TF_THIN(1); else if (N == 2) TF_THIN(2); else if (N == 3) TF_THIN(3); else TF_THIN(4);
#undef TF_THIN
    TF_LAUNCH_CHECK("tf_linear_bwd(thin)");
    return TF_OK;
  }
  const bool g2 = !h3 && TF_GEMM2 && K % 4 == 0 && N % 4 == 0 && N >= 32 && K >= 32 && wide_cols_pay(N) && wide_cols_pay(K) &&
                  aligned16(X) && aligned16(W) && aligned16(gZ);
  if (gX) {   // gX [n,K] = gZ [n,N] . W [N,K]
    int rc;
    bool fused_here = false;
    if (g2) {
      Gemm2Args G2{gZ, N, W, K, gX, K, nullptr, n, K, N, N, TF_ACT_NONE, 0.f, 0, n_dev, 1};
      if (fuse) { G2.mulY = X; G2.mul_act = xact; G2.mul_param = xact_param; G2.colsum = gbx; fused_here = true; }
      rc = launch2<true, false>(G2, 1, b3, stream, "tf_linear_bwd(data)");
    } else {
      GemmArgs G{gZ, N, 1, W, K, 1, gX, K, nullptr, n, K, N, N, TF_ACT_NONE, 0.f, 0, n_dev, 1};
      rc = launch<true, true>(G, 1, h3, stream, "tf_linear_bwd(data)");
    }
    if (rc != TF_OK) return rc;
    if (fuse && !fused_here) {      // the activation pass of the layer below, in place
      int slab = 256;
      while (slab > 8 && n / slab < 1024) slab >>= 1;
      if (K >= 64) act_bwd_kernel<<<tf_blocks(n, slab), 256, 0, stream>>>(gX, X, n, K, xact, xact_param, gX, gbx, n_dev, slab);
      else act_bwd_small_kernel<<<tf_blocks(n * K, 4096), 256, 0, stream>>>(gX, X, n * K, K, xact, xact_param, gX, gbx, n_dev);
      TF_LAUNCH_CHECK("tf_linear_bwd_fused(act below)");
    }
  }
  if (gW) {   // gW [N,K] = gZ^T . X : A(m = unit, k = row) = gZ[row N + unit], B(k = row, n = k) = X[row K + n]
#ifndef TF_WGRAD_SPLIT
#define TF_WGRAD_SPLIT 1024
#endif
    // rows per workgroup: 1 024 for long matrices (enough workgroups to fill the chip at n ~ 2e5; larger slabs measured no faster);
    // a 2 048-row layer cut that way was two slabs of 64 dependent chunks each -- 85 us for a 2 MB product: short matrices get
    // slabs of n / 256 rows (>= 64).  Each workgroup adds its tile atomically.
    long long split = TF_WGRAD_SPLIT;
    if (n / 256 < split) split = ((n / 256 + 15) / 16) * 16 < 64 ? 64 : ((n / 256 + 15) / 16) * 16;
    const int splits = (int)((n + split - 1) / split);
    int rc;
    if (g2) {
      Gemm2Args G2{gZ, N, X, K, gW, K, nullptr, N, K, n, split, TF_ACT_NONE, 0.f, 1, n_dev, 0};
      rc = launch2<false, false>(G2, splits, b3, stream, "tf_linear_bwd(weight)");
    } else {
      GemmArgs G{gZ, 1, N, X, K, 1, gW, K, nullptr, N, K, n, split, TF_ACT_NONE, 0.f, 1, n_dev, 0};
      rc = launch<false, true>(G, splits, h3, stream, "tf_linear_bwd(weight)");
    }
    if (rc != TF_OK) return rc;
  }
  return TF_OK;
}

extern "C" int tf_linear_fwd(const float* X, const float* W, const float* b, int64_t n, int32_t K, int32_t N, int32_t act, float act_param,
                             int32_t precision, float* Y, const int64_t* n_dev, tf_stream_t stream) {
  TF_REQUIRE(n >= 0 && K > 0 && N > 0, TF_ESHAPE, "tf_linear_fwd: bad sizes n=%lld K=%d N=%d", (long long)n, K, N);
  TF_REQUIRE(act >= TF_ACT_NONE && act <= TF_ACT_EXP_CLAMP, TF_EINVAL, "tf_linear_fwd: unknown activation %d", act);
  TF_REQUIRE(precision == TF_PREC_F32 || precision == TF_PREC_F16X3 || precision == TF_PREC_BF16X3, TF_EINVAL, "tf_linear_fwd: precision %d (TF_PREC_F32, TF_PREC_BF16X3 or TF_PREC_F16X3)", precision);
  const bool h3 = precision == TF_PREC_F16X3;
  if (n == 0) return TF_OK;
  TF_REQUIRE(X && W && Y, TF_EINVAL, "tf_linear_fwd: null pointer");
  if (!h3 && thin_ok(K, N) && aligned16(X) && aligned16(W)) {
    const unsigned blocks = (unsigned)min((long long)tf_blocks(n, 4), 8192LL);
    const long long* nd = (const long long*)n_dev;
    hipStream_t st = (hipStream_t)stream;
    if (N == 1) thin_fwd_kernel<1><<<blocks, 256, 0, st>>>(X, W, b, n, K, act, act_param, Y, nd);
    else if (N == 2) thin_fwd_kernel<2><<<blocks, 256, 0, st>>>(X, W, b, n, K, act, act_param, Y, nd);
    else if (N == 3) thin_fwd_kernel<3><<<blocks, 256, 0, st>>>(X, W, b, n, K, act, act_param, Y, nd);
    else thin_fwd_kernel<4><<<blocks, 256, 0, st>>>(X, W, b, n, K, act, act_param, Y, nd);
    TF_LAUNCH_CHECK("tf_linear_fwd(thin)");
    return TF_OK;
  }
  if (!h3 && TF_GEMM2 && K % 4 == 0 && N >= 32 && wide_cols_pay(N) && aligned16(X) && aligned16(W)) {
    Gemm2Args G2{X, K, W, K, Y, N, b, n, N, K, K, act, act_param, 0, (const long long*)n_dev, 1};
    return launch2<true, true>(G2, 1, precision == TF_PREC_BF16X3, (hipStream_t)stream, "tf_linear_fwd");
  }
  GemmArgs G{X, K, 1, W, 1, K, Y, N, b, n, N, K, K, act, act_param, 0, (const long long*)n_dev, 1};
  return launch<true, false>(G, 1, h3, (hipStream_t)stream, "tf_linear_fwd");
}

extern "C" int tf_linear_bwd(const float* X, const float* W, const float* Y, const float* gY, int64_t n, int32_t K, int32_t N, int32_t act,
                             float act_param, int32_t precision, float* gZ, float* gX, float* gW, float* gb, const int64_t* n_dev,
                             tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(precision == TF_PREC_F32 || precision == TF_PREC_F16X3 || precision == TF_PREC_BF16X3, TF_EINVAL, "tf_linear_bwd: precision %d (TF_PREC_F32, TF_PREC_BF16X3 or TF_PREC_F16X3)", precision);
  const bool h3 = precision == TF_PREC_F16X3;
  TF_REQUIRE(n >= 0 && K > 0 && N > 0, TF_ESHAPE, "tf_linear_bwd: bad sizes n=%lld K=%d N=%d", (long long)n, K, N);
  TF_REQUIRE(act >= TF_ACT_NONE && act <= TF_ACT_EXP_CLAMP, TF_EINVAL, "tf_linear_bwd: unknown activation %d", act);
  TF_REQUIRE(W && (n == 0 || (X && Y && gY && gZ)), TF_EINVAL, "tf_linear_bwd: null pointer");
  if (gW) { hipError_t e = hipMemsetAsync(gW, 0, sizeof(float) * (size_t)N * K, stream); TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_linear_bwd: memset failed"); }
  if (gb) { hipError_t e = hipMemsetAsync(gb, 0, sizeof(float) * (size_t)N, stream); TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_linear_bwd: memset failed"); }
  if (n == 0) return TF_OK;
  if (N >= 64) {
    int slab = 256;
    while (slab > 8 && n / slab < 1024) slab >>= 1;       // >= 1 024 workgroups where the matrix has the rows for it
    act_bwd_kernel<<<tf_blocks(n, slab), 256, 0, stream>>>(gY, Y, n, N, act, act_param, gZ, gb, (const long long*)n_dev, slab);
  }
  else act_bwd_small_kernel<<<tf_blocks(n * N, 4096), 256, 0, stream>>>(gY, Y, n * N, N, act, act_param, gZ, gb, (const long long*)n_dev);
  TF_LAUNCH_CHECK("tf_linear_bwd(act)");
  return tf_linear_products(X, W, gZ, n, K, N, precision, gX, gW, (const long long*)n_dev, stream);
}

// One layer of a backward CHAIN through stacked dense layers (LightsFn.backward: the inner-light net's four layers): as tf_linear_bwd,
// and in the same launches the activation backward of the layer BELOW -- gX then holds the gradient wrt that layer's pre-activation and
// gbx its bias gradient, so the next call of the chain passes them on with gy_is_gz = 1 and runs no activation pass of its own.
extern "C" int tf_linear_bwd_fused(const float* X, const float* W, const float* Y, const float* gY, int64_t n, int32_t K, int32_t N, int32_t act,
                                   float act_param, int32_t gy_is_gz, int32_t x_act, float x_act_param, int32_t precision, float* gZ,
                                   float* gX, float* gW, float* gb, float* gbx, const int64_t* n_dev, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(precision == TF_PREC_F32 || precision == TF_PREC_F16X3 || precision == TF_PREC_BF16X3, TF_EINVAL, "tf_linear_bwd_fused: precision %d (TF_PREC_F32, TF_PREC_BF16X3 or TF_PREC_F16X3)", precision);
  TF_REQUIRE(n >= 0 && K > 0 && N > 0, TF_ESHAPE, "tf_linear_bwd_fused: bad sizes n=%lld K=%d N=%d", (long long)n, K, N);
  TF_REQUIRE(act >= TF_ACT_NONE && act <= TF_ACT_EXP_CLAMP && x_act >= TF_ACT_NONE && x_act <= TF_ACT_EXP_CLAMP, TF_EINVAL,
             "tf_linear_bwd_fused: unknown activation %d / %d", act, x_act);
  TF_REQUIRE(W && (n == 0 || (X && gY && (gy_is_gz || (Y && gZ)))), TF_EINVAL, "tf_linear_bwd_fused: null pointer");
  const bool zeroed = (gy_is_gz & TF_BWD_GRADS_ZEROED) != 0;      // the caller filled gW / gb / gbx with zeros (one fill for a whole chain)
  gy_is_gz &= 1;
  TF_REQUIRE(!gy_is_gz || !gb, TF_EINVAL, "tf_linear_bwd_fused: with gy_is_gz the bias gradient came out of the previous call (gbx)");
  TF_REQUIRE(!gbx || gX, TF_EINVAL, "tf_linear_bwd_fused: gbx needs gX");
  if (gW && !zeroed) { hipError_t e = hipMemsetAsync(gW, 0, sizeof(float) * (size_t)N * K, stream); TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_linear_bwd_fused: memset failed"); }
  if (gb && !zeroed) { hipError_t e = hipMemsetAsync(gb, 0, sizeof(float) * (size_t)N, stream); TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_linear_bwd_fused: memset failed"); }
  if (gbx && !zeroed) { hipError_t e = hipMemsetAsync(gbx, 0, sizeof(float) * (size_t)K, stream); TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_linear_bwd_fused: memset failed"); }
  if (n == 0) return TF_OK;
  const float* gz = gY;
  if (!gy_is_gz) {
    if (N >= 64) {
      int slab = 256;
      while (slab > 8 && n / slab < 1024) slab >>= 1;
      act_bwd_kernel<<<tf_blocks(n, slab), 256, 0, stream>>>(gY, Y, n, N, act, act_param, gZ, gb, (const long long*)n_dev, slab);
    }
    else act_bwd_small_kernel<<<tf_blocks(n * N, 4096), 256, 0, stream>>>(gY, Y, n * N, N, act, act_param, gZ, gb, (const long long*)n_dev);
    TF_LAUNCH_CHECK("tf_linear_bwd_fused(act)");
    gz = gZ;
  }
  return linear_products_x(X, W, gz, n, K, N, precision, gX, gW, (const long long*)n_dev, stream, gX ? x_act : (int)TF_ACT_NONE, x_act_param, gbx);
}

