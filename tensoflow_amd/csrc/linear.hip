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
#include "tf_common.h"

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

// gZ = gY * act'(Y); gb[j] += sum over rows (one partial per workgroup, atomics)
__global__ void __launch_bounds__(256) act_bwd_kernel(const float* __restrict__ gY, const float* __restrict__ Y, long long n, int N, int act,
                                                      float p, float* __restrict__ gZ, float* __restrict__ gb, const long long* __restrict__ n_dev) {
  if (n_dev) n = min(n, max(0LL, *n_dev));
  // thread = column j (strided), workgroup = a slab of 256 rows
  const long long r0 = (long long)blockIdx.x * 256, r1 = min(r0 + 256, n);
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
// N < 64 columns: one thread per element; the bias gradient is reduced over the wave before the atomic (one atomic per wave and
// column instead of one per element: with N = 1 .. 3 every element of the matrix hit the same few words)
__global__ void __launch_bounds__(256) act_bwd_small_kernel(const float* __restrict__ gY, const float* __restrict__ Y, long long total, int N,
                                                            int act, float p, float* __restrict__ gZ, float* __restrict__ gb,
                                                            const long long* __restrict__ n_dev) {
  if (n_dev) total = min(total, max(0LL, *n_dev) * N);
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if ((e & ~63LL) >= total) return;             // the whole wave is past the valid elements: nothing to store, nothing to reduce
  float g = 0.f;
  if (e < total) {
    g = gY[e] * act_bwd_from_y(Y[e], act, p);
    gZ[e] = g;
  }
  if (!gb) return;
  const int col = (int)(e % N);                 // lanes of a wave cycle through the N columns
  for (int c = 0; c < N; ++c) {
    float s = (col == c && e < total) ? g : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0 && s != 0.f) atomicAdd(gb + c, s);
  }
}
}  // namespace

extern "C" int tf_linear_fwd(const float* X, const float* W, const float* b, int64_t n, int32_t K, int32_t N, int32_t act, float act_param,
                             int32_t precision, float* Y, const int64_t* n_dev, tf_stream_t stream) {
  TF_REQUIRE(n >= 0 && K > 0 && N > 0, TF_ESHAPE, "tf_linear_fwd: bad sizes n=%lld K=%d N=%d", (long long)n, K, N);
  TF_REQUIRE(act >= TF_ACT_NONE && act <= TF_ACT_EXP_CLAMP, TF_EINVAL, "tf_linear_fwd: unknown activation %d", act);
  TF_REQUIRE(precision == TF_PREC_F32 || precision == TF_PREC_F16X3, TF_EINVAL, "tf_linear_fwd: precision %d (TF_PREC_F32 or TF_PREC_F16X3)", precision);
  const bool h3 = precision == TF_PREC_F16X3;
  if (n == 0) return TF_OK;
  TF_REQUIRE(X && W && Y, TF_EINVAL, "tf_linear_fwd: null pointer");
  GemmArgs G{X, K, 1, W, 1, K, Y, N, b, n, N, K, K, act, act_param, 0, (const long long*)n_dev, 1};
  return launch<true, false>(G, 1, h3, (hipStream_t)stream, "tf_linear_fwd");
}

extern "C" int tf_linear_bwd(const float* X, const float* W, const float* Y, const float* gY, int64_t n, int32_t K, int32_t N, int32_t act,
                             float act_param, int32_t precision, float* gZ, float* gX, float* gW, float* gb, const int64_t* n_dev,
                             tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(precision == TF_PREC_F32 || precision == TF_PREC_F16X3, TF_EINVAL, "tf_linear_bwd: precision %d (TF_PREC_F32 or TF_PREC_F16X3)", precision);
  const bool h3 = precision == TF_PREC_F16X3;
  TF_REQUIRE(n >= 0 && K > 0 && N > 0, TF_ESHAPE, "tf_linear_bwd: bad sizes n=%lld K=%d N=%d", (long long)n, K, N);
  TF_REQUIRE(act >= TF_ACT_NONE && act <= TF_ACT_EXP_CLAMP, TF_EINVAL, "tf_linear_bwd: unknown activation %d", act);
  TF_REQUIRE(W && (n == 0 || (X && Y && gY && gZ)), TF_EINVAL, "tf_linear_bwd: null pointer");
  if (gW) { hipError_t e = hipMemsetAsync(gW, 0, sizeof(float) * (size_t)N * K, stream); TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_linear_bwd: memset failed"); }
  if (gb) { hipError_t e = hipMemsetAsync(gb, 0, sizeof(float) * (size_t)N, stream); TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_linear_bwd: memset failed"); }
  if (n == 0) return TF_OK;
  if (N >= 64) act_bwd_kernel<<<tf_blocks(n, 256), 256, 0, stream>>>(gY, Y, n, N, act, act_param, gZ, gb, (const long long*)n_dev);
  else act_bwd_small_kernel<<<tf_blocks(n * N, 256), 256, 0, stream>>>(gY, Y, n * N, N, act, act_param, gZ, gb, (const long long*)n_dev);
  TF_LAUNCH_CHECK("tf_linear_bwd(act)");
  if (gX) {   // gX [n,K] = gZ [n,N] . W [N,K]
    GemmArgs G{gZ, N, 1, W, K, 1, gX, K, nullptr, n, K, N, N, TF_ACT_NONE, 0.f, 0, (const long long*)n_dev, 1};
    const int rc = launch<true, true>(G, 1, h3, stream, "tf_linear_bwd(data)");
    if (rc != TF_OK) return rc;
  }
  if (gW) {   // gW [N,K] = gZ^T . X : A(m = unit, k = row) = gZ[row N + unit], B(k = row, n = k) = X[row K + n]
    const long long split = 1024;     // rows per workgroup: enough workgroups to fill the chip at n ~ 2e5; each adds a [128 x 64] tile atomically
    const int splits = (int)((n + split - 1) / split);
    GemmArgs G{gZ, 1, N, X, K, 1, gW, K, nullptr, N, K, n, split, TF_ACT_NONE, 0.f, 1, (const long long*)n_dev, 0};
    const int rc = launch<false, true>(G, splits, h3, stream, "tf_linear_bwd(weight)");
    if (rc != TF_OK) return rc;
  }
  return TF_OK;
}
