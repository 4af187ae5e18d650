// Encodings of the training direction as kernels with analytic backward (round 4).
//   tf_ide5_fwd / tf_ide5_bwd : generate_ide_fn(5) (utils/ref_utils.py:53-117): the integrated directional encoding, 72 values per row
//                               [Re(36) | Im(36)] of (x + i y)^m P_{l,m}(z) exp(-l (l + 1) / 2 * kappa_inv), l = 1, 2, 4, 8, 16, m = 0..l,
//                               and its gradient wrt the direction and kappa_inv.
//   tf_posenc_fwd             : get_embedder (utils/network_utils.py:38-50): [x, sin(2^k x), cos(2^k x)]_k.
// The torch composition of rounds 1-3 (16 pow launches, 64 element-wise launches of the complex recurrence, a dense layer for the
// polynomials, and autograd's mirror image of all of it) was ~350 launches per shape-stage training step.  One lane = one row; the
// polynomials run in fp64 Horner form on the fp32-ROUNDED coefficient table (the table is part of the reference function; its degree-16
// columns cancel catastrophically in fp32: the reference's own fp32 values are good to ~6e-4 -- this evaluation sits inside that noise).
#include "tf_common.h"

namespace {
constexpr int kIdeCols = 36;
__device__ __forceinline__ void ide_col(int c, int& l, int& m) {      // columns ordered (l = 1: m = 0, 1), (2: 0..2), (4: 0..4), (8: 0..8), (16: 0..16)
  if (c < 2) { l = 1; m = c; }
  else if (c < 5) { l = 2; m = c - 2; }
  else if (c < 10) { l = 4; m = c - 5; }
  else if (c < 19) { l = 8; m = c - 10; }
  else { l = 16; m = c - 19; }
}

template <bool BWD>
__global__ void __launch_bounds__(256) ide5_kernel(const float* __restrict__ xyz, const float* __restrict__ kappa, const float* __restrict__ mat /* [17,36] */,
                                                   long long n, float* __restrict__ out /* [n,72] */, const float* __restrict__ g_out,
                                                   float* __restrict__ g_xyz, float* __restrict__ g_kappa) {
  __shared__ float tab[17 * kIdeCols];
  for (int i = threadIdx.x; i < 17 * kIdeCols; i += 256) tab[i] = mat[i];
  __syncthreads();
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const double x = xyz[3 * r], y = xyz[3 * r + 1], z = xyz[3 * r + 2];
  const double kap = kappa ? (double)kappa[r] : 0.0;
  double re[17], im[17];
  re[0] = 1.0; im[0] = 0.0;
#pragma unroll
  for (int k = 1; k < 17; ++k) { re[k] = re[k - 1] * x - im[k - 1] * y; im[k] = re[k - 1] * y + im[k - 1] * x; }
  double gx = 0.0, gy = 0.0, gz = 0.0, gk = 0.0;
#pragma unroll 1
  for (int c = 0; c < kIdeCols; ++c) {
    int l, m;
    ide_col(c, l, m);
    const int deg = l - m;                              // mat[k, c] = 0 for k > l - m
    double p = (double)tab[deg * kIdeCols + c], dp = 0.0;
    for (int k = deg - 1; k >= 0; --k) { dp = dp * z + p; p = p * z + (double)tab[k * kIdeCols + c]; }
    const double att = exp(-0.5 * l * (l + 1) * kap);
    // dynamic index into re / im: a select chain over the five possible ... no: m ranges 0..16 -> read through a small switch-free loop
    double rm = re[0], imm = im[0], rm1 = 0.0, im1 = 0.0;
#pragma unroll
    for (int k = 1; k < 17; ++k) { const bool s = k == m, s1 = k == m - 1; rm = s ? re[k] : rm; imm = s ? im[k] : imm; rm1 = s1 ? re[k] : rm1; im1 = s1 ? im[k] : im1; }
    if (m == 1) { rm1 = 1.0; im1 = 0.0; }
    if (!BWD) {
      out[r * 72 + c] = (float)(rm * p * att);
      out[r * 72 + 36 + c] = (float)(imm * p * att);
    } else {
      const double gr = g_out[r * 72 + c], gi = g_out[r * 72 + 36 + c];
      const double pa = p * att;
      // d (x + i y)^m / dx = m (x + i y)^(m-1),  d / dy = i m (x + i y)^(m-1)
      gx += (gr * rm1 + gi * im1) * (m * pa);
      gy += (-gr * im1 + gi * rm1) * (m * pa);
      gz += (gr * rm + gi * imm) * (dp * att);
      gk += (gr * rm + gi * imm) * (pa * (-0.5 * l * (l + 1)));
    }
  }
  if (BWD) {
    g_xyz[3 * r] = (float)gx; g_xyz[3 * r + 1] = (float)gy; g_xyz[3 * r + 2] = (float)gz;
    if (g_kappa) g_kappa[r] = (float)gk;
  }
}

__global__ void __launch_bounds__(256) posenc_kernel(const float* __restrict__ x, long long total /* n * d */, int d, int n_freq,
                                                     float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long long r = e / d;
  const int a = (int)(e - r * d);
  const int ld = d * (1 + 2 * n_freq);
  const float v = x[e];
  float* o = out + r * ld;
  o[a] = v;
  float f = 1.f;
  for (int k = 0; k < n_freq; ++k) {
    const float t = v * f;
    o[d * (1 + 2 * k) + a] = sinf(t);
    o[d * (2 + 2 * k) + a] = cosf(t);
    f *= 2.f;
  }
}
}  // namespace

extern "C" int tf_ide5_fwd(const float* xyz, const float* kappa_inv, const float* coef, int64_t n, float* out, tf_stream_t stream) {
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_ide5_fwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(xyz && coef && out, TF_EINVAL, "tf_ide5_fwd: null pointer");
  ide5_kernel<false><<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(xyz, kappa_inv, coef, n, out, nullptr, nullptr, nullptr);
  TF_LAUNCH_CHECK("tf_ide5_fwd");
  return TF_OK;
}

extern "C" int tf_ide5_bwd(const float* xyz, const float* kappa_inv, const float* coef, const float* g_out, int64_t n, float* g_xyz,
                           float* g_kappa, tf_stream_t stream) {
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_ide5_bwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(xyz && coef && g_out && g_xyz, TF_EINVAL, "tf_ide5_bwd: null pointer");
  ide5_kernel<true><<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(xyz, kappa_inv, coef, n, nullptr, g_out, g_xyz, g_kappa);
  TF_LAUNCH_CHECK("tf_ide5_bwd");
  return TF_OK;
}

// linear_to_srgb (utils/raw_utils.py:4-17), optionally clamped to [0, 1] (the `clamp(.., 0, 1)` fields.py wraps around most of its uses),
// and its derivative: one launch each instead of nine element-wise launches forward and about as many backward per call.
__device__ __forceinline__ float srgb_of(float x) {
  const float eps = 1.1920928955078125e-07f;
  return x <= 0.0031308f ? 12.92f * x : (211.f * powf(fmaxf(x, eps), 5.f / 12.f) - 11.f) / 200.f;
}
__global__ void __launch_bounds__(256) srgb_kernel(const float* __restrict__ lin, const float* __restrict__ g_out, long long n, int clamp01,
                                                   float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = lin[i];
  const float y = srgb_of(x);
  if (!g_out) { out[i] = clamp01 ? fminf(fmaxf(y, 0.f), 1.f) : y; return; }
  const float eps = 1.1920928955078125e-07f;
  // the branch taken, as autograd differentiates torch.where / clamp(min=eps) / pow
  float d = x <= 0.0031308f ? 12.92f : (x >= eps ? (211.f / 200.f) * (5.f / 12.f) * powf(x, -7.f / 12.f) : 0.f);
  if (clamp01 && !(y >= 0.f && y <= 1.f)) d = 0.f;
  out[i] = g_out[i] * d;
}
extern "C" int tf_linear_to_srgb_fwd(const float* lin, int64_t n, int32_t clamp01, float* out, tf_stream_t stream) {
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_linear_to_srgb_fwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(lin && out, TF_EINVAL, "tf_linear_to_srgb_fwd: null pointer");
  srgb_kernel<<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(lin, nullptr, n, clamp01, out);
  TF_LAUNCH_CHECK("tf_linear_to_srgb_fwd");
  return TF_OK;
}
extern "C" int tf_linear_to_srgb_bwd(const float* lin, const float* g_out, int64_t n, int32_t clamp01, float* g_lin, tf_stream_t stream) {
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_linear_to_srgb_bwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(lin && g_out && g_lin, TF_EINVAL, "tf_linear_to_srgb_bwd: null pointer");
  srgb_kernel<<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(lin, g_out, n, clamp01, g_lin);
  TF_LAUNCH_CHECK("tf_linear_to_srgb_bwd");
  return TF_OK;
}

extern "C" int tf_posenc_fwd(const float* x, int64_t n, int32_t d, int32_t n_freq, float* out, tf_stream_t stream) {
  TF_REQUIRE(n >= 0 && d >= 1 && n_freq >= 0 && n_freq <= 16, TF_ESHAPE, "tf_posenc_fwd: bad sizes");
  if (n == 0) return TF_OK;
  TF_REQUIRE(x && out, TF_EINVAL, "tf_posenc_fwd: null pointer");
  posenc_kernel<<<tf_blocks(n * d, 256), 256, 0, (hipStream_t)stream>>>(x, n * d, d, n_freq, out);
  TF_LAUNCH_CHECK("tf_posenc_fwd");
  return TF_OK;
}
