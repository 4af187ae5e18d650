// TVLoss (network/other_field.py:170-191) of a [C,H,W] grid -- the parameter-only regulariser of the SDF planes and lines
// (TensoSDF.TV_loss_sdf, fields.py:133-138; shapeRenderer.py:1222-1223) -- as one forward and one backward kernel per grid instead of
// ~12 element-wise launches forward and autograd's slice backward (a zero-filled plane-sized tensor + copy + add per slice) on six
// grids per training step.
//   forward : partial[b] = (sum of squared differences along H, along W) of block b's elements, fixed grid of blocks: the caller sums
//             the partials in a fixed order (deterministic, no float atomics)
//   backward: g_x[c,i,j] = ch * 2 ((x[i,j] - x[i-1,j]) - (x[i+1,j] - x[i,j])) + cw * 2 (same along W), ch / cw = upstream gradient x
//             weight factors, read from device memory (no host sync)
#include "tf_common.h"

namespace {
constexpr int kTvBlocks = 1024;

__global__ void __launch_bounds__(256) tv_fwd_kernel(const float* __restrict__ x, long long total, int H, int W, float* __restrict__ partial) {
  float sh = 0.f, sw = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % W), i = (int)((e / W) % H);
    const float v = x[e];
    if (i + 1 < H) { const float d = x[e + W] - v; sh += d * d; }
    if (j + 1 < W) { const float d = x[e + 1] - v; sw += d * d; }
  }
  __shared__ float red[2][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sh += __shfl_xor(sh, o); sw += __shfl_xor(sw, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sh; red[1][threadIdx.x >> 6] = sw; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ void __launch_bounds__(256) tv_bwd_kernel(const float* __restrict__ x, long long total, int H, int W, const float* __restrict__ g,
                                                     float ch, float cw, float* __restrict__ gx) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % W), i = (int)((e / W) % H);
  const float v = x[e], up = g[0];
  float a = 0.f, b = 0.f;
  if (i > 0) a += v - x[e - W];
  if (i + 1 < H) a -= x[e + W] - v;
  if (j > 0) b += v - x[e - 1];
  if (j + 1 < W) b -= x[e + 1] - v;
  gx[e] = up * (2.f * ch * a + 2.f * cw * b);
}
// loss[0] += coef_h * sum of the H partials + coef_w * sum of the W partials, in a fixed order (one workgroup): the six grids of
// TensoSDF.TV_loss_sdf accumulate into one scalar without an element-wise launch between them (round 5)
__global__ void __launch_bounds__(256) tv_finish_kernel(const float* __restrict__ partial, float ch, float cw, float* __restrict__ loss) {
  float sh = 0.f, sw = 0.f;
#pragma unroll
  for (int k = 0; k < kTvBlocks / 256; ++k) {
    sh += partial[2 * (k * 256 + threadIdx.x)];
    sw += partial[2 * (k * 256 + threadIdx.x) + 1];
  }
  __shared__ float red[2][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sh += __shfl_xor(sh, o); sw += __shfl_xor(sw, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sh; red[1][threadIdx.x >> 6] = sw; }
  __syncthreads();
  if (threadIdx.x == 0)
    loss[0] += ch * ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) + cw * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
}
}  // namespace

extern "C" int32_t tf_tv_partials(void) { return 2 * kTvBlocks; }

extern "C" int tf_tv_finish(const float* partial, float coef_h, float coef_w, float* loss, tf_stream_t stream) {
  TF_REQUIRE(partial && loss, TF_EINVAL, "tf_tv_finish: null pointer");
  tv_finish_kernel<<<1, 256, 0, (hipStream_t)stream>>>(partial, coef_h, coef_w, loss);
  TF_LAUNCH_CHECK("tf_tv_finish");
  return TF_OK;
}

extern "C" int tf_tv_fwd(const float* x, int32_t C, int32_t H, int32_t W, float* partial, tf_stream_t stream) {
  TF_REQUIRE(C >= 1 && H >= 1 && W >= 1, TF_ESHAPE, "tf_tv_fwd: bad sizes");
  TF_REQUIRE(x && partial, TF_EINVAL, "tf_tv_fwd: null pointer");
  tv_fwd_kernel<<<kTvBlocks, 256, 0, (hipStream_t)stream>>>(x, (long long)C * H * W, H, W, partial);
  TF_LAUNCH_CHECK("tf_tv_fwd");
  return TF_OK;
}

extern "C" int tf_tv_bwd(const float* x, int32_t C, int32_t H, int32_t W, const float* g_dev, float coef_h, float coef_w, float* g_x,
                         tf_stream_t stream) {
  TF_REQUIRE(C >= 1 && H >= 1 && W >= 1, TF_ESHAPE, "tf_tv_bwd: bad sizes");
  TF_REQUIRE(x && g_dev && g_x, TF_EINVAL, "tf_tv_bwd: null pointer");
  const long long total = (long long)C * H * W;
  tv_bwd_kernel<<<tf_blocks(total, 256), 256, 0, (hipStream_t)stream>>>(x, total, H, W, g_dev, coef_h, coef_w, g_x);
  TF_LAUNCH_CHECK("tf_tv_bwd");
  return TF_OK;
}
