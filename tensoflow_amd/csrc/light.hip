// Environment light lookup: EnvLight.direct_light (network/light.py:125-162) =
// exp(dr.texture(base[None], dirs, filter_mode='linear', boundary_mode='cube')).
// The cube map (6*R*R*3 floats, 1.2 MB at R=128) is L2-resident; one lane per direction,
// 4 taps x 3 channels; taps that leave the face are re-projected onto the neighbouring face,
// the tap that falls off a cube corner is dropped and the other three renormalised
// (same rule as oracle/texture.py:cube_bilinear).
#include "cube.h"
#include "tf_common.h"

__global__ void __launch_bounds__(256) cube_lookup_fwd_kernel(const float* __restrict__ base, int R,
                                                              const float* __restrict__ dirs, long long m, int apply_exp,
                                                              const float* __restrict__ depth, float near_eps,
                                                              float* __restrict__ out) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  CubeTaps T;
  cube_taps(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], R, T);
  float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float* p = base + 3LL * T.idx[t];
    r += T.w[t] * p[0]; g += T.w[t] * p[1]; b += T.w[t] * p[2];
  }
  if (apply_exp) { r = expf(r); g = expf(g); b = expf(b); }
  if (depth && !(depth[i] > near_eps)) { r = 0.f; g = 0.f; b = 0.f; }   // fields.py:973-974 near mask
  out[3 * i] = r; out[3 * i + 1] = g; out[3 * i + 2] = b;
}

__global__ void __launch_bounds__(256) cube_lookup_bwd_kernel(const float* __restrict__ base, int R,
                                                              const float* __restrict__ dirs, long long m, int apply_exp,
                                                              const float* __restrict__ g_out, float* __restrict__ g_base) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  CubeTaps T;
  cube_taps(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], R, T);
  float gr = g_out[3 * i], gg = g_out[3 * i + 1], gb = g_out[3 * i + 2];
  if (apply_exp) {
    float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* p = base + 3LL * T.idx[t];
      r += T.w[t] * p[0]; g += T.w[t] * p[1]; b += T.w[t] * p[2];
    }
    gr *= expf(r); gg *= expf(g); gb *= expf(b);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (T.w[t] == 0.f) continue;
    float* p = g_base + 3LL * T.idx[t];
    atomicAdd(p, T.w[t] * gr); atomicAdd(p + 1, T.w[t] * gg); atomicAdd(p + 2, T.w[t] * gb);
  }
}

extern "C" int tf_cube_lookup_fwd(const float* base, int32_t res, const float* dirs, int64_t m, int32_t apply_exp,
                                  const float* depth, float near_eps, float* out, tf_stream_t stream) {
  TF_REQUIRE(m >= 0 && res > 0, TF_ESHAPE, "tf_cube_lookup_fwd: m < 0 or res <= 0");
  if (m == 0) return TF_OK;
  TF_REQUIRE(base && dirs && out, TF_EINVAL, "tf_cube_lookup_fwd: null pointer");
  cube_lookup_fwd_kernel<<<tf_blocks(m, 256), 256, 0, (hipStream_t)stream>>>(base, res, dirs, m, apply_exp, depth, near_eps, out);
  TF_LAUNCH_CHECK("tf_cube_lookup_fwd");
  return TF_OK;
}

extern "C" int tf_cube_lookup_bwd(const float* base, int32_t res, const float* dirs, int64_t m, int32_t apply_exp,
                                  const float* g_out, float* g_base, tf_stream_t stream) {
  TF_REQUIRE(m >= 0 && res > 0, TF_ESHAPE, "tf_cube_lookup_bwd: m < 0 or res <= 0");
  if (m == 0) return TF_OK;
  TF_REQUIRE(base && dirs && g_out && g_base, TF_EINVAL, "tf_cube_lookup_bwd: null pointer");
  cube_lookup_bwd_kernel<<<tf_blocks(m, 256), 256, 0, (hipStream_t)stream>>>(base, res, dirs, m, apply_exp, g_out, g_base);
  TF_LAUNCH_CHECK("tf_cube_lookup_bwd");
  return TF_OK;
}
