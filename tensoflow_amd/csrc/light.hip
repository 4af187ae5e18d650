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

// Backward of the lookup.  g_base (nullable): += d out / d map (float atomics).  g_dirs (nullable): d out / d direction --
// dr.texture is differentiable in its coordinates, and the shape stage reaches the SDF through them (envlight(normal),
// envlight(reflective, roughness), fields.py:436-446): the bilinear weights are differentiated in the face coordinates
// (x, y) of the major face (taps re-projected across a seam keep the weight of the major face; at a cube corner the
// dropped tap renormalises the other three -- quotient rule), then mapped through x = +-d_minor / |d_major|.
__global__ void __launch_bounds__(256) cube_lookup_bwd_kernel(const float* __restrict__ base, int R,
                                                              const float* __restrict__ dirs, long long m, int apply_exp,
                                                              const float* __restrict__ g_out, float* __restrict__ g_base,
                                                              float* __restrict__ g_dirs) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  // a lane past the end stays in the wave (the map scatter below exchanges values across lanes) with a harmless direction and no gradient
  const bool live = i < m;
  if (!live) i = m - 1;
  const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
  CubeTaps T;
  cube_taps(dx, dy, dz, R, T);
  float gr = live ? g_out[3 * i] : 0.f, gg = live ? g_out[3 * i + 1] : 0.f, gb = live ? g_out[3 * i + 2] : 0.f;
  float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float* p = base + 3LL * T.idx[t];
    r += T.w[t] * p[0]; g += T.w[t] * p[1]; b += T.w[t] * p[2];
  }
  if (apply_exp) { gr *= expf(r); gg *= expf(g); gb *= expf(b); }
  if (g_base) {
    // The map (18.9 MB at 512^2) sits behind the L2, where a scatter is bound by atomic REQUESTS: the three channels of a texel are
    // three consecutive words, so three NEIGHBOURING lanes add them in one instruction (one request per texel instead of three).
    // Per tap, four sub-steps of 16 rays: lane 3 r' + c serves channel c of ray 16 s + r' (values fetched across lanes).
    const int lane = threadIdx.x & 63;
    const int grp = lane / 3, c = lane - 3 * grp;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float w0 = T.w[t] * gr, w1 = T.w[t] * gg, w2 = T.w[t] * gb;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int src = (16 * sub + grp) & 63;
        const int idx_s = __shfl(T.idx[t], src);
        const float v0 = __shfl(w0, src), v1 = __shfl(w1, src), v2 = __shfl(w2, src);
        const float val = c == 0 ? v0 : c == 1 ? v1 : v2;
        if (lane < 48 && val != 0.f) atomicAdd(g_base + 3LL * idx_s + c, val);
      }
    }
  }
  if (g_dirs && live) {
    int face;
    float x, y;
    cube_face_uv(dx, dy, dz, face, x, y);
    const float u = (x * 0.5f + 0.5f) * (float)R - 0.5f, v = (y * 0.5f + 0.5f) * (float)R - 0.5f;
    const float fu = u - floorf(u), fv = v - floorf(v);
    // raw (un-normalised) weights and their u / v derivatives; a dropped corner tap has T.w == 0 and raw weight 0
    float wsum = 0.f, su = 0.f, sv = 0.f, au = 0.f, av = 0.f;   // sum w, sum dw/du, sum dw/dv, sum dw/du * (g . T), sum dw/dv * (g . T)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int du = t & 1, dv = t >> 1;
      const bool dropped = T.w[t] == 0.f && ((du ? fu : 1.f - fu) * (dv ? fv : 1.f - fv)) != 0.f;
      if (dropped) continue;
      const float* p = base + 3LL * T.idx[t];
      const float gt = gr * p[0] + gg * p[1] + gb * p[2];
      const float wu = (du ? 1.f : -1.f) * (dv ? fv : 1.f - fv), wv = (du ? fu : 1.f - fu) * (dv ? 1.f : -1.f);
      wsum += (du ? fu : 1.f - fu) * (dv ? fv : 1.f - fv);
      su += wu; sv += wv; au += wu * gt; av += wv * gt;
    }
    const float gval = gr * r + gg * g + gb * b;
    const float inv = 1.f / wsum;
    const float gu = (au - gval * su) * inv, gv = (av - gval * sv) * inv;
    const float gx = gu * 0.5f * (float)R, gy = gv * 0.5f * (float)R;
    float ox, oy, oz;
    if (face >= 4) {            // major z: x = +-dx / |dz|, y = -dy / |dz|
      const float mz = 1.f / fabsf(dz), sg = dz < 0.f ? -1.f : 1.f;
      ox = sg * mz * gx; oy = -mz * gy; oz = -(x * gx + y * gy) * mz * sg;
    } else if (face >= 2) {     // major y: x = dx / |dy|, y = +-dz / |dy|
      const float my = 1.f / fabsf(dy), sg = dy < 0.f ? -1.f : 1.f;
      ox = my * gx; oz = sg * my * gy; oy = -(x * gx + y * gy) * my * sg;
    } else {                    // major x: x = -+dz / |dx|, y = -dy / |dx|
      const float mx = 1.f / fabsf(dx), sg = dx < 0.f ? -1.f : 1.f;
      oz = -sg * mx * gx; oy = -mx * gy; ox = -(x * gx + y * gy) * mx * sg;
    }
    g_dirs[3 * i] = ox; g_dirs[3 * i + 1] = oy; g_dirs[3 * i + 2] = oz;
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
  cube_lookup_bwd_kernel<<<tf_blocks(m, 256), 256, 0, (hipStream_t)stream>>>(base, res, dirs, m, apply_exp, g_out, g_base, nullptr);
  TF_LAUNCH_CHECK("tf_cube_lookup_bwd");
  return TF_OK;
}

extern "C" int tf_cube_lookup_bwd_dirs(const float* base, int32_t res, const float* dirs, int64_t m, int32_t apply_exp,
                                       const float* g_out, float* g_base, float* g_dirs, tf_stream_t stream) {
  TF_REQUIRE(m >= 0 && res > 0, TF_ESHAPE, "tf_cube_lookup_bwd_dirs: m < 0 or res <= 0");
  if (m == 0) return TF_OK;
  TF_REQUIRE(base && dirs && g_out && (g_base || g_dirs), TF_EINVAL, "tf_cube_lookup_bwd_dirs: null pointer");
  cube_lookup_bwd_kernel<<<tf_blocks(m, 256), 256, 0, (hipStream_t)stream>>>(base, res, dirs, m, apply_exp, g_out, g_base, g_dirs);
  TF_LAUNCH_CHECK("tf_cube_lookup_bwd_dirs");
  return TF_OK;
}
