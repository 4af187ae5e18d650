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

// ---- EnvLight.__call__ with a roughness (network/light.py:95-122): dr.texture(specular[0], dirs, mip=specular[1:], mip_level_bias,
// filter_mode='linear-mipmap-linear', boundary_mode='cube') = the bilinear cube fetch of the two levels around `mip`, blended by its
// fraction, then exp.  One launch each way instead of the per-level torch composition of rounds 1-4 (every level fetched for every
// sample, ~13 element-wise launches per level and autograd's mirror image: ~130 of a shape training step's 665 launches).
struct CubeStack {
  const float* tex[8];
  float* g_tex[8];
  int res[8];
  int n;
};

__device__ __forceinline__ void mip_split(float mip, int n, int& l0, int& l1, float& f) {
  const float fl = fminf(fmaxf(floorf(mip), 0.f), (float)(n - 1));   // (the caller clamps mip to [0, n - 1]; a stray value must not index outside the stack)
  l0 = (int)fl;
  f = mip - fl;
  l1 = min(l0 + 1, n - 1);
  if (l1 == l0) f = 0.f;
}

__global__ void __launch_bounds__(256) cube_lookup_mips_fwd_kernel(CubeStack S, const float* __restrict__ dirs, const float* __restrict__ mip,
                                                                   long long m, int apply_exp, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  int l0, l1;
  float f;
  mip_split(mip[i], S.n, l0, l1, f);
  const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
  float r0, g0, b0, r1 = 0.f, g1 = 0.f, b1 = 0.f;
  cube_fetch_rgb(S.tex[l0], S.res[l0], dx, dy, dz, r0, g0, b0);
  if (f != 0.f) cube_fetch_rgb(S.tex[l1], S.res[l1], dx, dy, dz, r1, g1, b1);
  // (1 - f) * a + f * b as the composition wrote it: two products and a sum
  float r = (1.f - f) * r0 + f * r1, g = (1.f - f) * g0 + f * g1, b = (1.f - f) * b0 + f * b1;
  if (apply_exp) { r = expf(r); g = expf(g); b = expf(b); }
  out[3 * i] = r; out[3 * i + 1] = g; out[3 * i + 2] = b;
}

// gradient of ONE level's bilinear fetch, the wave working together on the map scatter (see cube_lookup_bwd_kernel): (gr, gg, gb) is the
// gradient wrt this level's fetched value (zero for a lane that takes no part), `lvl` this lane's level; adds d / d direction into (ox, oy, oz)
// (S: the stack table in LDS -- indexed by a per-lane level; indexing the kernel-argument struct that way sends it through scratch memory)
__device__ __forceinline__ void cube_level_bwd(const CubeStack& S, int lvl, bool want_dirs, float dx, float dy, float dz, float gr, float gg,
                                               float gb, float& ox, float& oy, float& oz) {
  const float* base = S.tex[lvl];
  const int R = S.res[lvl];
  CubeTaps T;
  cube_taps(dx, dy, dz, R, T);
  const bool any_map = S.g_tex[0] != nullptr;
  if (any_map) {
    const int lane = threadIdx.x & 63;
    const int grp = lane / 3, c = lane - 3 * grp;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float w0 = T.w[t] * gr, w1 = T.w[t] * gg, w2 = T.w[t] * gb;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int src = (16 * sub + grp) & 63;
        const int idx_s = __shfl(T.idx[t], src);
        const int lvl_s = __shfl(lvl, src);
        const float v0 = __shfl(w0, src), v1 = __shfl(w1, src), v2 = __shfl(w2, src);
        const float val = c == 0 ? v0 : c == 1 ? v1 : v2;
        if (lane < 48 && val != 0.f) atomicAdd(S.g_tex[lvl_s] + 3LL * idx_s + c, val);
      }
    }
  }
  if (want_dirs && (gr != 0.f || gg != 0.f || gb != 0.f)) {
    int face;
    float x, y;
    cube_face_uv(dx, dy, dz, face, x, y);
    const float u = (x * 0.5f + 0.5f) * (float)R - 0.5f, v = (y * 0.5f + 0.5f) * (float)R - 0.5f;
    const float fu = u - floorf(u), fv = v - floorf(v);
    float wsum = 0.f, su = 0.f, sv = 0.f, au = 0.f, av = 0.f, gval = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int du = t & 1, dv = t >> 1;
      const float* p = base + 3LL * T.idx[t];
      const float gt = gr * p[0] + gg * p[1] + gb * p[2];
      gval += T.w[t] * gt;
      const bool dropped = T.w[t] == 0.f && ((du ? fu : 1.f - fu) * (dv ? fv : 1.f - fv)) != 0.f;
      if (dropped) continue;
      const float wu = (du ? 1.f : -1.f) * (dv ? fv : 1.f - fv), wv = (du ? fu : 1.f - fu) * (dv ? 1.f : -1.f);
      wsum += (du ? fu : 1.f - fu) * (dv ? fv : 1.f - fv);
      su += wu; sv += wv; au += wu * gt; av += wv * gt;
    }
    const float inv = 1.f / wsum;
    const float gu = (au - gval * su) * inv, gv = (av - gval * sv) * inv;
    const float gx = gu * 0.5f * (float)R, gy = gv * 0.5f * (float)R;
    if (face >= 4) {
      const float mz = 1.f / fabsf(dz), sg = dz < 0.f ? -1.f : 1.f;
      ox += sg * mz * gx; oy += -mz * gy; oz += -(x * gx + y * gy) * mz * sg;
    } else if (face >= 2) {
      const float my = 1.f / fabsf(dy), sg = dy < 0.f ? -1.f : 1.f;
      ox += my * gx; oz += sg * my * gy; oy += -(x * gx + y * gy) * my * sg;
    } else {
      const float mx = 1.f / fabsf(dx), sg = dx < 0.f ? -1.f : 1.f;
      oz += -sg * mx * gx; oy += -mx * gy; ox += -(x * gx + y * gy) * mx * sg;
    }
  }
}

__global__ void __launch_bounds__(256) cube_lookup_mips_bwd_kernel(CubeStack S_arg, const float* __restrict__ dirs, const float* __restrict__ mip,
                                                                   long long m, int apply_exp, const float* __restrict__ g_out,
                                                                   float* __restrict__ g_dirs, float* __restrict__ g_mip) {
  __shared__ CubeStack S;
  if (threadIdx.x < 8) {
    S.tex[threadIdx.x] = S_arg.tex[threadIdx.x];
    S.g_tex[threadIdx.x] = S_arg.g_tex[threadIdx.x];
    S.res[threadIdx.x] = S_arg.res[threadIdx.x];
  }
  if (threadIdx.x == 0) S.n = S_arg.n;
  __syncthreads();
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < m;                   // a lane past the end stays in the wave for the cooperative scatter, without a gradient
  if (!live) i = m - 1;
  int l0, l1;
  float f;
  mip_split(mip[i], S.n, l0, l1, f);
  const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
  float r0, g0, b0, r1 = 0.f, g1 = 0.f, b1 = 0.f;
  cube_fetch_rgb(S.tex[l0], S.res[l0], dx, dy, dz, r0, g0, b0);
  if (l1 != l0) cube_fetch_rgb(S.tex[l1], S.res[l1], dx, dy, dz, r1, g1, b1);   // also at f == 0: d out / d mip = level l1 - level l0
  float gr = live ? g_out[3 * i] : 0.f, gg = live ? g_out[3 * i + 1] : 0.f, gb = live ? g_out[3 * i + 2] : 0.f;
  if (apply_exp) {
    gr *= expf((1.f - f) * r0 + f * r1); gg *= expf((1.f - f) * g0 + f * g1); gb *= expf((1.f - f) * b0 + f * b1);
  }
  float ox = 0.f, oy = 0.f, oz = 0.f;
  const float w0 = 1.f - f, w1 = f;
  cube_level_bwd(S, l0, g_dirs != nullptr, dx, dy, dz, gr * w0, gg * w0, gb * w0, ox, oy, oz);
  // the second level: a wave-uniform decision (the scatter exchanges values across lanes); lanes with f == 0 carry zeros
  if (__any(w1 != 0.f)) cube_level_bwd(S, l1, g_dirs != nullptr, dx, dy, dz, gr * w1, gg * w1, gb * w1, ox, oy, oz);
  if (live) {
    if (g_dirs) { g_dirs[3 * i] = ox; g_dirs[3 * i + 1] = oy; g_dirs[3 * i + 2] = oz; }
    if (g_mip) g_mip[i] = (l1 == l0) ? 0.f : gr * (r1 - r0) + gg * (g1 - g0) + gb * (b1 - b0);
  }
}

static int cube_stack_of(const float* const* texs, float* const* g_texs, const int32_t* res, int32_t n_levels, CubeStack& S, const char* who) {
  TF_REQUIRE(n_levels >= 1 && n_levels <= 8 && texs && res, TF_ESHAPE, "%s: 1..8 levels", who);
  S.n = n_levels;
  for (int l = 0; l < 8; ++l) {
    S.tex[l] = l < n_levels ? texs[l] : nullptr;
    S.g_tex[l] = (g_texs && l < n_levels) ? g_texs[l] : nullptr;
    S.res[l] = l < n_levels ? res[l] : 1;
    if (l < n_levels) TF_REQUIRE(texs[l] && res[l] > 0, TF_EINVAL, "%s: level %d is null / empty", who, l);
    if (g_texs && l < n_levels) TF_REQUIRE(g_texs[l] != nullptr, TF_EINVAL, "%s: gradient map %d is null (pass g_texs = NULL for none)", who, l);
  }
  return TF_OK;
}

extern "C" int tf_cube_lookup_mips_fwd(const float* const* texs, const int32_t* res, int32_t n_levels, const float* dirs, const float* mip,
                                       int64_t m, int32_t apply_exp, float* out, tf_stream_t stream) {
  TF_REQUIRE(m >= 0, TF_ESHAPE, "tf_cube_lookup_mips_fwd: m < 0");
  if (m == 0) return TF_OK;
  TF_REQUIRE(dirs && mip && out, TF_EINVAL, "tf_cube_lookup_mips_fwd: null pointer");
  CubeStack S;
  if (int rc = cube_stack_of(texs, nullptr, res, n_levels, S, "tf_cube_lookup_mips_fwd")) return rc;
  cube_lookup_mips_fwd_kernel<<<tf_blocks(m, 256), 256, 0, (hipStream_t)stream>>>(S, dirs, mip, m, apply_exp, out);
  TF_LAUNCH_CHECK("tf_cube_lookup_mips_fwd");
  return TF_OK;
}

extern "C" int tf_cube_lookup_mips_bwd(const float* const* texs, const int32_t* res, int32_t n_levels, const float* dirs, const float* mip,
                                       int64_t m, int32_t apply_exp, const float* g_out, float* const* g_texs, float* g_dirs, float* g_mip,
                                       tf_stream_t stream) {
  TF_REQUIRE(m >= 0, TF_ESHAPE, "tf_cube_lookup_mips_bwd: m < 0");
  if (m == 0) return TF_OK;
  TF_REQUIRE(dirs && mip && g_out && (g_texs || g_dirs || g_mip), TF_EINVAL, "tf_cube_lookup_mips_bwd: null pointer");
  CubeStack S;
  if (int rc = cube_stack_of(texs, g_texs, res, n_levels, S, "tf_cube_lookup_mips_bwd")) return rc;
  cube_lookup_mips_bwd_kernel<<<tf_blocks(m, 256), 256, 0, (hipStream_t)stream>>>(S, dirs, mip, m, apply_exp, g_out, g_dirs, g_mip);
  TF_LAUNCH_CHECK("tf_cube_lookup_mips_bwd");
  return TF_OK;
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
