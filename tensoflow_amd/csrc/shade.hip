// Monte-Carlo shading integral of MCShadingNetwork.shade_mixed (network/fields.py:1075-1235):
// direction construction from the flow samples + the fixed cosine set, pdfs, GGX/Schlick BRDF
// weights, and the final reduction + sRGB.  Elementwise over pn*T slots; HBM-bound
// (reads 12 B of flow output per slot, writes 24 B of direction + weight).
#include "cube.h"
#include "mfma_mlp.h"   // tf_sincos_small
#include "tf_common.h"

static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kTwoPi = 6.28318530717958647692f;
static constexpr float kHalfPi_ = 1.57079632679489661923f;
static constexpr float kEPS = 1e-6f;

#include "shade_frame.h"

__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float sat(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__device__ __forceinline__ float ggx_d(float NoH, float a) {
  const float a2 = a * a;
  const float den = NoH * NoH * (a2 - 1.f) + 1.f;
  return a2 / fmaxf(kPi * den * den, kEPS);
}
__device__ __forceinline__ float schlick_g1(float c, float a) {
  const float k = a / 2.f;
  return c / (c * (1.f - k) + k + 1e-5f);
}

// geometry_ggx_smith_correlated (fields.py:1000-1008): 1 / (1 + L(NoV) + L(NoL)), L(c) = (sqrt(1 + a^2 tan^2) - 1) / 2 with a^2 = roughness^2
// and tan^2 = (1 - c^2) / (c^2 + 1e-7)
__device__ __forceinline__ float smith_lambda(float c, float a2) {
  const float c2 = c * c;
  const float t2 = (1.f - c2) / (c2 + 1e-7f);
  return 0.5f * sqrtf(1.f + a2 * t2) - 0.5f;
}
#define TF_SHADE_WHOLE_DIFFUSE 1   // mode bits of the direction kernels (tf_shade_dirs_ex): the flow of that lobe samples the outgoing direction
#define TF_SHADE_WHOLE_SPECULAR 2
#define TF_SHADE_GGX_SMITH 4       // cfg geometry_type = 'ggx_smith' instead of 'schlick'

// 12-byte rows as ONE memory instruction (global_load_dwordx3 / global_store_dwordx3) instead of three strided dword accesses
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float* p) { return *reinterpret_cast<const F3*>(p); }
__device__ __forceinline__ void st3(float* p, float x, float y, float z) { *reinterpret_cast<F3*>(p) = F3{x, y, z}; }

// The kernel is vector-instruction bound (201 M slots x ~600 instructions per step of the bench), most of them inside libm:
// sinf / cosf carry their large-argument path to every call site although every angle here is below 4 pi, powf(x, 5) and expf
// their full-range scaffolding.  (tf_sincos_small: max abs error 1.2e-7 for |x| < 200.)
__device__ __forceinline__ float pow5(float x) { const float x2 = x * x; return x2 * x2 * x; }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__global__ void __launch_bounds__(256) shade_dirs_kernel(
    const float* __restrict__ normals, const float* __restrict__ view, const float* __restrict__ metallic,
    const float* __restrict__ roughness, const float* __restrict__ albedo, const float* __restrict__ ang_d,
    const float* __restrict__ logq_d, int sd, const float* __restrict__ fixed_d, const float* __restrict__ az_jitter, int nf,
    const float* __restrict__ ang_s, const float* __restrict__ logq_s, const float* __restrict__ fixed_s,
    const float* __restrict__ az_jitter_s, int ss, long long pn, float* __restrict__ dirs,
    float* __restrict__ wgt, unsigned char* __restrict__ spec_mask, unsigned char* __restrict__ live,
    float* __restrict__ flow_logjac, const int* __restrict__ slot_of_pos, int pos0, int npos, int whole_mask) {
  const int T = sd + nf + ss;
  // rows [pos0, pos0 + npos) of every point (the whole point by default): a caller whose direction sets become ready one after the
  // other builds the rows of the sets it has while the next set is still being sampled
  const long long work = (long long)blockIdx.x * 256 + threadIdx.x;
  if (work >= pn * npos) return;
  const long long pt = work / npos;
  const long long e = pt * T + pos0 + (work - pt * npos);
  // row e of dirs / wgt / live is POSITION e % T of its point; the slot (which sample of which direction set) it holds is
  // slot_of_pos[position] when the caller stores the rays of a point in traversal order (tf_bvh_trace then reads and writes
  // consecutive rows from consecutive lanes), the position itself otherwise.  spec_mask / flow_logjac stay indexed by sample.
  const int slot = slot_of_pos ? slot_of_pos[(int)(e % T)] : (int)(e % T);
  Frame F;
  make_frame(normals + 3 * pt, F);
  float v[3] = {view[3 * pt], view[3 * pt + 1], view[3 * pt + 2]};
  normalize3(v);
  const float met = metallic[pt], rough = roughness[pt];
  const float alb[3] = {albedo[3 * pt], albedo[3 * pt + 1], albedo[3 * pt + 2]};
  float dir[3], pdf;
  const bool is_spec = slot >= sd + nf;
  if (is_spec && fixed_s) {
    // fixed specular set: GGX half-vector warp of the Fibonacci samples by the point's (squared) roughness
    // (sample_specular_directions, fields.py:858-903) -- the sampler of the non-NIS pass of shade_mixed
    const int s = slot - sd - nf;
    float phi = fixed_s[2 * s] * kPi * 2.f;
    const float el = fixed_s[2 * s + 1];
    if (az_jitter_s) phi = fmodf(phi + az_jitter_s[pt] * kPi * 2.f, kTwoPi);
    const float ct = sqrtf(fmaxf((1.f - el) / fmaxf(1.f + (rough * rough - 1.f) * el, kEPS), kEPS));
    const float st = sqrtf(fmaxf(1.f - ct * ct, kEPS));
    float sphi, cphi;
    tf_sincos_small(phi, sphi, cphi);
    const float cxh = cphi * st, cyh = sphi * st;
    float H[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) H[k] = cxh * F.x[k] + cyh * F.y[k] + ct * F.n[k];
    const float VoH = sat(dot3(v, H));
#pragma unroll
    for (int k = 0; k < 3; ++k) dir[k] = VoH * H[k] * 2.f - v[k];
    const float NoH = fmaxf(ct, 0.f);
    float sj, cj;
    tf_sincos_small((1.f - el) * kPi / 2.f, sj, cj);
    pdf = ggx_d(NoH, rough) * NoH / fmaxf(4.f * VoH, kEPS) * (cj * kPi / 2.f);
  } else if (slot < sd || is_spec) {
    // flow sample = half-vector angles in [0,1]^2 (fields.py:1085-1108 / :1164-1188)
    const long long r = is_spec ? pt * ss + (slot - sd - nf) : pt * sd + slot;
    const float* ang = is_spec ? ang_s : ang_d;
    const float lq = (is_spec ? logq_s : logq_d)[r];
    const float phi = ang[2 * r] * kTwoPi, theta = ang[2 * r + 1] * kHalfPi_;
    float st, ct, sphi, cphi;
    tf_sincos_small(theta, st, ct);
    tf_sincos_small(phi, sphi, cphi);
    const float cxh = st * cphi, cyh = st * sphi;
    float H[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) H[k] = cxh * F.x[k] + cyh * F.y[k] + ct * F.n[k];
    float jac;      // d (angles in [0,1]^2) -> d (outgoing direction): the density's denominator and the NIS loss' log-Jacobian
    if (whole_mask & (is_spec ? 2 : 1)) {
      // use_half_diffuse / use_half_specular = False (fields.py:1117-1134, :1190-1203): the flow samples the OUTGOING direction itself
#pragma unroll
      for (int k = 0; k < 3; ++k) dir[k] = H[k];
      jac = kPi * kPi * st;
    } else {
      const float HoV = sat(dot3(v, H));
#pragma unroll
      for (int k = 0; k < 3; ++k) dir[k] = HoV * H[k] * 2.f - v[k];
      jac = 4.f * kPi * kPi * HoV * st;
    }
    pdf = fast_exp(-fminf(fmaxf(lq, -8.f), 8.f)) / fmaxf(jac, kEPS);
    // log of the (angles -> outgoing direction) Jacobian used by the NIS loss (fields.py:1275-1280, :1312-1318)
    if (flow_logjac) flow_logjac[pt * (sd + ss) + (is_spec ? sd + (slot - sd - nf) : slot)] = logf(fmaxf(jac, kEPS));
  } else {
    // fixed cosine set (fields.py:824-847)
    const int s = slot - sd;
    float az = fixed_d[2 * s] * kPi * 2.f;
    const float el = fixed_d[2 * s + 1];
    if (az_jitter) az = fmodf(az + az_jitter[pt] * kPi * 2.f, kTwoPi);
    const float el_sqrt = sqrtf(el + 1e-7f);
    const float cz = sqrtf(1.f - el + 1e-7f);
    float saz, caz, sj, cj;
    tf_sincos_small(az, saz, caz);
    tf_sincos_small((1.f - el) * kPi / 2.f, sj, cj);
    const float cx = el_sqrt * caz, cy = el_sqrt * saz;
#pragma unroll
    for (int k = 0; k < 3; ++k) dir[k] = cx * F.x[k] + cy * F.y[k] + cz * F.n[k];
    pdf = sat(dot3(dir, F.n)) / kPi * (cj * kPi / 2.f);
  }
  float w[3];
  if (!is_spec) {
    const float kd = 1.f - met;
    const float c = sat(dot3(dir, F.n)) / kPi;
    const float inv = 1.f / fmaxf(pdf, kEPS) / (float)(sd + nf);
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = alb[k] * kd * c * inv;
  } else {
    const bool keep = dot3(dir, F.n) > 0.f;
    spec_mask[pt * ss + (slot - sd - nf)] = keep ? 1 : 0;
    float Hs[3] = {v[0] + dir[0], v[1] + dir[1], v[2] + dir[2]};
    normalize3(Hs);
    const float HoV = sat(dot3(Hs, v));
    const float f5 = pow5(sat(1.f - HoV));
    const float NoV = sat(dot3(F.n, v)), NoL = sat(dot3(F.n, dir)), NoH = sat(dot3(F.n, Hs));
    const float geo = (whole_mask & TF_SHADE_GGX_SMITH) ? 1.f / (1.f + smith_lambda(NoV, rough * rough) + smith_lambda(NoL, rough * rough))
                                                        : schlick_g1(NoV, rough) * schlick_g1(NoL, rough);
    const float D = ggx_d(NoH, rough);
    const float inv = 1.f / fmaxf(4.f * NoV, kEPS) / fmaxf(pdf, kEPS) / (float)ss;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float F0 = 0.04f * (1.f - met) + met * alb[k];
      const float fres = F0 + (1.f - F0) * f5;
      w[k] = keep ? D * fres * geo * inv : 0.f;
    }
  }
  st3(dirs + 3 * e, dir[0], dir[1], dir[2]);
  st3(wgt + 3 * e, w[0], w[1], w[2]);
  if (live) live[e] = (w[0] != 0.f || w[1] != 0.f || w[2] != 0.f) ? 1 : 0;
}

// Backward of the BRDF weights wrt the per-point materials: given g_wgt [pn,T,3] accumulates
//   g_albedo [pn,3], g_metallic [pn], g_roughness [pn]   (directions and pdfs do not depend on trainable parameters:
//   the sampling flows are frozen copies, fields.py:1054-1065).  One wave per point, slots strided over lanes.
__global__ void __launch_bounds__(256) shade_dirs_bwd_kernel(
    const float* __restrict__ normals, const float* __restrict__ view, const float* __restrict__ metallic,
    const float* __restrict__ roughness, const float* __restrict__ albedo, const float* __restrict__ dirs,
    const float* __restrict__ wgt, const float* __restrict__ g_wgt, int sd, int nf, int ss, long long pn,
    float* __restrict__ g_albedo, float* __restrict__ g_metallic, float* __restrict__ g_roughness, int mode) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= pn) return;
  const int T = sd + nf + ss;
  Frame F;
  make_frame(normals + 3 * pt, F);
  float v[3] = {view[3 * pt], view[3 * pt + 1], view[3 * pt + 2]};
  normalize3(v);
  const float met = metallic[pt], a = roughness[pt];
  const float alb[3] = {albedo[3 * pt], albedo[3 * pt + 1], albedo[3 * pt + 2]};
  float ga[3] = {0, 0, 0}, gm = 0.f, gr = 0.f;
  for (int t = lane; t < T; t += 64) {
    const long long e = pt * T + t;
    const float gw[3] = {g_wgt[3 * e], g_wgt[3 * e + 1], g_wgt[3 * e + 2]};
    const float w[3] = {wgt[3 * e], wgt[3 * e + 1], wgt[3 * e + 2]};
    if (t < sd + nf) {
      // w_k = alb_k * (1 - met) * C   ->  C = w_k / (alb_k (1-met)); use the stored weight to avoid recomputing pdfs
      const float kd = 1.f - met;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float C = (alb[k] * kd != 0.f) ? w[k] / (alb[k] * kd) : 0.f;
        ga[k] += gw[k] * kd * C;
        gm -= gw[k] * alb[k] * C;
      }
    } else {
      if (w[0] == 0.f && w[1] == 0.f && w[2] == 0.f) continue;          // masked (below horizon)
      const float dir[3] = {dirs[3 * e], dirs[3 * e + 1], dirs[3 * e + 2]};
      float Hs[3] = {v[0] + dir[0], v[1] + dir[1], v[2] + dir[2]};
      normalize3(Hs);
      const float HoV = sat(dot3(Hs, v));
      const float f5 = powf(sat(1.f - HoV), 5.f);
      const float NoV = sat(dot3(F.n, v)), NoL = sat(dot3(F.n, dir)), NoH = sat(dot3(F.n, Hs));
      // D(a), G(a) and their logarithmic derivatives
      const float a2 = a * a, q = NoH * NoH * (a2 - 1.f) + 1.f;
      const float dlogD = (kPi * q * q > kEPS) ? (2.f / a) * (1.f - 2.f * a2 * NoH * NoH / q) : 2.f / a;
      const float k_ = a / 2.f;
      const float denV = NoV * (1.f - k_) + k_ + 1e-5f, denL = NoL * (1.f - k_) + k_ + 1e-5f;
      float dlogG = -(1.f - NoV) / (2.f * denV) - (1.f - NoL) / (2.f * denL);
      if (mode & TF_SHADE_GGX_SMITH) {
        // G = 1 / (1 + L_v + L_l), L = (sqrt(1 + a^2 t) - 1) / 2  ->  dL / da = a t / (2 sqrt(1 + a^2 t))
        const float tv = (1.f - NoV * NoV) / (NoV * NoV + 1e-7f), tl = (1.f - NoL * NoL) / (NoL * NoL + 1e-7f);
        const float rv = sqrtf(1.f + a2 * tv), rl = sqrtf(1.f + a2 * tl);
        dlogG = -(a * tv / (2.f * rv) + a * tl / (2.f * rl)) / (1.f + (0.5f * rv - 0.5f) + (0.5f * rl - 0.5f));
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float F0 = 0.04f * (1.f - met) + met * alb[k];
        const float fres = F0 + (1.f - F0) * f5;
        // w_k = D * fres_k * G * inv  ->  d w_k / d a = w_k (dlogD + dlogG);  d w_k / d F0 = w_k (1 - f5) / fres_k
        gr += gw[k] * w[k] * (dlogD + dlogG);
        const float dwdF0 = fres != 0.f ? w[k] * (1.f - f5) / fres : 0.f;
        gm += gw[k] * dwdF0 * (alb[k] - 0.04f);
        ga[k] += gw[k] * dwdF0 * met;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gm += __shfl_xor(gm, o); gr += __shfl_xor(gr, o);
#pragma unroll
    for (int k = 0; k < 3; ++k) ga[k] += __shfl_xor(ga[k], o);
  }
  if (lane == 0) {
    g_metallic[pt] = gm; g_roughness[pt] = gr;
#pragma unroll
    for (int k = 0; k < 3; ++k) g_albedo[3 * pt + k] = ga[k];
  }
}

__device__ __forceinline__ float srgb(float x) {  // utils/raw_utils.py:4-11
  const float eps = 1.1920928955078125e-07f;
  const float s0 = 12.92f * x;
  const float s1 = (211.f * powf(fmaxf(x, eps), 5.f / 12.f) - 11.f) / 200.f;
  return x <= 0.0031308f ? s0 : s1;
}

// one wave per point
__global__ void __launch_bounds__(256) shade_reduce_kernel(const float* __restrict__ wgt, const float* __restrict__ lights,
                                                           long long pn, int n_diff, int ss, float* __restrict__ colors,
                                                           float* __restrict__ diffuse_lin, float* __restrict__ specular_lin) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= pn) return;
  const int T = n_diff + ss;
  float d[3] = {0, 0, 0}, s[3] = {0, 0, 0};
  for (int t = lane; t < T; t += 64) {
    const long long e = (pt * T + t) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float c = wgt[e + k] * lights[e + k];
      if (t < n_diff) d[k] += c; else s[k] += c;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { d[k] += __shfl_xor(d[k], o); s[k] += __shfl_xor(s[k], o); }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      colors[3 * pt + k] = srgb(d[k] + s[k]);
      if (diffuse_lin) diffuse_lin[3 * pt + k] = d[k];
      if (specular_lin) specular_lin[3 * pt + k] = s[k];
    }
  }
}

// Reduction with the environment light evaluated on the fly: a ray that missed the mesh takes exp(cube(dirs)) * near-mask
// (MCShadingNetwork.get_lights, fields.py:951-975 + EnvLight.direct_light) computed HERE instead of being written to and
// re-read from a [pn,T,3] light array; a ray that hit reads the inner-light result from `hit_lights` (rows of missing rays are
// never touched).  Slots with zero weight (culled rays, masked specular samples) cost nothing.
__global__ void __launch_bounds__(256) shade_reduce_env_kernel(const float* __restrict__ wgt, const float* __restrict__ dirs,
                                                               const float* __restrict__ depth, const unsigned char* __restrict__ hit,
                                                               const float* __restrict__ hit_lights, const float* __restrict__ env,
                                                               int env_res, float near_eps, long long pn, int n_diff, int ss,
                                                               float* __restrict__ colors, float* __restrict__ diffuse_lin,
                                                               float* __restrict__ specular_lin, const int* __restrict__ slot_of_pos) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= pn) return;
  const int T = n_diff + ss;
  float d[3] = {0, 0, 0}, s[3] = {0, 0, 0};
  for (int t = lane; t < T; t += 64) {
    const long long r = pt * T + t, e = r * 3;
    // all four rows are requested together (85 % of the rays miss and need every one of them).  Measured neutral against fetching
    // them behind the branches (2.65 ms either way).  Round 6, measured and not adopted: the rows of 2 / 4 / 6 slots per lane requested
    // together (2.61 / 2.87 / 2.78 ms against 2.47), RGBA-padded texels (one aligned dwordx4 per tap: 2.455 against 2.450), the cube
    // edge taps in integer arithmetic (cube.h; kept: same texels, 2.51 -> 2.50).  Counters: 230 vector instructions per
    // ray, vector unit ~50 % busy at 6.6 waves per SIMD, 44 % of wave time waiting: row round trip + dependent texel round trip per trip
    const F3 w = ld3(wgt + e);
    const F3 dd = ld3(dirs + e);
    const float dr = depth[r];
    const bool hr = hit ? hit[r] != 0 : dr < TF_MISS_DEPTH;
    const float w0 = w.x, w1 = w.y, w2 = w.z;
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
    if (w0 != 0.f || w1 != 0.f || w2 != 0.f) {
      if (hr || !env) {      // env == NULL: the miss rows hold the outer-light net's answer (tf_outer_light_indexed_fwd)
        const F3 hl = ld3(hit_lights + e);
        l0 = hl.x; l1 = hl.y; l2 = hl.z;
      } else if (dr > near_eps) {
        cube_fetch_rgb(env, env_res, dd.x, dd.y, dd.z, l0, l1, l2);
        l0 = fast_exp(l0); l1 = fast_exp(l1); l2 = fast_exp(l2);
      }
    }
    const float c0 = w0 * l0, c1 = w1 * l1, c2 = w2 * l2;
    const int slot = slot_of_pos ? slot_of_pos[t] : t;      // rows in traversal order: the lobe a row belongs to follows its slot
    if (slot < n_diff) { d[0] += c0; d[1] += c1; d[2] += c2; } else { s[0] += c0; s[1] += c1; s[2] += c2; }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { d[k] += __shfl_xor(d[k], o); s[k] += __shfl_xor(s[k], o); }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      colors[3 * pt + k] = srgb(d[k] + s[k]);
      if (diffuse_lin) diffuse_lin[3 * pt + k] = d[k];
      if (specular_lin) specular_lin[3 * pt + k] = s[k];
    }
  }
}

extern "C" int tf_shade_reduce_env(const float* wgt, const float* dirs, const float* depth, const uint8_t* hit,
                                   const float* hit_lights, const float* env_base, int32_t env_res, float near_eps, int64_t pn,
                                   int32_t n_diffuse, int32_t ss, float* colors, float* diffuse_lin, float* specular_lin,
                                   const int32_t* slot_of_pos, tf_stream_t stream) {
  TF_REQUIRE(pn >= 0 && n_diffuse >= 0 && ss >= 0 && (env_res > 0 || !env_base), TF_ESHAPE, "tf_shade_reduce_env: negative size / env_res <= 0");
  if (pn == 0) return TF_OK;
  TF_REQUIRE(wgt && dirs && depth && hit_lights && colors, TF_EINVAL, "tf_shade_reduce_env: null pointer");
  shade_reduce_env_kernel<<<tf_blocks(pn, 4), 256, 0, (hipStream_t)stream>>>(wgt, dirs, depth, hit, hit_lights, env_base, env_res,
                                                                            near_eps, pn, n_diffuse, ss, colors, diffuse_lin,
                                                                            specular_lin, slot_of_pos);
  TF_LAUNCH_CHECK("tf_shade_reduce_env");
  return TF_OK;
}

// The reduction WITH the auxiliary per-point statistics of shade_mixed's output dict (fields.py:1232-1256, :1288-1291): besides the
// colour sums, per point
//   aux[0..2]  sum of the lights of the n_diff diffuse rays                     -> diffuse_light, approximate_light
//   aux[3..5]  sum of the lights of the UNMASKED specular rays                  -> specular_light
//   aux[6..8]  the same restricted to rays that hit the mesh, aux[9] their count -> indirect_light, visibility
//   aux[10..12] count, mean, M2 (sum of squared deviations) of g = mean_c(fx_c) / p over the unmasked specular rays
//   aux[13..14] mean, M2 of g over the n_diff diffuse rays                      -> variance, variance_{diffuse,specular}_vis
// (Welford per lane, Chan's merge across the wave: the reference's E[g^2] - E[g]^2 cancels in fp32, this does not.)  Every ray's
// light is evaluated, zero weight or not: the unweighted maps average over all of them.  `lights` [pn,T,3] non-NULL: the light
// array is given (training composition); NULL: hit rows from hit_lights, missing rays from the cube map as in shade_reduce_env.
__device__ __forceinline__ void welford_add(float& n, float& mean, float& m2, float g) {
  n += 1.f;
  const float dl = g - mean;
  mean += dl / n;
  m2 += dl * (g - mean);
}
__device__ __forceinline__ void welford_merge_wave(float& n, float& mean, float& m2) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float nb = __shfl_xor(n, o), mb = __shfl_xor(mean, o), qb = __shfl_xor(m2, o);
    const float nt = n + nb;
    if (nt > 0.f) {
      const float dl = mb - mean;
      mean += dl * (nb / nt);
      m2 += qb + dl * dl * (n * nb / nt);
    }
    n = nt;
  }
}

__global__ void __launch_bounds__(256) shade_reduce_aux_kernel(const float* __restrict__ wgt, const float* __restrict__ lights,
                                                               const float* __restrict__ dirs, const float* __restrict__ depth,
                                                               const unsigned char* __restrict__ hit, const float* __restrict__ hit_lights,
                                                               const float* __restrict__ env, int env_res, float near_eps,
                                                               const unsigned char* __restrict__ spec_mask, long long pn, int n_diff, int ss,
                                                               float* __restrict__ colors, float* __restrict__ diffuse_lin,
                                                               float* __restrict__ specular_lin, float* __restrict__ aux,
                                                               const int* __restrict__ slot_of_pos) {
  const int lane = threadIdx.x & 63;
  const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= pn) return;
  const int T = n_diff + ss;
  float d[3] = {0, 0, 0}, s[3] = {0, 0, 0}, sd_[3] = {0, 0, 0}, ss_[3] = {0, 0, 0}, si_[3] = {0, 0, 0};
  float nhit = 0.f, n_s = 0.f, mean_s = 0.f, m2_s = 0.f, n_d = 0.f, mean_d = 0.f, m2_d = 0.f;
  const float gd = (float)n_diff / 3.f, gs = (float)ss / 3.f;      // wgt = (fx / p) / count per channel: g = mean over channels x count
  for (int t = lane; t < T; t += 64) {
    const long long r = pt * T + t, e = r * 3;
    const F3 w = ld3(wgt + e);
    const float dr = depth ? depth[r] : 0.f;
    const bool hr = hit ? hit[r] != 0 : dr < TF_MISS_DEPTH;
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
    if (lights) {
      const F3 l = ld3(lights + e);
      l0 = l.x; l1 = l.y; l2 = l.z;
    } else if (hr || !env) {
      const F3 hl = ld3(hit_lights + e);
      l0 = hl.x; l1 = hl.y; l2 = hl.z;
    } else if (dr > near_eps) {
      const F3 dd = ld3(dirs + e);
      cube_fetch_rgb(env, env_res, dd.x, dd.y, dd.z, l0, l1, l2);
      l0 = fast_exp(l0); l1 = fast_exp(l1); l2 = fast_exp(l2);
    }
    const float c0 = w.x * l0, c1 = w.y * l1, c2 = w.z * l2;
    const int slot = slot_of_pos ? slot_of_pos[t] : t;
    if (slot < n_diff) {
      d[0] += c0; d[1] += c1; d[2] += c2;
      sd_[0] += l0; sd_[1] += l1; sd_[2] += l2;
      welford_add(n_d, mean_d, m2_d, (c0 + c1 + c2) * gd);
    } else {
      s[0] += c0; s[1] += c1; s[2] += c2;
      if (spec_mask[pt * ss + (slot - n_diff)]) {
        ss_[0] += l0; ss_[1] += l1; ss_[2] += l2;
        if (hr) { si_[0] += l0; si_[1] += l1; si_[2] += l2; nhit += 1.f; }
        welford_add(n_s, mean_s, m2_s, (c0 + c1 + c2) * gs);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      d[k] += __shfl_xor(d[k], o); s[k] += __shfl_xor(s[k], o);
      sd_[k] += __shfl_xor(sd_[k], o); ss_[k] += __shfl_xor(ss_[k], o); si_[k] += __shfl_xor(si_[k], o);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nhit += __shfl_xor(nhit, o);
  welford_merge_wave(n_s, mean_s, m2_s);
  welford_merge_wave(n_d, mean_d, m2_d);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (colors) colors[3 * pt + k] = srgb(d[k] + s[k]);
      if (diffuse_lin) diffuse_lin[3 * pt + k] = d[k];
      if (specular_lin) specular_lin[3 * pt + k] = s[k];
      aux[16 * pt + k] = sd_[k]; aux[16 * pt + 3 + k] = ss_[k]; aux[16 * pt + 6 + k] = si_[k];
    }
    aux[16 * pt + 9] = nhit; aux[16 * pt + 10] = n_s; aux[16 * pt + 11] = mean_s; aux[16 * pt + 12] = m2_s;
    aux[16 * pt + 13] = mean_d; aux[16 * pt + 14] = m2_d; aux[16 * pt + 15] = 0.f;
  }
}

extern "C" int tf_shade_reduce_aux(const float* wgt, const float* lights, const float* dirs, const float* depth, const uint8_t* hit,
                                   const float* hit_lights, const float* env_base, int32_t env_res, float near_eps,
                                   const uint8_t* spec_mask, int64_t pn, int32_t n_diffuse, int32_t ss, float* colors,
                                   float* diffuse_lin, float* specular_lin, float* aux, const int32_t* slot_of_pos, tf_stream_t stream) {
  TF_REQUIRE(pn >= 0 && n_diffuse >= 0 && ss >= 0 && (env_res > 0 || !env_base), TF_ESHAPE, "tf_shade_reduce_aux: negative size / env_res <= 0");
  if (pn == 0) return TF_OK;
  TF_REQUIRE(wgt && aux && (ss == 0 || spec_mask), TF_EINVAL, "tf_shade_reduce_aux: null pointer");
  TF_REQUIRE(lights || (dirs && depth && hit_lights), TF_EINVAL, "tf_shade_reduce_aux: either `lights` or (dirs, depth, hit_lights)");
  TF_REQUIRE(!lights || hit || depth, TF_EINVAL, "tf_shade_reduce_aux: with `lights`, the hit flags come from `hit` or `depth`");
  shade_reduce_aux_kernel<<<tf_blocks(pn, 4), 256, 0, (hipStream_t)stream>>>(wgt, lights, dirs, depth, hit, hit_lights, env_base, env_res,
                                                                            near_eps, spec_mask, pn, n_diffuse, ss, colors, diffuse_lin,
                                                                            specular_lin, aux, slot_of_pos);
  TF_LAUNCH_CHECK("tf_shade_reduce_aux");
  return TF_OK;
}

static int shade_dirs_launch(const float* normals, const float* view, const float* metallic, const float* roughness,
                             const float* albedo, const float* ang_d, const float* logq_d, int32_t sd, const float* fixed_d,
                             const float* az_jitter, int32_t nf, const float* ang_s, const float* logq_s, const float* fixed_s,
                             const float* az_jitter_s, int32_t ss, int64_t pn, float* dirs, float* wgt, uint8_t* spec_mask,
                             uint8_t* live, float* flow_logjac, const int32_t* slot_of_pos, int32_t pos0, int32_t npos, tf_stream_t stream,
                             const char* who, int32_t whole_mask = 0) {
  TF_REQUIRE(pn >= 0 && sd >= 0 && nf >= 0 && ss >= 0, TF_ESHAPE, "%s: negative size", who);
  if (npos < 0) { pos0 = 0; npos = sd + nf + ss; }
  TF_REQUIRE(pos0 >= 0 && pos0 + npos <= sd + nf + ss, TF_ESHAPE, "%s: row range [%d, %d) outside [0, %d)", who, pos0, pos0 + npos, sd + nf + ss);
  if (pn == 0 || npos == 0) return TF_OK;
  TF_REQUIRE(normals && view && metallic && roughness && albedo && dirs && wgt, TF_EINVAL, "%s: null pointer", who);
  TF_REQUIRE((sd == 0 || (ang_d && logq_d)) && (nf == 0 || fixed_d) && (ss == 0 || (((ang_s && logq_s) || fixed_s) && spec_mask)),
             TF_EINVAL, "%s: null sample pointer for a non-empty sample set", who);
  TF_REQUIRE(!(fixed_s && ang_s), TF_EINVAL, "%s: the specular set is either flow-sampled (ang_s) or fixed (fixed_s), not both", who);
  long long work = (long long)pn * npos;
  shade_dirs_kernel<<<tf_blocks(work, 256), 256, 0, (hipStream_t)stream>>>(normals, view, metallic, roughness, albedo, ang_d,
                                                                          logq_d, sd, fixed_d, az_jitter, nf, ang_s, logq_s, fixed_s,
                                                                          az_jitter_s, ss, pn, dirs, wgt, spec_mask, live,
                                                                          fixed_s ? nullptr : flow_logjac, slot_of_pos, pos0, npos, whole_mask);
  TF_LAUNCH_CHECK(who);
  return TF_OK;
}

extern "C" int tf_shade_dirs(const float* normals, const float* view, const float* metallic, const float* roughness,
                             const float* albedo, const float* ang_d, const float* logq_d, int32_t sd, const float* fixed_d,
                             const float* az_jitter, int32_t nf, const float* ang_s, const float* logq_s, int32_t ss,
                             int64_t pn, float* dirs, float* wgt, uint8_t* spec_mask, uint8_t* live, float* flow_logjac,
                             const int32_t* slot_of_pos, int32_t row_begin, int32_t row_count, tf_stream_t stream) {
  return shade_dirs_launch(normals, view, metallic, roughness, albedo, ang_d, logq_d, sd, fixed_d, az_jitter, nf, ang_s, logq_s,
                           nullptr, nullptr, ss, pn, dirs, wgt, spec_mask, live, flow_logjac, slot_of_pos, row_begin, row_count, stream,
                           "tf_shade_dirs");
}

extern "C" int tf_shade_dirs_whole(const float* normals, const float* view, const float* metallic, const float* roughness,
                                   const float* albedo, const float* ang_d, const float* logq_d, int32_t sd, const float* fixed_d,
                                   const float* az_jitter, int32_t nf, const float* ang_s, const float* logq_s, int32_t ss,
                                   int64_t pn, float* dirs, float* wgt, uint8_t* spec_mask, uint8_t* live, float* flow_logjac,
                                   const int32_t* slot_of_pos, int32_t row_begin, int32_t row_count, int32_t whole_mask, tf_stream_t stream) {
  TF_REQUIRE(whole_mask >= 0 && whole_mask <= 7, TF_EINVAL, "tf_shade_dirs_whole: whole_mask must be 0..7");
  return shade_dirs_launch(normals, view, metallic, roughness, albedo, ang_d, logq_d, sd, fixed_d, az_jitter, nf, ang_s, logq_s,
                           nullptr, nullptr, ss, pn, dirs, wgt, spec_mask, live, flow_logjac, slot_of_pos, row_begin, row_count, stream,
                           "tf_shade_dirs_whole", whole_mask);
}

extern "C" int tf_shade_dirs_fixed(const float* normals, const float* view, const float* metallic, const float* roughness,
                                   const float* albedo, const float* fixed_d, const float* az_jitter, int32_t nf,
                                   const float* fixed_s, const float* az_jitter_s, int32_t ss, int64_t pn, float* dirs, float* wgt,
                                   uint8_t* spec_mask, uint8_t* live, tf_stream_t stream) {
  return shade_dirs_launch(normals, view, metallic, roughness, albedo, nullptr, nullptr, 0, fixed_d, az_jitter, nf, nullptr, nullptr,
                           fixed_s, az_jitter_s, ss, pn, dirs, wgt, spec_mask, live, nullptr, nullptr, 0, -1, stream, "tf_shade_dirs_fixed");
}

extern "C" int tf_shade_dirs_fixed_mode(const float* normals, const float* view, const float* metallic, const float* roughness,
                                        const float* albedo, const float* fixed_d, const float* az_jitter, int32_t nf,
                                        const float* fixed_s, const float* az_jitter_s, int32_t ss, int64_t pn, float* dirs, float* wgt,
                                        uint8_t* spec_mask, uint8_t* live, int32_t mode, tf_stream_t stream) {
  TF_REQUIRE(mode == 0 || mode == TF_SHADE_GGX_SMITH, TF_EINVAL, "tf_shade_dirs_fixed_mode: mode must be 0 or 4 (ggx_smith geometry)");
  return shade_dirs_launch(normals, view, metallic, roughness, albedo, nullptr, nullptr, 0, fixed_d, az_jitter, nf, nullptr, nullptr,
                           fixed_s, az_jitter_s, ss, pn, dirs, wgt, spec_mask, live, nullptr, nullptr, 0, -1, stream, "tf_shade_dirs_fixed_mode", mode);
}

extern "C" int tf_shade_reduce(const float* wgt, const float* lights, int64_t pn, int32_t n_diffuse, int32_t ss,
                               float* colors, float* diffuse_lin, float* specular_lin, tf_stream_t stream) {
  TF_REQUIRE(pn >= 0 && n_diffuse >= 0 && ss >= 0, TF_ESHAPE, "tf_shade_reduce: negative size");
  if (pn == 0) return TF_OK;
  TF_REQUIRE(wgt && lights && colors, TF_EINVAL, "tf_shade_reduce: null pointer");
  shade_reduce_kernel<<<tf_blocks(pn, 4), 256, 0, (hipStream_t)stream>>>(wgt, lights, pn, n_diffuse, ss, colors, diffuse_lin,
                                                                        specular_lin);
  TF_LAUNCH_CHECK("tf_shade_reduce");
  return TF_OK;
}

static int shade_dirs_bwd_launch(const float* normals, const float* view, const float* metallic, const float* roughness,
                                 const float* albedo, const float* dirs, const float* wgt, const float* g_wgt, int32_t sd, int32_t nf,
                                 int32_t ss, int64_t pn, float* g_albedo, float* g_metallic, float* g_roughness, int32_t mode,
                                 tf_stream_t stream, const char* who) {
  TF_REQUIRE(pn >= 0 && sd >= 0 && nf >= 0 && ss >= 0, TF_ESHAPE, "%s: negative size", who);
  TF_REQUIRE(mode == 0 || mode == TF_SHADE_GGX_SMITH, TF_EINVAL, "%s: mode must be 0 or 4 (ggx_smith geometry)", who);
  if (pn == 0) return TF_OK;
  TF_REQUIRE(normals && view && metallic && roughness && albedo && dirs && wgt && g_wgt && g_albedo && g_metallic && g_roughness,
             TF_EINVAL, "%s: null pointer", who);
  shade_dirs_bwd_kernel<<<tf_blocks(pn, 4), 256, 0, (hipStream_t)stream>>>(normals, view, metallic, roughness, albedo, dirs, wgt,
                                                                          g_wgt, sd, nf, ss, pn, g_albedo, g_metallic, g_roughness, mode);
  TF_LAUNCH_CHECK(who);
  return TF_OK;
}

extern "C" int tf_shade_dirs_bwd(const float* normals, const float* view, const float* metallic, const float* roughness,
                                 const float* albedo, const float* dirs, const float* wgt, const float* g_wgt, int32_t sd, int32_t nf,
                                 int32_t ss, int64_t pn, float* g_albedo, float* g_metallic, float* g_roughness, tf_stream_t stream) {
  return shade_dirs_bwd_launch(normals, view, metallic, roughness, albedo, dirs, wgt, g_wgt, sd, nf, ss, pn, g_albedo, g_metallic,
                               g_roughness, 0, stream, "tf_shade_dirs_bwd");
}

extern "C" int tf_shade_dirs_bwd_mode(const float* normals, const float* view, const float* metallic, const float* roughness,
                                      const float* albedo, const float* dirs, const float* wgt, const float* g_wgt, int32_t sd, int32_t nf,
                                      int32_t ss, int64_t pn, float* g_albedo, float* g_metallic, float* g_roughness, int32_t mode,
                                      tf_stream_t stream) {
  return shade_dirs_bwd_launch(normals, view, metallic, roughness, albedo, dirs, wgt, g_wgt, sd, nf, ss, pn, g_albedo, g_metallic,
                               g_roughness, mode, stream, "tf_shade_dirs_bwd_mode");
}
