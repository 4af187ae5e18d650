// Shared host/device helpers for libtensoflow_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tensoflow_hip.h"

// ---------------------------------------------------------------- host-side error plumbing
void tf_set_error(const char* fmt, ...);

#define TF_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      tf_set_error(__VA_ARGS__);     \
      return code;                   \
    }                                \
  } while (0)

#define TF_LAUNCH_CHECK(name)                                                    \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      tf_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
      return TF_EHIP;                                                            \
    }                                                                            \
  } while (0)

// Per-DEVICE one-time set-up (kernel attributes are per device; a process may drive several).  `mask` is a function-local
// std::atomic<unsigned long long>: returns true exactly when the calling thread should run the set-up for the current device
// (racing first calls may both run it: the set-up calls are idempotent).  tf_once_done() marks it done.
#include <atomic>
static inline bool tf_once_needed(std::atomic<unsigned long long>& mask, int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  *dev_out = dev;
  return (mask.load(std::memory_order_acquire) & (1ull << dev)) == 0;
}
static inline void tf_once_done(std::atomic<unsigned long long>& mask, int dev) {
  mask.fetch_or(1ull << dev, std::memory_order_release);
}

static inline unsigned tf_blocks(int64_t n, int per_block) {
  return (unsigned)((n + per_block - 1) / per_block);
}

// ---------------------------------------------------------------- packed VM pyramid geometry
struct VmGeom {
  int C, n_levels;
  int ph[3], pw[3], ll[3];
  // float offsets of plane i level l / line i level l inside the packed buffer
  long long poff[3][4];
  long long loff[3][4];
  long long total;
  float aabb_lo[3], aabb_inv[3];  // contraction: (x - lo) * inv  -- see note in vm_field.hip
  float aabb_size[3];
  int texel_f16;                  // 1: `packed` holds halves (TfVmDesc.texel_f16)
};

static inline int vm_geom_init(const TfVmDesc* d, const float* aabb_host, VmGeom* g) {
  if (!d) return -1;
  if (d->C <= 0 || d->C % 4 != 0 || d->n_levels < 1 || d->n_levels > 4) return -1;
  g->C = d->C;
  g->n_levels = d->n_levels;
  g->texel_f16 = d->texel_f16 ? 1 : 0;
  long long off = 0;
  const int m = (1 << (d->n_levels - 1)) - 1;
  for (int i = 0; i < 3; ++i) {
    g->ph[i] = d->ph[i]; g->pw[i] = d->pw[i]; g->ll[i] = d->ll[i];
    if (d->ph[i] < 1 || d->pw[i] < 1 || d->ll[i] < 1) return -1;
    if ((d->ph[i] > 1 && (d->ph[i] & m)) || (d->pw[i] > 1 && (d->pw[i] & m)) || (d->ll[i] > 1 && (d->ll[i] & m)))
      return -2;
  }
  for (int i = 0; i < 3; ++i)
    for (int l = 0; l < 4; ++l) {
      g->poff[i][l] = off;
      if (l < d->n_levels) {
        int h = d->ph[i] >> l, w = d->pw[i] >> l;
        h = h < 1 ? 1 : h; w = w < 1 ? 1 : w;
        off += (long long)h * w * d->C;
      }
    }
  for (int i = 0; i < 3; ++i)
    for (int l = 0; l < 4; ++l) {
      g->loff[i][l] = off;
      if (l < d->n_levels) {
        int n = d->ll[i] >> l;
        n = n < 1 ? 1 : n;
        off += (long long)n * d->C;
      }
    }
  g->total = off;
  for (int k = 0; k < 3; ++k) {
    if (aabb_host) {
      g->aabb_lo[k] = aabb_host[k];
      g->aabb_size[k] = aabb_host[3 + k] - aabb_host[k];
      g->aabb_inv[k] = 1.0f / g->aabb_size[k];
    } else {
      g->aabb_lo[k] = 0.f; g->aabb_size[k] = 1.f; g->aabb_inv[k] = 1.f;
    }
  }
  return 0;
}

#ifdef __HIPCC__
__device__ __forceinline__ int vm_dim(int n, int l) { int v = n >> l; return v < 1 ? 1 : v; }

// Four consecutive channels of a texel at ELEMENT offset `off` of the packed pyramid (off % 4 == 0): 16 bytes of an fp32 pyramid,
// 8 bytes of a half pyramid widened to fp32 (the half pyramid keeps the element offsets of the fp32 one).
template <bool F16>
__device__ __forceinline__ float4 vm_texel4(const float* __restrict__ packed, long long off) {
  if (F16) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 v = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(packed) + off);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  return *reinterpret_cast<const float4*>(packed + off);
}
__device__ __forceinline__ float4 vm_texel4(const float* __restrict__ packed, long long off, int f16) {
  return f16 ? vm_texel4<true>(packed, off) : vm_texel4<false>(packed, off);
}

// Bilinear tap set on one axis of size n (texel-centre addressing, clamp): i0, i1, frac.
__device__ __forceinline__ void axis_taps(float t01, int n, int& i0, int& i1, float& f) {
  float u = t01 * (float)n - 0.5f;
  float fl = floorf(u);
  f = u - fl;
  int i = (int)fl;
  i0 = min(max(i, 0), n - 1);
  i1 = min(max(i + 1, 0), n - 1);
}

// Mip selection of dr.texture with mip_level_bias only: level clamped to [0, n_levels-1].
__device__ __forceinline__ void mip_select(float level, int n_levels, int& l0, int& l1, float& f) {
  float lv = fminf(fmaxf(level, 0.f), (float)(n_levels - 1));
  float fl = floorf(lv);
  l0 = (int)fl;
  f = lv - fl;
  l1 = min(l0 + 1, n_levels - 1);
  if (l1 == l0) f = 0.f;
}

__device__ __forceinline__ float softplus100(float x) {
  // torch.nn.Softplus(beta=100, threshold=20): log1p(exp(100 x)) / 100, and x itself where 100 x > 20.
  // Round 4: SEVEN instructions with TWO transcendentals -- v_mul, v_min, v_exp (2^a), v_add, v_log (log2), v_mul, v_med3:
  //     h = max(x, log2(1 + 2^min(100 x log2 e, 126)) ln 2 / 100).
  // The decoder applies it to 128 activations per lane per field evaluation, and a transcendental issues at a quarter of the vector
  // rate: rounds 1-3 spent 13 instructions with three of them (exp, log, rcp) on the classic log1p correction
  // log1p(t) = log(u) t / (u - 1), which keeps the RELATIVE error of tiny outputs -- of no use here: what the next layer needs is
  // the ABSOLUTE error of h, and rounding 1 + t costs 6e-8 in the logarithm's argument, i.e. <= 6e-10 in h (1 + t = 1 below
  // t = 6e-8 returns 0 for a true value under 6e-10).  max(x, .): softplus(x) >= x, the clamped exponent makes the formula fall
  // BELOW x once 100 x > 87 (no overflow to inf), and above the threshold x + log1p(e^-20) / 100 = x + 2e-11 rounds to x: the
  // branch of the reference comes out of the maximum.  One v_med3 with an opaque upper bound instead of v_max: an MFMA result is
  // not a known-canonical float to the compiler, which would canonicalise it first (see tf_relu).
  const float a = fminf(x * 144.269504088896341f, 126.f);
  const float l = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(a)) * 0.0069314718055994531f;
  float big = 3.0e38f;
  asm("" : "+s"(big));
  return __builtin_amdgcn_fmed3f(x, l, big);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
#endif
