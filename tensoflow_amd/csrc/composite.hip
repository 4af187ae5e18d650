// Packed-ray transmittance scan + accumulation (nerfacc.render_weight_from_alpha /
// accumulate_along_rays as used at network/shapeRenderer.py:1166-1206, :1249).
// One wave per ray: the ray's contiguous segment of the packed sample list is walked 64 samples at a
// time with a wave-level product scan; reads are fully coalesced, no atomics, deterministic.
#include "tf_common.h"

#define COMP_MAXK 8

__device__ __forceinline__ long long lower_bound_ll(const long long* __restrict__ a, long long n, long long key) {
  long long lo = 0, hi = n;
  while (lo < hi) {
    long long mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int K>
__global__ void __launch_bounds__(256) composite_fwd_kernel(const float* __restrict__ alpha,
                                                            const long long* __restrict__ ridx,
                                                            const float* __restrict__ values, long long n,
                                                            long long n_rays, float* __restrict__ weights,
                                                            float* __restrict__ acc, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const long long s0 = lower_bound_ll(ridx, n, ray), s1 = lower_bound_ll(ridx, n, ray + 1);
  float T = 1.f, a_sum = 0.f, v_sum[K > 0 ? K : 1];
#pragma unroll
  for (int k = 0; k < K; ++k) v_sum[k] = 0.f;
  for (long long base = s0; base < s1; base += 64) {
    const long long i = base + lane;
    const bool ok = i < s1;
    const float a = ok ? alpha[i] : 0.f;
    // inclusive product scan of (1 - a) across the wave
    float p = 1.f - a;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      float q = __shfl_up(p, o);
      if (lane >= o) p *= q;
    }
    float excl = __shfl_up(p, 1);
    if (lane == 0) excl = 1.f;
    const float w = a * (T * excl);
    if (ok) {
      weights[i] = w;
      a_sum += w;
#pragma unroll
      for (int k = 0; k < K; ++k) v_sum[k] += w * values[i * K + k];
    }
    T *= __shfl(p, 63);
  }
  a_sum = wave_sum(a_sum);
#pragma unroll
  for (int k = 0; k < K; ++k) v_sum[k] = wave_sum(v_sum[k]);
  if (lane == 0) {
    acc[ray] = a_sum;
#pragma unroll
    for (int k = 0; k < K; ++k) out[ray * K + k] = v_sum[k];
  }
}

// d/d alpha_i = gw_i*T_i - (sum_{j>i} gw_j*w_j) / (1-alpha_i);  d/d v_ik = w_i * g_out[r,k]
template <int K>
__global__ void __launch_bounds__(256) composite_bwd_kernel(const float* __restrict__ alpha,
                                                            const long long* __restrict__ ridx,
                                                            const float* __restrict__ values,
                                                            const float* __restrict__ weights,
                                                            const float* __restrict__ g_acc, const float* __restrict__ g_out,
                                                            long long n, long long n_rays, float* __restrict__ g_alpha,
                                                            float* __restrict__ g_values) {
  const int lane = threadIdx.x & 63;
  const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const long long s0 = lower_bound_ll(ridx, n, ray), s1 = lower_bound_ll(ridx, n, ray + 1);
  const float ga = g_acc ? g_acc[ray] : 0.f;
  float go[K > 0 ? K : 1];
#pragma unroll
  for (int k = 0; k < K; ++k) go[k] = g_out ? g_out[ray * K + k] : 0.f;
  // pass 1 (forward): stash the transmittance T_i in g_alpha[i] (each lane only touches its own i)
  if (g_alpha) {
    float T = 1.f;
    for (long long base = s0; base < s1; base += 64) {
      const long long i = base + lane;
      const bool ok = i < s1;
      float p = 1.f - (ok ? alpha[i] : 0.f);
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        float q = __shfl_up(p, o);
        if (lane >= o) p *= q;
      }
      float excl = __shfl_up(p, 1);
      if (lane == 0) excl = 1.f;
      if (ok) g_alpha[i] = T * excl;
      T *= __shfl(p, 63);
    }
  }
  float suffix = 0.f;  // sum over samples after the current chunk of gw_j * w_j
  const long long len = s1 - s0;
  const long long nchunks = (len + 63) / 64;
  for (long long c = nchunks - 1; c >= 0; --c) {
    const long long i = s0 + c * 64 + lane;
    const bool ok = i < s1;
    const float a = ok ? alpha[i] : 0.f;
    const float w = ok ? weights[i] : 0.f;
    float gw = ga;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (ok) {
        gw += go[k] * values[i * K + k];
        if (g_values) g_values[i * K + k] = w * go[k];
      }
    }
    float x = ok ? gw * w : 0.f;
    // inclusive suffix sum across the wave (lane l gets sum_{l' >= l} x)
    float s = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      float q = __shfl_down(s, o);
      if (lane + o < 64) s += q;
    }
    const float after = s - x + suffix;  // strictly after lane
    if (ok && g_alpha) {
      const float one_m = fmaxf(1.f - a, 1e-10f);
      const float Ti = g_alpha[i];
      g_alpha[i] = gw * Ti - after / one_m;
    }
    suffix += __shfl(s, 0);
  }
}

// exact T for tiny alpha in bwd: second variant recomputing T by a forward scan is used when requested
template <int K>
static int launch_fwd(const float* alpha, const long long* ridx, const float* values, long long n, long long n_rays,
                      float* weights, float* acc, float* out, hipStream_t stream) {
  composite_fwd_kernel<K><<<tf_blocks(n_rays, 4), 256, 0, stream>>>(alpha, ridx, values, n, n_rays, weights, acc, out);
  return 0;
}
template <int K>
static int launch_bwd(const float* alpha, const long long* ridx, const float* values, const float* weights,
                      const float* g_acc, const float* g_out, long long n, long long n_rays, float* g_alpha,
                      float* g_values, hipStream_t stream) {
  composite_bwd_kernel<K><<<tf_blocks(n_rays, 4), 256, 0, stream>>>(alpha, ridx, values, weights, g_acc, g_out, n, n_rays,
                                                                   g_alpha, g_values);
  return 0;
}

#define DISPATCH_K(k, CALL)                                                          \
  switch (k) {                                                                       \
    case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break;          \
    case 3: CALL(3); break; case 4: CALL(4); break; case 5: CALL(5); break;          \
    case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;          \
    default: tf_set_error("tf_composite: k=%d > %d unsupported", k, COMP_MAXK); return TF_ESHAPE; \
  }

extern "C" int tf_composite_fwd(const float* alpha, const int64_t* ray_indices, const float* values, int64_t n,
                                int64_t n_rays, int32_t k, float* weights, float* acc, float* out, tf_stream_t stream) {
  TF_REQUIRE(n >= 0 && n_rays >= 0 && k >= 0, TF_ESHAPE, "tf_composite_fwd: negative size");
  if (n_rays == 0) return TF_OK;
  TF_REQUIRE(acc && (k == 0 || out) && (n == 0 || (alpha && ray_indices && weights && (k == 0 || values))), TF_EINVAL,
             "tf_composite_fwd: null pointer");
#define CALL(KK) launch_fwd<KK>(alpha, (const long long*)ray_indices, values, n, n_rays, weights, acc, out, (hipStream_t)stream)
  DISPATCH_K(k, CALL)
#undef CALL
  TF_LAUNCH_CHECK("tf_composite_fwd");
  return TF_OK;
}

extern "C" int tf_composite_bwd(const float* alpha, const int64_t* ray_indices, const float* values, const float* weights,
                                const float* g_acc, const float* g_out, int64_t n, int64_t n_rays, int32_t k,
                                float* g_alpha, float* g_values, tf_stream_t stream) {
  TF_REQUIRE(n >= 0 && n_rays >= 0 && k >= 0, TF_ESHAPE, "tf_composite_bwd: negative size");
  if (n_rays == 0 || n == 0) return TF_OK;
  TF_REQUIRE(alpha && ray_indices && weights && (k == 0 || values), TF_EINVAL, "tf_composite_bwd: null pointer");
#define CALL(KK) launch_bwd<KK>(alpha, (const long long*)ray_indices, values, weights, g_acc, g_out, n, n_rays, g_alpha, g_values, (hipStream_t)stream)
  DISPATCH_K(k, CALL)
#undef CALL
  TF_LAUNCH_CHECK("tf_composite_bwd");
  return TF_OK;
}
