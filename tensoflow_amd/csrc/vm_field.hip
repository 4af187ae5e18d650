// VM-decomposed tensorial field: pyramid pack / unpack and the feature gather.
// Replaces dr.texture x6 + permute().contiguous() + per-call mip rebuild of
// network/fields.py:262-291, :776-806 and network/flow.py:709-738 (see include/tensoflow_hip.h).
//
// Data layout: channel-last packed pyramid (one texel = C*4 contiguous bytes), built once per
// optimizer step.  Gather: one lane per (point, 16-byte channel chunk): the 3*C/4 lanes of a point
// read each touched texel as one contiguous C*4-byte segment.
#include <algorithm>

#include "tf_common.h"
#include "tf_internal.h"

// ------------------------------------------------------------------------------------ pack fwd
// level 0: [C,H,W] -> [H,W,C].  One thread per pixel per pass over channels; transposed through LDS
// so both the global read (along W) and the global write (along C) are coalesced.
__global__ void __launch_bounds__(256) vm_pack_level0_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int C, long long npix) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [256][C+1]
  const long long p0 = (long long)blockIdx.x * 256;
  const int t = threadIdx.x;
  const int stride = C + 1;
  if (p0 + t < npix)
    for (int c = 0; c < C; ++c) tile[t * stride + c] = src[(long long)c * npix + p0 + t];
  __syncthreads();
  const long long nvalid = min((long long)256, npix - p0);
  for (long long e = t; e < nvalid * C; e += 256) {
    int pix = (int)(e / C), c = (int)(e % C);
    dst[p0 * C + e] = tile[pix * stride + c];
  }
}

// level l from level l-1 (channel-last): 2x2 box (2x1 when one axis has size 1).
__global__ void __launch_bounds__(256) vm_pack_down_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                           int hs, int ws, int hd, int wd) {
  long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)hd * wd * C;
  if (e >= total) return;
  int c = (int)(e % C);
  long long pix = e / C;
  int x = (int)(pix % wd), y = (int)(pix / wd);
  int sy = hs > 1 ? 2 : 1, sx = ws > 1 ? 2 : 1;
  float acc = 0.f;
  // same association as the oracle: average over H first, then over W
  if (sy == 2 && sx == 2) {
    float a = 0.5f * (src[((long long)(2 * y) * ws + 2 * x) * C + c] + src[((long long)(2 * y + 1) * ws + 2 * x) * C + c]);
    float b = 0.5f * (src[((long long)(2 * y) * ws + 2 * x + 1) * C + c] + src[((long long)(2 * y + 1) * ws + 2 * x + 1) * C + c]);
    acc = 0.5f * (a + b);
  } else if (sy == 2) {
    acc = 0.5f * (src[((long long)(2 * y) * ws + x) * C + c] + src[((long long)(2 * y + 1) * ws + x) * C + c]);
  } else if (sx == 2) {
    acc = 0.5f * (src[((long long)y * ws + 2 * x) * C + c] + src[((long long)y * ws + 2 * x + 1) * C + c]);
  } else {
    acc = src[((long long)y * ws + x) * C + c];
  }
  dst[e] = acc;
}

// ------------------------------------------------------------------------------------ pack bwd
// g[c,h,w] = sum_l  gp_l[h>>l, w>>l, c] / (box area at level l).  One thread per pixel, LDS transpose.
__global__ void __launch_bounds__(256) vm_unpack_kernel(const float* __restrict__ gp, float* __restrict__ dst, int C,
                                                        int H, int W, int n_levels, long long off0, long long off1,
                                                        long long off2, long long off3) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [256][C+1]
  const long long npix = (long long)H * W;
  const long long p0 = (long long)blockIdx.x * 256;
  const int t = threadIdx.x;
  const int stride = C + 1;
  const long long offs[4] = {off0, off1, off2, off3};
  const long long nvalid = min((long long)256, npix - p0);
  for (long long e = t; e < nvalid * C; e += 256) {
    int pix = (int)(e / C), c = (int)(e % C);
    long long p = p0 + pix;
    int y = (int)(p / W), x = (int)(p % W);
    float acc = 0.f, scale = 1.f;
    int h = H, w = W;
    for (int l = 0; l < n_levels; ++l) {
      int yl = H > 1 ? (y >> l) : 0, xl = W > 1 ? (x >> l) : 0;
      acc += scale * gp[offs[l] + ((long long)yl * w + xl) * C + c];
      if (h > 1) { h >>= 1; scale *= 0.5f; }
      if (w > 1) { w >>= 1; scale *= 0.5f; }
    }
    tile[pix * stride + c] = acc;
  }
  __syncthreads();
  if (p0 + t < npix)
    for (int c = 0; c < C; ++c) dst[(long long)c * npix + p0 + t] = tile[t * stride + c];
}

// ---- the same three kernels over ALL six arrays of a field (3 planes + 3 lines) in one launch: blockIdx.y = array.  A pyramid was 18
// launches to pack (6 x (level 0 + 2 mips)) and 6 to unpack; a training step packs 3-4 pyramids and unpacks as many.
struct VmArr { const float* src; float* dst; long long npix; int hs, ws, hd, wd, H, W; long long off[4]; };
struct VmArrs { VmArr a[6]; };

__global__ void __launch_bounds__(256) vm_pack_level0_multi_kernel(VmArrs A, int C) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [256][C+1]
  const VmArr& q = A.a[blockIdx.y];
  const long long p0 = (long long)blockIdx.x * 256;
  if (p0 >= q.npix) return;                                   // workgroup-uniform
  const int t = threadIdx.x;
  const int stride = C + 1;
  if (p0 + t < q.npix)
    for (int c = 0; c < C; ++c) tile[t * stride + c] = q.src[(long long)c * q.npix + p0 + t];
  __syncthreads();
  const long long nvalid = min((long long)256, q.npix - p0);
  for (long long e = t; e < nvalid * C; e += 256) {
    int pix = (int)(e / C), c = (int)(e % C);
    q.dst[p0 * C + e] = tile[pix * stride + c];
  }
}

__global__ void __launch_bounds__(256) vm_pack_down_multi_kernel(VmArrs A, int C) {
  const VmArr& q = A.a[blockIdx.y];
  const float* __restrict__ src = q.src;
  const int hs = q.hs, ws = q.ws, hd = q.hd, wd = q.wd;
  long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)hd * wd * C;
  if (e >= total) return;
  int c = (int)(e % C);
  long long pix = e / C;
  int x = (int)(pix % wd), y = (int)(pix / wd);
  int sy = hs > 1 ? 2 : 1, sx = ws > 1 ? 2 : 1;
  float acc = 0.f;
  // same association as the oracle (and as vm_pack_down_kernel): average over H first, then over W
  if (sy == 2 && sx == 2) {
    float a = 0.5f * (src[((long long)(2 * y) * ws + 2 * x) * C + c] + src[((long long)(2 * y + 1) * ws + 2 * x) * C + c]);
    float b = 0.5f * (src[((long long)(2 * y) * ws + 2 * x + 1) * C + c] + src[((long long)(2 * y + 1) * ws + 2 * x + 1) * C + c]);
    acc = 0.5f * (a + b);
  } else if (sy == 2) {
    acc = 0.5f * (src[((long long)(2 * y) * ws + x) * C + c] + src[((long long)(2 * y + 1) * ws + x) * C + c]);
  } else if (sx == 2) {
    acc = 0.5f * (src[((long long)y * ws + 2 * x) * C + c] + src[((long long)y * ws + 2 * x + 1) * C + c]);
  } else {
    acc = src[((long long)y * ws + x) * C + c];
  }
  q.dst[e] = acc;
}

__global__ void __launch_bounds__(256) vm_unpack_multi_kernel(VmArrs A, const float* __restrict__ gp, int C, int n_levels) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [256][C+1]
  const VmArr& q = A.a[blockIdx.y];
  const int H = q.H, W = q.W;
  const long long npix = (long long)H * W;
  const long long p0 = (long long)blockIdx.x * 256;
  if (p0 >= npix) return;                                     // workgroup-uniform
  const int t = threadIdx.x;
  const int stride = C + 1;
  const long long nvalid = min((long long)256, npix - p0);
  for (long long e = t; e < nvalid * C; e += 256) {
    int pix = (int)(e / C), c = (int)(e % C);
    long long p = p0 + pix;
    int y = (int)(p / W), x = (int)(p % W);
    float acc = 0.f, scale = 1.f;
    int h = H, w = W;
    for (int l = 0; l < n_levels; ++l) {
      int yl = H > 1 ? (y >> l) : 0, xl = W > 1 ? (x >> l) : 0;
      acc += scale * gp[q.off[l] + ((long long)yl * w + xl) * C + c];
      if (h > 1) { h >>= 1; scale *= 0.5f; }
      if (w > 1) { w >>= 1; scale *= 0.5f; }
    }
    tile[pix * stride + c] = acc;
  }
  __syncthreads();
  if (p0 + t < npix)
    for (int c = 0; c < C; ++c) q.dst[(long long)c * npix + p0 + t] = tile[t * stride + c];
}

// ------------------------------------------------------------------------------------ gather
__device__ __forceinline__ float4 lerp4(float4 a, float4 b, float t) {
  const float s = 1.f - t;
  return make_float4(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t, a.w * s + b.w * t);
}

template <bool BWD>
__global__ void __launch_bounds__(256) vm_gather_kernel(VmGeom g, const float* __restrict__ packed,
                                                        const float* __restrict__ xyz, const float* __restrict__ level,
                                                        long long n, const float* __restrict__ gfeat,
                                                        float* __restrict__ out) {
  // one lane per (point, float4 chunk); chunks per point = 3*C/4
  const int cpp = 3 * g.C / 4;
  const int cpl = g.C / 4;
  long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * cpp) return;
  long long pt = e / cpp;
  int q = (int)(e % cpp);
  int i = q / cpl, j = q % cpl;
  float p[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = (xyz[pt * 3 + k] - g.aabb_lo[k]) / g.aabb_size[k];
  const int m0 = i == 2 ? 1 : 0, m1 = i == 0 ? 1 : 2, vm = 2 - i;
  float u = p[m0], v = p[m1], wv = p[vm];
  int l0, l1;
  float fl;
  mip_select(level ? level[pt] : 0.f, g.n_levels, l0, l1, fl);
  float4 pv = make_float4(0, 0, 0, 0), lv = make_float4(0, 0, 0, 0);
  float4 gsum = make_float4(0, 0, 0, 0);
  if (BWD) gsum = reinterpret_cast<const float4*>(gfeat)[e];
  // two passes in BWD: first recompute pv, lv; then scatter.  FWD: single pass.
  for (int pass = 0; pass < (BWD ? 2 : 1); ++pass) {
    for (int li = 0; li < 2; ++li) {
      int l = li ? l1 : l0;
      float wl = li ? fl : 1.f - fl;
      if (li && fl == 0.f) break;
      int H = vm_dim(g.ph[i], l), W = vm_dim(g.pw[i], l), L = vm_dim(g.ll[i], l);
      int x0, x1, y0, y1, z0, z1;
      float fx, fy, fz;
      axis_taps(u, W, x0, x1, fx);
      axis_taps(v, H, y0, y1, fy);
      axis_taps(wv, L, z0, z1, fz);
      const long long pb = g.poff[i][l], lb = g.loff[i][l];
      const long long a00 = pb + ((long long)y0 * W + x0) * g.C + 4 * j, a10 = pb + ((long long)y0 * W + x1) * g.C + 4 * j;
      const long long a01 = pb + ((long long)y1 * W + x0) * g.C + 4 * j, a11 = pb + ((long long)y1 * W + x1) * g.C + 4 * j;
      const long long b0 = lb + (long long)z0 * g.C + 4 * j, b1 = lb + (long long)z1 * g.C + 4 * j;
      if (pass == 0) {
        const int f16 = BWD ? 0 : g.texel_f16;          // wave-uniform
        float4 t00 = vm_texel4(packed, a00, f16), t10 = vm_texel4(packed, a10, f16);
        float4 t01 = vm_texel4(packed, a01, f16), t11 = vm_texel4(packed, a11, f16);
        float4 s0 = vm_texel4(packed, b0, f16), s1 = vm_texel4(packed, b1, f16);
        float4 top = lerp4(t00, t10, fx), bot = lerp4(t01, t11, fx);
        float4 pl = lerp4(top, bot, fy), ln = lerp4(s0, s1, fz);
        pv.x += wl * pl.x; pv.y += wl * pl.y; pv.z += wl * pl.z; pv.w += wl * pl.w;
        lv.x += wl * ln.x; lv.y += wl * ln.y; lv.z += wl * ln.z; lv.w += wl * ln.w;
      } else {
        // d feat = gsum;  d plane-sample = gsum * lv ; d line-sample = gsum * pv
        float4 gp_ = make_float4(gsum.x * lv.x * wl, gsum.y * lv.y * wl, gsum.z * lv.z * wl, gsum.w * lv.w * wl);
        float4 gl_ = make_float4(gsum.x * pv.x * wl, gsum.y * pv.y * wl, gsum.z * pv.z * wl, gsum.w * pv.w * wl);
#define ATOM4(addr, G, wt_) atomicAdd(out + (addr), G.x * (wt_)); atomicAdd(out + (addr) + 1, G.y * (wt_)); atomicAdd(out + (addr) + 2, G.z * (wt_)); atomicAdd(out + (addr) + 3, G.w * (wt_));
        ATOM4(a00, gp_, (1.f - fx) * (1.f - fy)) ATOM4(a10, gp_, fx * (1.f - fy))
        ATOM4(a01, gp_, (1.f - fx) * fy) ATOM4(a11, gp_, fx * fy)
        ATOM4(b0, gl_, 1.f - fz) ATOM4(b1, gl_, fz)
      }
    }
  }
  if (!BWD) {
    float4 f = make_float4(pv.x * lv.x, pv.y * lv.y, pv.z * lv.z, pv.w * lv.w);
    reinterpret_cast<float4*>(out)[e] = f;
  }
}

// Backward of the gather with ONE CHANNEL per lane: the C channels of a texel are C consecutive floats of the channel-last pyramid, so
// the C lanes of a (point, plane) add to C consecutive words with one atomic instruction per texel -- 16 words per 64-byte atomic
// request instead of the 4 the float4-chunk mapping of the forward kernel gave (its four per-component atomics each touched every
// fourth word).  The gradient buffer (51 MB at R = 300) lives behind the L2: the scatter is bound by atomic REQUESTS.
// Rows beyond the first n_pts are FINITE-DIFFERENCE TAPS of the same points (tf_sdf_alpha_bwd): row r belongs to point r % n_pts and
// tap r / n_pts (0: the point itself; 1 + 2 ax / 2 + 2 ax: +- tap_units[ax] along axis ax, fields.py:234-256); gfeat rows have stride ld.
struct VmTaps { long long n_pts; int ld; float units[3]; };
__global__ void __launch_bounds__(256) vm_scatter_kernel(VmGeom g, const float* __restrict__ packed, const float* __restrict__ xyz,
                                                         const float* __restrict__ level, long long n, const float* __restrict__ gfeat,
                                                         float* __restrict__ out, VmTaps T) {
  const int cpp = 3 * g.C;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * cpp) return;
  const long long row = e / cpp;
  const long long pt = row % T.n_pts;
  const int tap = (int)(row / T.n_pts);
  const int q = (int)(e % cpp);
  const int i = q / g.C, c = q % g.C;
  float p[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float xk = xyz[pt * 3 + k];
    if (tap > 0 && (tap - 1) / 2 == k) xk += (tap & 1) ? T.units[k] : -T.units[k];      // x + u, x - u exactly as the forward kernel forms them
    p[k] = (xk - g.aabb_lo[k]) / g.aabb_size[k];
  }
  const int m0 = i == 2 ? 1 : 0, m1 = i == 0 ? 1 : 2, vm = 2 - i;
  const float u = p[m0], v = p[m1], wv = p[vm];
  int l0, l1;
  float fl;
  mip_select(level ? level[pt] : 0.f, g.n_levels, l0, l1, fl);
  const float gs = gfeat[row * T.ld + q];
  float pv = 0.f, lv = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    for (int li = 0; li < 2; ++li) {
      const int l = li ? l1 : l0;
      const float wl = li ? fl : 1.f - fl;
      if (li && fl == 0.f) break;
      const int H = vm_dim(g.ph[i], l), W = vm_dim(g.pw[i], l), L = vm_dim(g.ll[i], l);
      int x0, x1, y0, y1, z0, z1;
      float fx, fy, fz;
      axis_taps(u, W, x0, x1, fx);
      axis_taps(v, H, y0, y1, fy);
      axis_taps(wv, L, z0, z1, fz);
      const long long pb = g.poff[i][l], lb = g.loff[i][l];
      const long long a00 = pb + ((long long)y0 * W + x0) * g.C + c, a10 = pb + ((long long)y0 * W + x1) * g.C + c;
      const long long a01 = pb + ((long long)y1 * W + x0) * g.C + c, a11 = pb + ((long long)y1 * W + x1) * g.C + c;
      const long long b0 = lb + (long long)z0 * g.C + c, b1 = lb + (long long)z1 * g.C + c;
      if (pass == 0) {       // same expressions as the forward kernel's lerp4 (a * (1 - t) + b * t), per channel
        const float top = packed[a00] * (1.f - fx) + packed[a10] * fx, bot = packed[a01] * (1.f - fx) + packed[a11] * fx;
        pv += wl * (top * (1.f - fy) + bot * fy);
        lv += wl * (packed[b0] * (1.f - fz) + packed[b1] * fz);
      } else {
        const float gp_ = gs * lv * wl, gl_ = gs * pv * wl;
        atomicAdd(out + a00, gp_ * ((1.f - fx) * (1.f - fy))); atomicAdd(out + a10, gp_ * (fx * (1.f - fy)));
        atomicAdd(out + a01, gp_ * ((1.f - fx) * fy)); atomicAdd(out + a11, gp_ * (fx * fy));
        atomicAdd(out + b0, gl_ * (1.f - fz)); atomicAdd(out + b1, gl_ * fz);
      }
    }
  }
}

// ---- the 7-tap scatter with the taps of a point MERGED before they leave the lane.  The finite-difference taps sit one texel pitch
// apart (units = aabbSize / (R - 1), texel pitch aabbSize / R: 1.003 texels at level 0, half of that per mip level), so their
// bilinear footprints overlap: on plane i the taps along the LINE axis share the centre's four texels outright, the taps along an
// in-plane axis land inside a window of six texels around the centre's pair, and on the line the five taps that do not move along it
// share the centre's two texels.  One lane = (point, plane, channel): it accumulates the seven taps' contributions per UNIQUE texel in
// registers (windows indexed relative to the centre's floor index) and issues one atomic per texel that received anything: ~16
// atomic instructions per level instead of 42 -- the scatter is bound by atomic REQUESTS behind the L2.
struct AxisTap { int i; float f; };      // UNCLAMPED floor index and fraction (axis_taps clamps i, i + 1 into [0, n - 1])
__device__ __forceinline__ AxisTap axis_tap_raw(float t01, int n) {
  const float u = t01 * (float)n - 0.5f;
  const float fl = floorf(u);
  AxisTap a; a.i = (int)fl; a.f = u - fl;
  return a;
}
__device__ __forceinline__ int clampi(int v, int n) { return min(max(v, 0), n - 1); }
__device__ __forceinline__ float pick6(const float (&a)[6], int r) {
  float v = a[0];
#pragma unroll
  for (int k = 1; k < 6; ++k) v = (r == k) ? a[k] : v;
  return v;
}
__device__ __forceinline__ void add6(float (&a)[6], int r, float v) {
#pragma unroll
  for (int k = 0; k < 6; ++k) a[k] += (r == k) ? v : 0.f;
}

__global__ void __launch_bounds__(256) vm_scatter7_kernel(VmGeom g, const float* __restrict__ packed, const float* __restrict__ xyz,
                                                          const float* __restrict__ level, long long n_pts, const float* __restrict__ gfeat,
                                                          float* __restrict__ out, VmTaps T) {
  const int cpp = 3 * g.C;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_pts * cpp) return;
  const long long pt = e / cpp;
  const int q = (int)(e % cpp);
  const int i = q / g.C, c = q % g.C;
  const int m0 = i == 2 ? 1 : 0, m1 = i == 0 ? 1 : 2, vm = 2 - i;       // world axes of the plane's x, y and of the line
  // normalised coordinates of the centre and of the +- taps per axis, formed exactly as the forward kernel forms them ((x +- u) - lo) / size
  float pc[3], pp[3], pm[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float xk = xyz[pt * 3 + k];
    pc[k] = (xk - g.aabb_lo[k]) / g.aabb_size[k];
    pp[k] = ((xk + T.units[k]) - g.aabb_lo[k]) / g.aabb_size[k];
    pm[k] = ((xk - T.units[k]) - g.aabb_lo[k]) / g.aabb_size[k];
  }
  int l0, l1;
  float fl;
  mip_select(level ? level[pt] : 0.f, g.n_levels, l0, l1, fl);
  // upstream gradients of this (plane, channel) for the seven taps: tap 0 centre, 1 + 2 k / 2 + 2 k = +- along world axis k
  float gs[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) gs[t] = gfeat[((long long)t * n_pts + pt) * T.ld + q];
  // taps of this plane by ROLE: 0 centre, 1 / 2 = + / - along the plane's x, 3 / 4 = + / - along its y, 5 / 6 = + / - along the line
  const float gr[7] = {gs[0], gs[1 + 2 * m0], gs[2 + 2 * m0], gs[1 + 2 * m1], gs[2 + 2 * m1], gs[1 + 2 * vm], gs[2 + 2 * vm]};
  float pv[7] = {0, 0, 0, 0, 0, 0, 0}, lv[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    for (int li = 0; li < 2; ++li) {
      if (li && fl == 0.f) break;
      const int l = li ? l1 : l0;
      const float wl = li ? fl : 1.f - fl;
      const int H = vm_dim(g.ph[i], l), W = vm_dim(g.pw[i], l), L = vm_dim(g.ll[i], l);
      const long long pb = g.poff[i][l] + c, lb = g.loff[i][l] + c;
      const AxisTap xc = axis_tap_raw(pc[m0], W), xp = axis_tap_raw(pp[m0], W), xm = axis_tap_raw(pm[m0], W);
      const AxisTap yc = axis_tap_raw(pc[m1], H), yp = axis_tap_raw(pp[m1], H), ym = axis_tap_raw(pm[m1], H);
      const AxisTap zc = axis_tap_raw(pc[vm], L), zp = axis_tap_raw(pp[vm], L), zm = axis_tap_raw(pm[vm], L);
      // window slots: index - (centre index - 2) in [0, 5]; a tap further than two texels from the centre (cannot happen while
      // units = size / (R - 1)) is clamped into the window's edge -- caught by the host-side check of tf_vm_scatter_taps
      const int rxp = min(max(xp.i - xc.i + 2, 0), 4), rxm = min(max(xm.i - xc.i + 2, 0), 4);
      const int ryp = min(max(yp.i - yc.i + 2, 0), 4), rym = min(max(ym.i - yc.i + 2, 0), 4);
      const int rzp = min(max(zp.i - zc.i + 2, 0), 4), rzm = min(max(zm.i - zc.i + 2, 0), 4);
      const int y0 = clampi(yc.i, H), y1 = clampi(yc.i + 1, H), x0 = clampi(xc.i, W), x1 = clampi(xc.i + 1, W);
      if (pass == 0) {
        // texel values of the three windows (rows y0 / y1 over six columns; columns x0 / x1 over six rows; six line texels)
        float PX0[6], PX1[6], PY0[6], PY1[6], LZ[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const int xr = clampi(xc.i - 2 + r, W), yr = clampi(yc.i - 2 + r, H), zr = clampi(zc.i - 2 + r, L);
          PX0[r] = packed[pb + ((long long)y0 * W + xr) * g.C];
          PX1[r] = packed[pb + ((long long)y1 * W + xr) * g.C];
          PY0[r] = packed[pb + ((long long)yr * W + x0) * g.C];
          PY1[r] = packed[pb + ((long long)yr * W + x1) * g.C];
          LZ[r] = packed[lb + (long long)zr * g.C];
        }
        auto bil_x = [&](int r, float fx) {      // plane value of a tap displaced along x: columns r, r + 1 of both rows, the centre's fy
          const float top = pick6(PX0, r) * (1.f - fx) + pick6(PX0, r + 1) * fx, bot = pick6(PX1, r) * (1.f - fx) + pick6(PX1, r + 1) * fx;
          return top * (1.f - yc.f) + bot * yc.f;
        };
        auto bil_y = [&](int r, float fy) {      // displaced along y: rows r, r + 1 of both columns, the centre's fx
          const float top = pick6(PY0, r) * (1.f - xc.f) + pick6(PY1, r) * xc.f, bot = pick6(PY0, r + 1) * (1.f - xc.f) + pick6(PY1, r + 1) * xc.f;
          return top * (1.f - fy) + bot * fy;
        };
        auto lin_z = [&](int r, float fz) { return pick6(LZ, r) * (1.f - fz) + pick6(LZ, r + 1) * fz; };
        const float pcv = bil_x(2, xc.f), lcv = lin_z(2, zc.f);
        pv[0] += wl * pcv; lv[0] += wl * lcv;
        pv[1] += wl * bil_x(rxp, xp.f); lv[1] += wl * lcv;
        pv[2] += wl * bil_x(rxm, xm.f); lv[2] += wl * lcv;
        pv[3] += wl * bil_y(ryp, yp.f); lv[3] += wl * lcv;
        pv[4] += wl * bil_y(rym, ym.f); lv[4] += wl * lcv;
        pv[5] += wl * pcv; lv[5] += wl * lin_z(rzp, zp.f);
        pv[6] += wl * pcv; lv[6] += wl * lin_z(rzm, zm.f);
      } else {
        float AX0[6] = {0, 0, 0, 0, 0, 0}, AX1[6] = {0, 0, 0, 0, 0, 0}, AY0[6] = {0, 0, 0, 0, 0, 0}, AY1[6] = {0, 0, 0, 0, 0, 0},
              AZ[6] = {0, 0, 0, 0, 0, 0};
        // plane adjoints gP_t = g_t lv_t wl, line adjoints gL_t = g_t pv_t wl
        float gP[7], gL[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) { gP[t] = gr[t] * lv[t] * wl; gL[t] = gr[t] * pv[t] * wl; }
        // centre and the two taps along the line: the centre's four texels (x window slots 2, 3; rows y0, y1)
        const float gc = gP[0] + gP[5] + gP[6];
        AX0[2] += gc * ((1.f - xc.f) * (1.f - yc.f)); AX0[3] += gc * (xc.f * (1.f - yc.f));
        AX1[2] += gc * ((1.f - xc.f) * yc.f); AX1[3] += gc * (xc.f * yc.f);
        // taps along x: columns r, r + 1 of rows y0, y1
        add6(AX0, rxp, gP[1] * ((1.f - xp.f) * (1.f - yc.f))); add6(AX0, rxp + 1, gP[1] * (xp.f * (1.f - yc.f)));
        add6(AX1, rxp, gP[1] * ((1.f - xp.f) * yc.f)); add6(AX1, rxp + 1, gP[1] * (xp.f * yc.f));
        add6(AX0, rxm, gP[2] * ((1.f - xm.f) * (1.f - yc.f))); add6(AX0, rxm + 1, gP[2] * (xm.f * (1.f - yc.f)));
        add6(AX1, rxm, gP[2] * ((1.f - xm.f) * yc.f)); add6(AX1, rxm + 1, gP[2] * (xm.f * yc.f));
        // taps along y: rows r, r + 1 of columns x0, x1
        add6(AY0, ryp, gP[3] * ((1.f - xc.f) * (1.f - yp.f))); add6(AY0, ryp + 1, gP[3] * ((1.f - xc.f) * yp.f));
        add6(AY1, ryp, gP[3] * (xc.f * (1.f - yp.f))); add6(AY1, ryp + 1, gP[3] * (xc.f * yp.f));
        add6(AY0, rym, gP[4] * ((1.f - xc.f) * (1.f - ym.f))); add6(AY0, rym + 1, gP[4] * ((1.f - xc.f) * ym.f));
        add6(AY1, rym, gP[4] * (xc.f * (1.f - ym.f))); add6(AY1, rym + 1, gP[4] * (xc.f * ym.f));
        // the y window's rows 2, 3 ARE the centre's rows y0, y1: fold them into the x window's columns 2, 3
        AX0[2] += AY0[2]; AX1[2] += AY0[3]; AX0[3] += AY1[2]; AX1[3] += AY1[3];
        AY0[2] = AY0[3] = AY1[2] = AY1[3] = 0.f;
        // line: the five taps that do not move along it share the centre's two texels
        const float glc = gL[0] + gL[1] + gL[2] + gL[3] + gL[4];
        AZ[2] += glc * (1.f - zc.f); AZ[3] += glc * zc.f;
        add6(AZ, rzp, gL[5] * (1.f - zp.f)); add6(AZ, rzp + 1, gL[5] * zp.f);
        add6(AZ, rzm, gL[6] * (1.f - zm.f)); add6(AZ, rzm + 1, gL[6] * zm.f);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const int xr = clampi(xc.i - 2 + r, W), yr = clampi(yc.i - 2 + r, H), zr = clampi(zc.i - 2 + r, L);
          if (AX0[r] != 0.f) atomicAdd(out + pb + ((long long)y0 * W + xr) * g.C, AX0[r]);
          if (AX1[r] != 0.f) atomicAdd(out + pb + ((long long)y1 * W + xr) * g.C, AX1[r]);
          if (AY0[r] != 0.f) atomicAdd(out + pb + ((long long)yr * W + x0) * g.C, AY0[r]);
          if (AY1[r] != 0.f) atomicAdd(out + pb + ((long long)yr * W + x1) * g.C, AY1[r]);
          if (AZ[r] != 0.f) atomicAdd(out + lb + (long long)zr * g.C, AZ[r]);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) vm_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long long n) {
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(src + e);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 o; o[0] = (_Float16)v.x; o[1] = (_Float16)v.y; o[2] = (_Float16)v.z; o[3] = (_Float16)v.w;
    *reinterpret_cast<h4*>(dst + e) = o;
  } else {
    for (long long k = e; k < n; ++k) dst[k] = (_Float16)src[k];
  }
}

// ------------------------------------------------------------------------------------ C ABI
extern "C" size_t tf_vm_packed_floats(const TfVmDesc* d) {
  VmGeom g;
  if (vm_geom_init(d, nullptr, &g) != 0) return 0;
  return (size_t)g.total;
}

static int check_geom(const TfVmDesc* d, const float* aabb, VmGeom* g, const char* who) {
  int rc = vm_geom_init(d, aabb, g);
  TF_REQUIRE(rc != -1, TF_EINVAL, "%s: bad TfVmDesc (C must be a positive multiple of 4, 1<=n_levels<=4, sizes>=1)", who);
  TF_REQUIRE(rc != -2, TF_ESHAPE, "%s: plane/line sizes > 1 must be divisible by 2^(n_levels-1)", who);
  return 0;
}

extern "C" int tf_vm_pack_fwd(const TfVmDesc* d, const float* const planes[3], const float* const lines[3],
                              float* packed, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VmGeom g;
  if (int rc = check_geom(d, nullptr, &g, "tf_vm_pack_fwd")) return rc;
  TF_REQUIRE(planes && lines && packed, TF_EINVAL, "tf_vm_pack_fwd: null pointer");
  const size_t lds = 256 * (g.C + 1) * sizeof(float);
  // level 0 of all six arrays in one launch, then one launch per mip level (level l reads level l - 1): n_levels launches instead of
  // 6 n_levels
  VmArrs A = {};
  long long max_pix = 0;
  for (int i = 0; i < 3; ++i) {
    TF_REQUIRE(planes[i] && lines[i], TF_EINVAL, "tf_vm_pack_fwd: null plane/line %d", i);
    A.a[2 * i] = VmArr{planes[i], packed + g.poff[i][0], (long long)g.ph[i] * g.pw[i]};
    A.a[2 * i + 1] = VmArr{lines[i], packed + g.loff[i][0], (long long)g.ll[i]};
    max_pix = std::max(max_pix, std::max(A.a[2 * i].npix, A.a[2 * i + 1].npix));
  }
  vm_pack_level0_multi_kernel<<<dim3(tf_blocks(max_pix, 256), 6), 256, lds, stream>>>(A, g.C);
  for (int l = 1; l < g.n_levels; ++l) {
    VmArrs D = {};
    long long max_el = 0;
    for (int i = 0; i < 3; ++i) {
      int hs = g.ph[i] >> (l - 1), ws = g.pw[i] >> (l - 1), hd = g.ph[i] >> l, wd = g.pw[i] >> l;
      hs = hs < 1 ? 1 : hs; ws = ws < 1 ? 1 : ws; hd = hd < 1 ? 1 : hd; wd = wd < 1 ? 1 : wd;
      D.a[2 * i] = VmArr{packed + g.poff[i][l - 1], packed + g.poff[i][l], 0, hs, ws, hd, wd};
      int ls = g.ll[i] >> (l - 1), ld = g.ll[i] >> l;
      ls = ls < 1 ? 1 : ls; ld = ld < 1 ? 1 : ld;
      D.a[2 * i + 1] = VmArr{packed + g.loff[i][l - 1], packed + g.loff[i][l], 0, ls, 1, ld, 1};
      max_el = std::max(max_el, std::max((long long)hd * wd * g.C, (long long)ld * g.C));
    }
    vm_pack_down_multi_kernel<<<dim3(tf_blocks(max_el, 256), 6), 256, 0, stream>>>(D, g.C);
  }
  TF_LAUNCH_CHECK("tf_vm_pack_fwd");
  return TF_OK;
}

extern "C" int tf_vm_pack_to_f16(const TfVmDesc* d, const float* packed, void* packed16, tf_stream_t stream_) {
  VmGeom g;
  if (int rc = check_geom(d, nullptr, &g, "tf_vm_pack_to_f16")) return rc;
  TF_REQUIRE(packed && packed16, TF_EINVAL, "tf_vm_pack_to_f16: null pointer");
  TF_REQUIRE((((uintptr_t)packed | (uintptr_t)packed16) & 15) == 0, TF_EINVAL, "tf_vm_pack_to_f16: buffers must be 16-byte aligned");
  vm_to_f16_kernel<<<tf_blocks((g.total + 3) / 4, 256), 256, 0, (hipStream_t)stream_>>>(packed, (_Float16*)packed16, g.total);
  TF_LAUNCH_CHECK("tf_vm_pack_to_f16");
  return TF_OK;
}

extern "C" int tf_vm_pack_bwd(const TfVmDesc* d, const float* gpacked, float* const gplanes[3], float* const glines[3],
                              tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VmGeom g;
  if (int rc = check_geom(d, nullptr, &g, "tf_vm_pack_bwd")) return rc;
  TF_REQUIRE(gpacked && gplanes && glines, TF_EINVAL, "tf_vm_pack_bwd: null pointer");
  const size_t lds = 256 * (g.C + 1) * sizeof(float);
  VmArrs A = {};
  long long max_pix = 0;
  for (int i = 0; i < 3; ++i) {
    TF_REQUIRE(gplanes[i] && glines[i], TF_EINVAL, "tf_vm_pack_bwd: null plane/line %d", i);
    A.a[2 * i] = VmArr{nullptr, gplanes[i], 0, 0, 0, 0, 0, g.ph[i], g.pw[i], {g.poff[i][0], g.poff[i][1], g.poff[i][2], g.poff[i][3]}};
    A.a[2 * i + 1] = VmArr{nullptr, glines[i], 0, 0, 0, 0, 0, g.ll[i], 1, {g.loff[i][0], g.loff[i][1], g.loff[i][2], g.loff[i][3]}};
    max_pix = std::max(max_pix, std::max((long long)g.ph[i] * g.pw[i], (long long)g.ll[i]));
  }
  vm_unpack_multi_kernel<<<dim3(tf_blocks(max_pix, 256), 6), 256, lds, stream>>>(A, gpacked, g.C, g.n_levels);
  TF_LAUNCH_CHECK("tf_vm_pack_bwd");
  return TF_OK;
}

extern "C" int tf_vm_gather_fwd(const TfVmDesc* d, const float* packed, const float* xyz, const float* level,
                                const float* aabb_host, int64_t n, float* feat, tf_stream_t stream) {
  VmGeom g;
  TF_REQUIRE(aabb_host, TF_EINVAL, "tf_vm_gather_fwd: aabb_host is null");
  if (int rc = check_geom(d, aabb_host, &g, "tf_vm_gather_fwd")) return rc;
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_vm_gather_fwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(packed && xyz && feat, TF_EINVAL, "tf_vm_gather_fwd: null pointer");
  long long work = (long long)n * (3 * g.C / 4);
  vm_gather_kernel<false><<<tf_blocks(work, 256), 256, 0, (hipStream_t)stream>>>(g, packed, xyz, level, n, nullptr, feat);
  TF_LAUNCH_CHECK("tf_vm_gather_fwd");
  return TF_OK;
}

extern "C" int tf_vm_gather_bwd(const TfVmDesc* d, const float* packed, const float* xyz, const float* level,
                                const float* aabb_host, int64_t n, const float* gfeat, float* gpacked,
                                tf_stream_t stream) {
  VmGeom g;
  TF_REQUIRE(aabb_host, TF_EINVAL, "tf_vm_gather_bwd: aabb_host is null");
  if (int rc = check_geom(d, aabb_host, &g, "tf_vm_gather_bwd")) return rc;
  TF_REQUIRE(!g.texel_f16, TF_EINVAL, "tf_vm_gather_bwd: the adjoint takes an fp32 pyramid (texel_f16 is an inference-only format)");
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_vm_gather_bwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(packed && xyz && gfeat && gpacked, TF_EINVAL, "tf_vm_gather_bwd: null pointer");
#ifdef VM_SCATTER_CHUNKS      // dev-only switch: the float4-chunk mapping of the forward kernel
  long long work = (long long)n * (3 * g.C / 4);
  vm_gather_kernel<true><<<tf_blocks(work, 256), 256, 0, (hipStream_t)stream>>>(g, packed, xyz, level, n, gfeat, gpacked);
#else
  long long work = (long long)n * (3 * g.C);
  VmTaps T{n, 3 * g.C, {0.f, 0.f, 0.f}};
  vm_scatter_kernel<<<tf_blocks(work, 256), 256, 0, (hipStream_t)stream>>>(g, packed, xyz, level, n, gfeat, gpacked, T);
#endif
  TF_LAUNCH_CHECK("tf_vm_gather_bwd");
  return TF_OK;
}


// Internal (tf_internal.h): the scatter of tf_vm_gather_bwd for the 7 finite-difference taps of n_pts points in one launch --
// gfeat [7 n_pts, ld] (row = tap * n_pts + point, the first 3C columns are read), gpacked += .
int tf_vm_scatter_taps(const VmGeom& g, const float* packed, const float* pts, const float* level, long long n_pts, const float* units,
                       const float* gfeat, int ld, float* gpacked, hipStream_t stream) {
  if (n_pts == 0) return TF_OK;
  VmTaps T{n_pts, ld, {units[0], units[1], units[2]}};
  // the merged kernel's six-texel windows hold taps at most two texels from the centre: units[k] <= 1.5 texel pitches of the level-0 grid
  // along axis k (the reference's units are size / (R - 1) = R / (R - 1) pitches).  Anything else takes the tap-per-lane kernel.
  bool merged = true;
  for (int i = 0; i < 3; ++i) {
    const int m0 = i == 2 ? 1 : 0, m1 = i == 0 ? 1 : 2, vm = 2 - i;
    merged = merged && units[m0] * g.pw[i] <= 1.5f * g.aabb_size[m0] && units[m1] * g.ph[i] <= 1.5f * g.aabb_size[m1] &&
             units[vm] * g.ll[i] <= 1.5f * g.aabb_size[vm] && units[m0] > 0.f && units[m1] > 0.f && units[vm] > 0.f;
  }
#ifdef VM_SCATTER_TAP_PER_LANE      // dev-only switch (A/B): round 4's first form, one lane per (tap, point, plane, channel)
  merged = false;
#endif
  if (merged) {
    const long long work = n_pts * (3 * g.C);
    vm_scatter7_kernel<<<tf_blocks(work, 256), 256, 0, stream>>>(g, packed, pts, level, n_pts, gfeat, gpacked, T);
  } else {
    const long long rows = 7 * n_pts, work = rows * (3 * g.C);
    vm_scatter_kernel<<<tf_blocks(work, 256), 256, 0, stream>>>(g, packed, pts, level, rows, gfeat, gpacked, T);
  }
  TF_LAUNCH_CHECK("tf_vm_scatter_taps");
  return TF_OK;
}
