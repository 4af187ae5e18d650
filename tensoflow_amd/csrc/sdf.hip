// Fused TensoSDF evaluation for the ray-march: VM gather -> feature assembly -> 111-256-129 decoder
// -> 6-tap central differences -> NeuS alpha, one launch.
// Replaces TensoSDF.forward x7 (network/fields.py:262-299, :227-260) + compute_sdf_alpha
// (network/shapeRenderer.py:995-1025): in the reference every one of the 7 evaluations is 6 dr.texture
// launches (each preceded by a full-plane permute().contiguous() copy and a mip rebuild), a [N,111]
// concat and two GEMMs that compute all 129 outputs and throw 128 away for the FD taps.
//
// Mapping (wave64, one wave per SIMD, 4 waves per workgroup, persistent workgroups):
//   * a wave owns 32 samples; sample r sits on MFMA column r; lane (r, h = lane>>5) gathers the 16-byte
//     channel chunks q with q%2 == h of the 27 plane*line chunks (+ the xyz chunk), which is exactly the
//     fp32-MFMA B-operand layout (mfma_mlp.h), so features never leave registers;
//   * W1 (fragment order, 112 KB) lives in LDS for the whole kernel; hidden activations stay in the
//     accumulator registers; the 6 FD taps only need the sdf row of W2 (a 256-long dot product, VALU +
//     one cross-lane add); the centre tap runs the 256x128 appearance GEMM with W2 streamed from L2 in
//     fragment order (coalesced 256 B per MFMA);
//   * the field (51 MB at R=300 incl. mips) is served from L2 / Infinity Cache after first touch.
// Algorithmic gather traffic: 7 x (3 planes x 4 texels + 3 lines x 2 texels) x C x 4 B = 18 144 B per live
// sample and mip level (C = 36); flops: 122 880 + 6 x 57 344 = 466 944 per sample (fp32 MFMA).
#include <type_traits>

#include "mfma_mlp.h"
#include "tf_common.h"
#include "tf_internal.h"

#define SDF_C 36
#define SDF_HID 256
#define SDF_APP 128
#define SDF_KSTEPS 56  // (108 features + xyz + pad) / 2

// workspace layout (floats)
static constexpr int kW1f = 0;                                   // [8][56][64]
static constexpr int kB1a = kW1f + 8 * SDF_KSTEPS * 64;          // accumulator order, lane half major: [2][8][16] (a lane's 128 values are contiguous: ds_read_b128)
static constexpr int kW2r0 = kB1a + 256;                         // sdf row of W2, [2][8][16]
static constexpr int kB2a = kW2r0 + 256;                         // b2[1:], [2][4][16]
static constexpr int kLdsFloats = kB2a + 128;                    // everything above is copied to LDS
static constexpr int kStream = kLdsFloats;                       // 2 x 4096 floats: W2 streaming double buffer
static constexpr int kGeo = kStream + 2 * 4096;                  // 12 x 8 ints: per (plane, level) geometry
static constexpr int kLdsTotal = kGeo + 96;
static constexpr int kW2f = kLdsFloats;                          // [4][128][64]  (global only)
static constexpr int kSdfWsFloats = kW2f + 4 * 128 * 64;

extern "C" size_t tf_sdf_workspace_floats(void) { return kSdfWsFloats; }

// (the bias b1, the sdf row of W2 and b2[1:] are stored in accumulator order, LANE-HALF MAJOR -- TfPackBatch::acc_major: the 16 * tiles
// values of a lane are contiguous, one ds_read_b128 per four; the [..][2] interleave of tf_pack_bias_kernel made each its own ds_read_b32)

struct SdfArgs {
  VmGeom g;
  const float* packed;
  const float* ws;      // packed weights (see layout above)
  const float* b2;      // [1+A] (device); b2[0] = sdf bias
  const float* pts;     // [n,3]
  const float* level;   // [n] or null
  long long n;
  // mode 0/1 outputs
  float* sdf;           // [n]
  float* feat;          // [n,A] or null
  // mode 2 (alpha)
  const float* dists;
  const float* dirs;
  float units[3];
  float inv_s, cos_anneal;
  float* alpha;
  float* grad;
  float* nhess;
  float* taps;          // [n,6] or null: the six finite-difference sdf values (x+, x-, y+, y-, z+, z-), kept for tf_sdf_alpha_bwd
};

__device__ __forceinline__ float4 f4_lerp(float4 a, float4 b, float t) {
  const float s = 1.f - t;
  return make_float4(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t, a.w * s + b.w * t);
}

// ---- gather, split into an asynchronous "issue the 6 texel loads of one mip level" half and a "blend" half so
// that the loads of the NEXT feature group fly while the MFMAs of the current one execute.
struct ChunkRaw {
  float4 t00, t10, t01, t11, s0, s1;
  float fx, fy, fz;
};

// Per (plane, level) geometry lives in LDS as 8 ints {H, W, L, plane offset, line offset, -, -, -}: the plane index
// of a chunk differs between the two lane halves and the level differs per lane, so the lookup is a per-lane LDS
// read instead of select chains over kernel arguments.
template <bool TEX16>
__device__ __forceinline__ void load_chunk(const int* __restrict__ geo, const float* __restrict__ packed, const float (&p)[3],
                                           int l, int q, ChunkRaw& r) {
  const int i = q / (SDF_C / 4), j = q % (SDF_C / 4);
  const float u = i == 2 ? p[1] : p[0], v = i == 0 ? p[1] : p[2], w = i == 0 ? p[2] : (i == 1 ? p[1] : p[0]);
  const int4 ga = *reinterpret_cast<const int4*>(geo + (i * 4 + l) * 8);
  const int loff = geo[(i * 4 + l) * 8 + 4];
  const int H = ga.x, W = ga.y, L = ga.z;
  int x0, x1, y0, y1, z0, z1;
  axis_taps(u, W, x0, x1, r.fx);
  axis_taps(v, H, y0, y1, r.fy);
  axis_taps(w, L, z0, z1, r.fz);
  const unsigned pb = (unsigned)(ga.w + 4 * j), lb = (unsigned)(loff + 4 * j);      // element offsets (the same in a half pyramid)
  r.t00 = vm_texel4<TEX16>(packed, pb + (unsigned)((y0 * W + x0) * SDF_C));
  r.t10 = vm_texel4<TEX16>(packed, pb + (unsigned)((y0 * W + x1) * SDF_C));
  r.t01 = vm_texel4<TEX16>(packed, pb + (unsigned)((y1 * W + x0) * SDF_C));
  r.t11 = vm_texel4<TEX16>(packed, pb + (unsigned)((y1 * W + x1) * SDF_C));
#ifdef SDF_ABLATE_LINES   // dev-only timing ablation (tools/exp_march_lines.py; results are garbage): the two line taps of a chunk are NOT fetched --
  // the upper bound of what staging the line factors in LDS (north_star) could return, since an LDS read is not free either
  r.s0 = make_float4(1.f, 1.f, 1.f, 1.f); r.s1 = r.s0;
  (void)lb; (void)z0; (void)z1;
#else
  r.s0 = vm_texel4<TEX16>(packed, lb + (unsigned)(z0 * SDF_C));
  r.s1 = vm_texel4<TEX16>(packed, lb + (unsigned)(z1 * SDF_C));
#endif
}

// The blend of a chunk, written on PAIRS of channels (xy | zw of the 16-byte texel segments) so that every multiply-add is one packed
// instruction (v_pk_mul_f32 / v_pk_fma_f32: two fp32 lanes per issue) on registers that are adjacent as loaded.  Left to itself the
// compiler packed ACROSS the two texels of a lerp ((t00.x, t10.x) * (1 - f, f)): every pair then had to be assembled with v_mov --
// 550 register moves and ~1 400 vector instructions of blend arithmetic per field evaluation, a third of the kernel's issue slots.
typedef float tf_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tf_f2 f2_lo(const float4& v) { return tf_f2{v.x, v.y}; }
__device__ __forceinline__ tf_f2 f2_hi(const float4& v) { return tf_f2{v.z, v.w}; }
__device__ __forceinline__ tf_f2 f2_lerp(tf_f2 a, tf_f2 b, float s, float t) { return a * s + b * t; }

template <int NL>
__device__ __forceinline__ float4 blend_chunk(const ChunkRaw (&r)[NL], float fl) {
  tf_f2 pa = {0.f, 0.f}, pb = {0.f, 0.f}, la = {0.f, 0.f}, lb = {0.f, 0.f};
#pragma unroll
  for (int li = 0; li < NL; ++li) {
    const float wl = NL == 1 ? 1.f : (li ? fl : 1.f - fl);
    const float sx = 1.f - r[li].fx, sy = 1.f - r[li].fy, sz = 1.f - r[li].fz;
    const tf_f2 ta = f2_lerp(f2_lo(r[li].t00), f2_lo(r[li].t10), sx, r[li].fx), tb = f2_lerp(f2_hi(r[li].t00), f2_hi(r[li].t10), sx, r[li].fx);
    const tf_f2 ba = f2_lerp(f2_lo(r[li].t01), f2_lo(r[li].t11), sx, r[li].fx), bb = f2_lerp(f2_hi(r[li].t01), f2_hi(r[li].t11), sx, r[li].fx);
    const tf_f2 pla = f2_lerp(ta, ba, sy, r[li].fy), plb = f2_lerp(tb, bb, sy, r[li].fy);
    const tf_f2 lna = f2_lerp(f2_lo(r[li].s0), f2_lo(r[li].s1), sz, r[li].fz), lnb = f2_lerp(f2_hi(r[li].s0), f2_hi(r[li].s1), sz, r[li].fz);
    if (NL == 1) { pa = pla; pb = plb; la = lna; lb = lnb; }
    else { pa += pla * wl; pb += plb * wl; la += lna * wl; lb += lnb * wl; }
  }
  const tf_f2 fa = pa * la, fb = pb * lb;
  return make_float4(fa.x, fa.y, fb.x, fb.y);
}

// hidden layer for this lane's sample at world position x: fills acc[8] (post-softplus) and returns the sdf.
// NL = number of mip levels fetched (1 when no lane of the wave has a fractional LOD, else 2).
// Layer 1 runs as 7 feature groups of 2 chunks (8 k-steps x 8 unit tiles = 64 MFMAs each): while group g's MFMAs
// execute, the 12*NL texel loads of group g+1 are already in flight.
template <int NL, bool H3, bool TEX16>
__device__ __forceinline__ float sdf_hidden_nl(const SdfArgs& A, const float* lds, const float (&x)[3], int l0,
                                               int l1, float fl, int lane, f32x16 (&acc)[8], float* xrow = nullptr) {
  // Make the LDS base opaque per call: every LDS operand of this function (W1 fragments, biases, the sdf row of
  // W2: ~700 values per lane) is loop-invariant across tiles and taps, and the compiler otherwise hoists them all
  // out of the tile loop, spills them, and reloads each one from scratch behind an s_waitcnt vmcnt(0).
  {
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    lds += opaque;
  }
  const int h = lane >> 5;
  const int* geo = reinterpret_cast<const int*>(lds + kGeo);
  float p[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = (x[k] - A.g.aabb_lo[k]) / A.g.aabb_size[k];
  asm volatile("" ::: "memory");  // W1 fragments are re-read from LDS per tap (no CSE across the 7 taps)
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {      // explicit 16-byte reads (the opaque base above hides the array's alignment from the compiler)
      const float4 b4 = *reinterpret_cast<const float4*>(lds + kB1a + h * 128 + t * 16 + 4 * jj);
      acc[t][4 * jj] = b4.x; acc[t][4 * jj + 1] = b4.y; acc[t][4 * jj + 2] = b4.z; acc[t][4 * jj + 3] = b4.w;
    }
  ChunkRaw raw[2][NL];
  auto issue = [&](int g) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int q = 4 * g + 2 * c + h;        // this lane's chunks of group g: q = 2*(2g+c) + h
      if (q > 26) q = 26;               // lane half 1 of the last group carries (x,y,z,0) instead: dummy fetch
      load_chunk<TEX16>(geo, A.packed, p, l0, q, raw[c][0]);
      if (NL == 2) load_chunk<TEX16>(geo, A.packed, p, l1, q, raw[c][NL - 1]);
    }
  };
  issue(0);
#pragma unroll
  for (int g = 0; g < 7; ++g) {
    float f8[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float4 f = blend_chunk<NL>(raw[c], fl);
      if (g == 6 && c == 1) f = h ? make_float4(x[0], x[1], x[2], 0.f) : f;   // raw xyz (fields.py:265,298)
      // backward pass (tf_sdf_alpha_bwd): the decoder's input row [108 features | xyz | 1] goes to memory as it is assembled -- the
      // 112th column holds 1 so that the weight-gradient product of the first layer returns the bias gradient in its last column
      if (xrow) *reinterpret_cast<float4*>(xrow + 4 * (4 * g + 2 * c + h)) = (g == 6 && c == 1 && h) ? make_float4(f.x, f.y, f.z, 1.f) : f;
      f8[4 * c + 0] = f.x; f8[4 * c + 1] = f.y; f8[4 * c + 2] = f.z; f8[4 * c + 3] = f.w;
    }
    if (g + 1 < 7) issue(g + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (H3) {
      // f16x3: the 16 features of group g are exactly one 32x32x16 k-step (lane half h supplies k = 16g + 4h + {0..3} and
      // 16g + 8 + 4h + {0..3}: the two chunks it has just blended); fragments [g][tile][hi|lo][lane][8 halves] in LDS
      tf_h8 b_hi, b_lo;
      tf_split8(f8, b_hi, b_lo);
      const tf_h8* wf16 = reinterpret_cast<const tf_h8*>(lds + kW1f) + (8 * g) * 128 + lane;
      tf_h8 a_hi = wf16[0], a_lo = wf16[64];
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        tf_h8 n_hi = a_hi, n_lo = a_lo;
        if (tt + 1 < 8) { n_hi = wf16[(tt + 1) * 128]; n_lo = wf16[(tt + 1) * 128 + 64]; }
        acc[tt] = tf_mfma_h(a_hi, b_hi, acc[tt]);
        acc[tt] = tf_mfma_h(a_hi, b_lo, acc[tt]);
        acc[tt] = tf_mfma_h(a_lo, b_hi, acc[tt]);
        a_hi = n_hi; a_lo = n_lo;
      }
    } else {
    const float* wf = lds + kW1f + (8 * g) * 64 + lane;
    float a_cur[8], a_nxt[8];
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) a_cur[tt] = wf[tt * SDF_KSTEPS * 64];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j + 1 < 8) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) a_nxt[tt] = wf[(tt * SDF_KSTEPS + j + 1) * 64];
      }
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) acc[tt] = tf_mfma(a_cur[tt], f8[j], acc[tt]);
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) a_cur[tt] = a_nxt[tt];
    }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float part = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float4 w4 = *reinterpret_cast<const float4*>(lds + kW2r0 + h * 128 + t * 16 + 4 * jj);
      const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[t][4 * jj + k] = softplus100(acc[t][4 * jj + k]);
        part += acc[t][4 * jj + k] * wv[k];
      }
    }
  part += __shfl_xor(part, 32);
  return part + A.b2[0];
}

template <bool H3, bool TEX16>
__device__ __forceinline__ float sdf_hidden(const SdfArgs& A, const float* lds, const float (&x)[3], int l0,
                                            int l1, float fl, int lane, f32x16 (&acc)[8], float* xrow = nullptr) {
  // wave-uniform choice: one mip level is enough when no lane has a fractional LOD
  if (__any(fl != 0.f)) return sdf_hidden_nl<2, H3, TEX16>(A, lds, x, l0, l1, fl, lane, acc, xrow);
  return sdf_hidden_nl<1, H3, TEX16>(A, lds, x, l0, l1, fl, lane, acc, xrow);
}

template <int MODE, bool H3, bool TEX16 = false>  // MODE 0: sdf + feat, 1: sdf only, 2: alpha (7 taps); H3: f16x3 matrix arithmetic; TEX16: half pyramid
__global__ void __launch_bounds__(256) sdf_kernel(SdfArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < kLdsFloats; i += 256) lds[i] = A.ws[i];
  if (threadIdx.x == 0) {
    int* geo = reinterpret_cast<int*>(lds + kGeo);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        int* e = geo + (i * 4 + l) * 8;
        e[0] = vm_dim(A.g.ph[i], l); e[1] = vm_dim(A.g.pw[i], l); e[2] = vm_dim(A.g.ll[i], l);
        e[3] = (int)A.g.poff[i][l]; e[4] = (int)A.g.loff[i][l]; e[5] = e[6] = e[7] = 0;
      }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const long long n_groups = (A.n + 127) / 128;  // the 4 waves advance in lockstep (W2 streaming uses workgroup barriers)
  for (long long tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
    const long long tile = tg * 4 + wave;
    asm volatile("" ::: "memory");  // do not hoist LDS weight fragments across tiles
    long long row = tile * 32 + (lane & 31);
    const bool valid = row < A.n;
    if (!valid) row = A.n - 1;
    float x[3] = {A.pts[3 * row], A.pts[3 * row + 1], A.pts[3 * row + 2]};
    int l0, l1;
    float fl;
    mip_select(A.level ? A.level[row] : 0.f, A.g.n_levels, l0, l1, fl);
    f32x16 acc[8];
    const float s_c = sdf_hidden<H3, TEX16>(A, lds, x, l0, l1, fl, lane, acc);
    if (MODE != 1 && A.feat) {
      // appearance features: [128 x 256] * H^T, W2 fragments streamed from L2
      f32x16 o[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float4 b4 = *reinterpret_cast<const float4*>(lds + kB2a + h * 64 + t * 16 + 4 * jj);
          o[t][4 * jj] = b4.x; o[t][4 * jj + 1] = b4.y; o[t][4 * jj + 2] = b4.z; o[t][4 * jj + 3] = b4.w;
        }
      if (H3) tf_layer_stream_h3<16, 4, 8, 2, 2>(reinterpret_cast<const _Float16*>(A.ws + kW2f), lds + kStream, threadIdx.x, lane, acc, o);
      else tf_layer_stream<128, 4, 8, 16>(A.ws + kW2f, lds + kStream, threadIdx.x, lane, acc, o);
      if (valid) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            float4 v4 = make_float4(o[t][4 * jj], o[t][4 * jj + 1], o[t][4 * jj + 2], o[t][4 * jj + 3]);
            *reinterpret_cast<float4*>(A.feat + row * SDF_APP + 32 * t + 8 * jj + 4 * h) = v4;
          }
      }
    }
    if (MODE != 2) {
      if (valid && h == 0) A.sdf[row] = s_c;
      continue;
    }
    // ---- 6 finite-difference taps (fields.py:234-256)
    float sp[3], sn[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      float xt[3] = {x[0], x[1], x[2]};
      xt[ax] = x[ax] + A.units[ax];
      sp[ax] = sdf_hidden<H3, TEX16>(A, lds, xt, l0, l1, fl, lane, acc);
      xt[ax] = x[ax] - A.units[ax];
      sn[ax] = sdf_hidden<H3, TEX16>(A, lds, xt, l0, l1, fl, lane, acc);
    }
    if (valid && h == 0) {
      float g[3], hs[3];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        g[ax] = (sp[ax] - sn[ax]) / (2.f * A.units[ax]);
        hs[ax] = (sp[ax] + sn[ax] - 2.f * s_c) / (A.units[ax] * A.units[ax]);
      }
      const float d0 = A.dirs[3 * row], d1 = A.dirs[3 * row + 1], d2 = A.dirs[3 * row + 2];
      const float true_cos = d0 * g[0] + d1 * g[1] + d2 * g[2];
      const float iter_cos = -(fmaxf(-true_cos * 0.5f + 0.5f, 0.f) * (1.f - A.cos_anneal) + fmaxf(-true_cos, 0.f) * A.cos_anneal);
      const float dist = A.dists[row];
      const float pc = sigmoidf_((s_c - iter_cos * dist * 0.5f) * A.inv_s);
      const float nc = sigmoidf_((s_c + iter_cos * dist * 0.5f) * A.inv_s);
      A.alpha[row] = fminf(fmaxf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.f), 1.f);
      A.grad[3 * row] = g[0]; A.grad[3 * row + 1] = g[1]; A.grad[3 * row + 2] = g[2];
      A.sdf[row] = s_c;
      if (A.nhess) A.nhess[row] = (g[0] * hs[0] + g[1] * hs[1] + g[2] * hs[2]) / (g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + 1e-5f);
      if (A.taps) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { A.taps[6 * row + 2 * ax] = sp[ax]; A.taps[6 * row + 2 * ax + 1] = sn[ax]; }
      }
    }
  }
}

static int sdf_prepare(const TfVmDesc* d, const TfSdfMlp* mlp, const float* aabb_host, float* workspace,
                       size_t workspace_floats, SdfArgs* A, int32_t precision, hipStream_t stream, const char* who) {
  TF_REQUIRE(precision == TF_PREC_F32 || precision == TF_PREC_F16X3, TF_EINVAL, "%s: unknown precision %d", who, precision);
  TF_REQUIRE(d && mlp && aabb_host && workspace, TF_EINVAL, "%s: null pointer", who);
  TF_REQUIRE(d->C == SDF_C && mlp->hidden == SDF_HID && mlp->app_dim == SDF_APP, TF_ESHAPE,
             "%s: this build instantiates C=%d hidden=%d app_dim=%d (got %d/%d/%d)", who, SDF_C, SDF_HID, SDF_APP, d->C,
             mlp->hidden, mlp->app_dim);
  int rc = vm_geom_init(d, aabb_host, &A->g);
  TF_REQUIRE(rc != -1, TF_EINVAL, "%s: bad TfVmDesc", who);
  TF_REQUIRE(rc != -2, TF_ESHAPE, "%s: plane/line sizes > 1 must be divisible by 2^(n_levels-1)", who);
  TF_REQUIRE(A->g.total < (1LL << 31), TF_ESHAPE, "%s: packed field exceeds 2^31 floats", who);
  TF_REQUIRE(!A->g.texel_f16 || precision == TF_PREC_F16X3, TF_EINVAL, "%s: a half pyramid (texel_f16) runs with the TF_PREC_F16X3 decoder only", who);
  TF_REQUIRE(workspace_floats >= (size_t)kSdfWsFloats, TF_ESHAPE, "%s: workspace too small (%zu < %d floats)", who,
             workspace_floats, kSdfWsFloats);
  TF_REQUIRE(mlp->w1 && mlp->b1 && mlp->w2 && mlp->b2, TF_EINVAL, "%s: null weight pointer", who);
  const int K = 3 * SDF_C + 3;
  // the f16x3 images have the same size as the fp32 fragment images they replace (hi + lo halves = 4 bytes per weight)
  TfPackBatch PB(stream);           // the five packs of the decoder: ONE launch
  if (precision == TF_PREC_F16X3) PB.wfrag_h3(mlp->w1, SDF_HID, K, 0, K, 8, 7, reinterpret_cast<_Float16*>(workspace + kW1f));
  else PB.wfrag(mlp->w1, SDF_HID, K, 0, K, 8, SDF_KSTEPS, workspace + kW1f);
  PB.acc_major(mlp->b1, SDF_HID, 8, workspace + kB1a);
  PB.acc_major(mlp->w2, SDF_HID, 8, workspace + kW2r0);           // row 0 of W2
  PB.acc_major(mlp->b2 + 1, SDF_APP, 4, workspace + kB2a);
  if (precision == TF_PREC_F16X3) PB.wfrag_h3(mlp->w2 + SDF_HID, SDF_APP, SDF_HID, 0, SDF_HID, 4, 16, reinterpret_cast<_Float16*>(workspace + kW2f));
  else PB.wfrag(mlp->w2 + SDF_HID, SDF_APP, SDF_HID, 0, SDF_HID, 4, 128, workspace + kW2f, 1);
  PB.flush();
  A->ws = workspace;
  return TF_OK;
}

template <int MODE, bool H3, bool TEX16 = false>
static int sdf_launch(SdfArgs& A, const float* b2_dev, hipStream_t stream, const char* who) {
  A.b2 = b2_dev;
  const size_t lds = (size_t)kLdsTotal * sizeof(float);  // weights + W2 streaming double buffer + geometry table
  static std::atomic<unsigned long long> attr_set{0};
  int attr_dev;
  if (tf_once_needed(attr_set, &attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)sdf_kernel<MODE, H3, TEX16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    TF_REQUIRE(e == hipSuccess, TF_EHIP, "%s: hipFuncSetAttribute failed: %s", who, hipGetErrorString(e));
    tf_once_done(attr_set, attr_dev);
  }
  long long blocks = (A.n + 127) / 128;
  if (blocks > 256) blocks = 256;  // one 150 KB-LDS workgroup per CU; waves loop over tile groups
  sdf_kernel<MODE, H3, TEX16><<<(unsigned)blocks, 256, lds, stream>>>(A);
  TF_LAUNCH_CHECK(who);
  return TF_OK;
}

extern "C" int tf_sdf_forward(const TfVmDesc* d, const float* packed, const TfSdfMlp* mlp, const float* xyz,
                              const float* level, const float* aabb_host, int64_t n, float* sdf, float* feat,
                              int32_t precision, float* workspace, size_t workspace_floats, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_sdf_forward: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(packed && xyz && sdf, TF_EINVAL, "tf_sdf_forward: null pointer");
  SdfArgs A = {};
  if (int rc = sdf_prepare(d, mlp, aabb_host, workspace, workspace_floats, &A, precision, stream, "tf_sdf_forward")) return rc;
  A.packed = packed; A.pts = xyz; A.level = level; A.n = n; A.sdf = sdf; A.feat = feat;
  if (A.g.texel_f16)
    return feat ? sdf_launch<0, true, true>(A, mlp->b2, stream, "tf_sdf_forward") : sdf_launch<1, true, true>(A, mlp->b2, stream, "tf_sdf_forward");
  if (precision == TF_PREC_F16X3)
    return feat ? sdf_launch<0, true>(A, mlp->b2, stream, "tf_sdf_forward") : sdf_launch<1, true>(A, mlp->b2, stream, "tf_sdf_forward");
  return feat ? sdf_launch<0, false>(A, mlp->b2, stream, "tf_sdf_forward") : sdf_launch<1, false>(A, mlp->b2, stream, "tf_sdf_forward");
}

extern "C" int tf_sdf_alpha_fwd(const TfVmDesc* d, const float* packed, const TfSdfMlp* mlp, const float* pts,
                                const float* level, const float* dists, const float* dirs, const float* aabb_host,
                                const float* units_host, float inv_s, float cos_anneal, int64_t n, float* alpha, float* grad,
                                float* feat, float* sdf, float* nhess, float* taps, int32_t precision, float* workspace,
                                size_t workspace_floats, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_sdf_alpha_fwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(packed && pts && dists && dirs && units_host && alpha && grad && sdf, TF_EINVAL, "tf_sdf_alpha_fwd: null pointer");
  SdfArgs A = {};
  if (int rc = sdf_prepare(d, mlp, aabb_host, workspace, workspace_floats, &A, precision, stream, "tf_sdf_alpha_fwd")) return rc;
  A.packed = packed; A.pts = pts; A.level = level; A.n = n; A.sdf = sdf; A.feat = feat;
  A.dists = dists; A.dirs = dirs; A.inv_s = inv_s; A.cos_anneal = cos_anneal;
  for (int k = 0; k < 3; ++k) A.units[k] = units_host[k];
  A.alpha = alpha; A.grad = grad; A.nhess = nhess; A.taps = taps;
  if (A.g.texel_f16) return sdf_launch<2, true, true>(A, mlp->b2, stream, "tf_sdf_alpha_fwd");
  return precision == TF_PREC_F16X3 ? sdf_launch<2, true>(A, mlp->b2, stream, "tf_sdf_alpha_fwd")
                                    : sdf_launch<2, false>(A, mlp->b2, stream, "tf_sdf_alpha_fwd");
}


// =====================================================================================================================
// Backward of ShapeRenderer.compute_sdf_alpha (shapeRenderer.py:995-1025 over fields.py:227-260, :262-299) as ONE entry point.
// The reference differentiates 7 x (6 dr.texture + concat + 2 GEMMs) with autograd; rounds 1-3 here re-ran a torch composition
// (~160 launches, two 939 MB tensors added to each other).  tf_sdf_alpha_bwd is a fixed pipeline of seven launches:
//   1. dh_app [n,256] = g_feat . W2[1:]                                   (exact-fp32 matrix cores, linear.hip)
//   2. sdf_bwd_kernel: per sample the closed-form adjoint of alpha / cos / finite differences / normal-hessian -> d s_t for the 7 taps
//      (from the tap values the forward kept); then per tap the SAME recompute as the forward kernel (gather -> features -> layer 1 ->
//      Softplus, W1 fragments in LDS) -> h_t; dz_t = (d s_t w2[0] + [t = 0] dh_app) * (1 - exp(-100 h_t)) in the accumulator
//      registers; rows h_t, dz_t, [features | xyz | 1] go to the workspace, tap-major
//   3. din [7n,112] = dz . W1,  dW1|db1 [256,112] += dz^T . [X | 1]          (one column of ones: the bias gradient rides along)
//   4. dW2[0] += ds^T . h (thin kernel),  dW2[1:] += g_feat^T . h_0,  db2 = column sums
//   5. the 7-tap scatter of din into the pyramid gradient (request-coalesced lane mapping of tf_vm_gather_bwd)
// Samples are processed in chunks of kBwdChunk so that the workspace stays bounded.
static constexpr long long kBwdChunk = 1LL << 18;
static constexpr int kXld = 112;

struct SdfBwdArgs {
  const float* sdf;      // [n]   forward outputs kept by the caller
  const float* taps;     // [n,6]
  const float* g_alpha;  // [n] or null
  const float* g_grad;   // [n,3] or null
  const float* g_sdf;    // [n] or null
  const float* g_nh;     // [n] or null
  const float* dh_app;   // [n,256] or null (g_feat . W2[1:])
  float* dz;             // [7n,256]
  float* hh;             // [7n,256]
  float* X;              // [7n,112]
  float* ds;             // [7n]
  float* scal;           // [2]: d inv_s, sum of ds (= d b2[0])
};

template <bool H3>
__global__ void __launch_bounds__(256) sdf_bwd_kernel(SdfArgs A, SdfBwdArgs B) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < kLdsFloats; i += 256) lds[i] = A.ws[i];
  if (threadIdx.x == 0) {
    int* geo = reinterpret_cast<int*>(lds + kGeo);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        int* e = geo + (i * 4 + l) * 8;
        e[0] = vm_dim(A.g.ph[i], l); e[1] = vm_dim(A.g.pw[i], l); e[2] = vm_dim(A.g.ll[i], l);
        e[3] = (int)A.g.poff[i][l]; e[4] = (int)A.g.loff[i][l]; e[5] = e[6] = e[7] = 0;
      }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const long long n_groups = (A.n + 127) / 128;
  float acc_inv = 0.f, acc_ds = 0.f;
  for (long long tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
    const long long tile = tg * 4 + wave;
    asm volatile("" ::: "memory");
    long long row = tile * 32 + (lane & 31);
    const bool valid = row < A.n;
    if (!valid) row = A.n - 1;
    const float x[3] = {A.pts[3 * row], A.pts[3 * row + 1], A.pts[3 * row + 2]};
    int l0, l1;
    float fl;
    mip_select(A.level ? A.level[row] : 0.f, A.g.n_levels, l0, l1, fl);
    // ---- closed-form adjoint of everything behind the seven sdf values (both lane halves of a sample compute the same numbers)
    float ds[7];
    {
      const float s0 = B.sdf[row];
      float sp[3], sn[3], g[3], hs[3];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        sp[ax] = B.taps[6 * row + 2 * ax]; sn[ax] = B.taps[6 * row + 2 * ax + 1];
        g[ax] = (sp[ax] - sn[ax]) / (2.f * A.units[ax]);
        hs[ax] = (sp[ax] + sn[ax] - 2.f * s0) / (A.units[ax] * A.units[ax]);
      }
      const float d[3] = {A.dirs[3 * row], A.dirs[3 * row + 1], A.dirs[3 * row + 2]};
      const float tc = d[0] * g[0] + d[1] * g[1] + d[2] * g[2];
      const float ra = -tc * 0.5f + 0.5f, rb = -tc;
      const float ic = -(fmaxf(ra, 0.f) * (1.f - A.cos_anneal) + fmaxf(rb, 0.f) * A.cos_anneal);
      const float dist = A.dists[row];
      const float e = ic * dist * 0.5f;
      const float pc = sigmoidf_((s0 - e) * A.inv_s), nc = sigmoidf_((s0 + e) * A.inv_s);
      const float D = pc + 1e-5f, Aq = (pc - nc + 1e-5f) / D;
      const float ga = B.g_alpha ? B.g_alpha[row] : 0.f;
      const float dA = (Aq >= 0.f && Aq <= 1.f) ? ga : 0.f;                     // clip passes its gradient on [0, 1] inclusive
      const float dpc = dA * (1.f - Aq) / D * (pc * (1.f - pc)), dnc = -dA / D * (nc * (1.f - nc));      // adjoints of the two sigmoid ARGUMENTS
      float ds0 = (dpc + dnc) * A.inv_s;
      const float de = (dnc - dpc) * A.inv_s;
      const float dinv = dpc * (s0 - e) + dnc * (s0 + e);
      const float dic = de * dist * 0.5f;
      const float dtc = (ra > 0.f ? 0.5f * dic * (1.f - A.cos_anneal) : 0.f) + (rb > 0.f ? dic * A.cos_anneal : 0.f);
      const float gn = B.g_nh ? B.g_nh[row] : 0.f;
      const float G2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + 1e-5f, NUM = g[0] * hs[0] + g[1] * hs[1] + g[2] * hs[2];
      const float dNUM = gn / G2, dG2 = -gn * NUM / (G2 * G2);
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const float dg = dtc * d[ax] + (B.g_grad ? B.g_grad[3 * row + ax] : 0.f) + dNUM * hs[ax] + 2.f * dG2 * g[ax];
        const float dh = dNUM * g[ax] / (A.units[ax] * A.units[ax]);
        ds[1 + 2 * ax] = dg / (2.f * A.units[ax]) + dh;
        ds[2 + 2 * ax] = -dg / (2.f * A.units[ax]) + dh;
        ds0 -= 2.f * dh;
      }
      ds[0] = ds0 + (B.g_sdf ? B.g_sdf[row] : 0.f);
      if (!valid) {
#pragma unroll
        for (int t = 0; t < 7; ++t) ds[t] = 0.f;
      }
      if (valid && h == 0) {
        acc_inv += dinv;
#pragma unroll
        for (int t = 0; t < 7; ++t) { acc_ds += ds[t]; B.ds[(long long)t * A.n + row] = ds[t]; }
      }
    }
    // ---- per tap: recompute the hidden layer, form dz in the accumulator registers, write h | dz | input row
    f32x16 acc[8];
    // rows of lanes past the end go to the TRASH row 7 n of every buffer: no divergent store paths (the first version -- `if (valid)`
    // around 64 float4 stores, a per-store `if (centre tap)` -- spilled 513 scalar and 224 vector registers)
    auto tap = [&](int t, auto app_tag) __attribute__((always_inline)) {
      constexpr bool APP = decltype(app_tag)::value;
      const int ax = (t + 1) / 2 - 1;                         // -1: the centre tap
      const float sgn = (t & 1) ? 1.f : -1.f;
      float xt[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) xt[k] = (k == ax) ? x[k] + sgn * A.units[k] : x[k];        // x + u / x + (-u): the forward's x + u / x - u
      float dst = ds[0];
#pragma unroll
      for (int k = 1; k < 7; ++k) dst = (t == k) ? ds[k] : dst;
      long long rt = valid ? (long long)t * A.n + row : 7 * A.n;
      asm volatile("" : "+v"(rt));                            // per-tap row offset: not a loop invariant to hoist and spill
      sdf_hidden<H3, false>(A, lds, xt, l0, l1, fl, lane, acc, B.X + rt * kXld);
      float* hrow = B.hh + rt * SDF_HID + 4 * h;
      float* zrow = B.dz + rt * SDF_HID + 4 * h;
      const float* arow = APP ? B.dh_app + row * SDF_HID + 4 * h : nullptr;
      const float* w2r = lds + kW2r0 + h * 128;
      {
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        w2r += opaque;
      }
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (APP) a4 = *reinterpret_cast<const float4*>(arow + 32 * tt + 8 * jj);
          const float ap[4] = {a4.x, a4.y, a4.z, a4.w};
          float hv[4], zv[4];
          const float4 w4 = *reinterpret_cast<const float4*>(w2r + tt * 16 + 4 * jj);
          const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float hval = acc[tt][4 * jj + k];
            const float dh = dst * wv[k] + ap[k];
            // Softplus'(z) = sigmoid(100 z) = 1 - exp(-100 h); the linear branch (100 z > 20, i.e. h = z > 0.2) has derivative 1
            const float sg = hval > 0.2f ? 1.f : 1.f - __builtin_amdgcn_exp2f(-144.269504088896341f * hval);
            hv[k] = hval; zv[k] = dh * sg;
          }
          *reinterpret_cast<float4*>(hrow + 32 * tt + 8 * jj) = make_float4(hv[0], hv[1], hv[2], hv[3]);
          *reinterpret_cast<float4*>(zrow + 32 * tt + 8 * jj) = make_float4(zv[0], zv[1], zv[2], zv[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // the centre tap (which alone carries the appearance features' gradient) is peeled; the six finite-difference taps share ONE copy
    // of the gather + layer-1 body (seven inlined copies -- the forward kernel's shape -- made every tap's row addresses loop
    // invariants of the tile loop)
    if (B.dh_app) tap(0, std::true_type{});
    else tap(0, std::false_type{});
#pragma unroll 1
    for (int t = 1; t < 7; ++t) tap(t, std::false_type{});
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { acc_inv += __shfl_xor(acc_inv, o); acc_ds += __shfl_xor(acc_ds, o); }
  if (lane == 0) { atomicAdd(B.scal, acc_inv); atomicAdd(B.scal + 1, acc_ds); }
}

namespace {
__global__ void __launch_bounds__(256) sdf_pad_w1_kernel(const float* __restrict__ w1, float* __restrict__ w1p) {
  const int e = blockIdx.x * 256 + threadIdx.x;             // [256, 112] <- [256, 111] | 0
  if (e >= SDF_HID * kXld) return;
  const int r = e / kXld, c = e % kXld;
  w1p[e] = c < 3 * SDF_C + 3 ? w1[r * (3 * SDF_C + 3) + c] : 0.f;
}
__global__ void __launch_bounds__(256) sdf_unpad_gw1_kernel(const float* __restrict__ gp, float* __restrict__ g_w1, float* __restrict__ g_b1) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= SDF_HID * kXld) return;
  const int r = e / kXld, c = e % kXld;
  if (c < 3 * SDF_C + 3) { if (g_w1) g_w1[r * (3 * SDF_C + 3) + c] = gp[e]; }
  else if (g_b1) g_b1[r] = gp[e];                           // the column of ones: d b1
}
// column sums of X [n, N] (N <= 256), added to out[N]
__global__ void __launch_bounds__(256) sdf_colsum_kernel(const float* __restrict__ X, long long n, int N, float* __restrict__ out) {
  const long long r0 = (long long)blockIdx.x * 512, r1 = min(r0 + 512, n);
  for (int j = threadIdx.x; j < N; j += 256) {
    float s = 0.f;
    for (long long r = r0; r < r1; ++r) s += X[r * N + j];
    atomicAdd(out + j, s);
  }
}
__global__ void sdf_finish_scalars_kernel(const float* __restrict__ scal, float* __restrict__ g_inv_s, float* __restrict__ g_b2) {
  if (threadIdx.x == 0) {
    if (g_inv_s) g_inv_s[0] = scal[0];
    if (g_b2) g_b2[0] = scal[1];
  }
}
}  // namespace

extern "C" size_t tf_sdf_alpha_bwd_workspace_floats(int64_t n) {
  const long long c = n < kBwdChunk ? (n < 1 ? 1 : n) : kBwdChunk;
  // fragment workspace | scalars + padded W1 + its gradient | dh_app [c,256] | dz, h [7c+1,256] | X, din [7c+1,112] | ds [7c]
  // (+1: the trash row that lanes past the end write to)
  return (size_t)kSdfWsFloats + 64 + 2 * SDF_HID * kXld + (size_t)c * SDF_HID + 2 * (size_t)(7 * c + 1) * SDF_HID + 2 * (size_t)(7 * c + 1) * kXld + (size_t)7 * c + 64;
}

extern "C" int tf_sdf_alpha_bwd(const TfVmDesc* d, const float* packed, const TfSdfMlp* mlp, const float* pts, const float* level,
                                const float* dists, const float* dirs, const float* aabb_host, const float* units_host, float inv_s,
                                float cos_anneal, int64_t n, const float* sdf, const float* taps, const float* g_alpha,
                                const float* g_grad, const float* g_feat, const float* g_sdf, const float* g_nhess, float* gpacked,
                                float* g_w1, float* g_b1, float* g_w2, float* g_b2, float* g_inv_s, int32_t precision, float* workspace,
                                size_t workspace_floats, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_sdf_alpha_bwd: n < 0");
  TF_REQUIRE(d && mlp && pts && dists && dirs && units_host && aabb_host && workspace, TF_EINVAL, "tf_sdf_alpha_bwd: null pointer");
  TF_REQUIRE(n == 0 || (sdf && taps && packed), TF_EINVAL, "tf_sdf_alpha_bwd: the forward's sdf / taps outputs and the pyramid are required");
  TF_REQUIRE(gpacked && g_w1 && g_b1 && g_w2 && g_b2, TF_EINVAL, "tf_sdf_alpha_bwd: null gradient pointer");
  TF_REQUIRE(!d->texel_f16, TF_EINVAL, "tf_sdf_alpha_bwd: the adjoint takes an fp32 pyramid (texel_f16 is an inference-only format)");
  TF_REQUIRE(workspace_floats >= tf_sdf_alpha_bwd_workspace_floats(n), TF_ESHAPE, "tf_sdf_alpha_bwd: workspace too small (%zu < %zu floats)",
             workspace_floats, tf_sdf_alpha_bwd_workspace_floats(n));
  SdfArgs A = {};
  if (int rc = sdf_prepare(d, mlp, aabb_host, workspace, workspace_floats, &A, precision, stream, "tf_sdf_alpha_bwd")) return rc;
  const int K1 = 3 * SDF_C + 3;
  // gradients of the decoder are overwritten (accumulated over the chunks below)
  hipError_t e = hipMemsetAsync(g_w2, 0, sizeof(float) * (size_t)(1 + SDF_APP) * SDF_HID, stream);
  TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_sdf_alpha_bwd: memset failed");
  e = hipMemsetAsync(g_b2, 0, sizeof(float) * (1 + SDF_APP), stream);
  TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_sdf_alpha_bwd: memset failed");
  float* scal = workspace + kSdfWsFloats;                 // [64]
  float* w1p = scal + 64;                                 // [256,112]
  float* gw1p = w1p + SDF_HID * kXld;                     // [256,112]
  e = hipMemsetAsync(scal, 0, sizeof(float) * (64 + 2 * SDF_HID * kXld), stream);
  TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_sdf_alpha_bwd: memset failed");
  sdf_pad_w1_kernel<<<tf_blocks(SDF_HID * kXld, 256), 256, 0, stream>>>(mlp->w1, w1p);
  const long long c_max = n < kBwdChunk ? (n < 1 ? 1 : n) : kBwdChunk;
  float* dh_app = gw1p + SDF_HID * kXld;
  float* dz = dh_app + c_max * SDF_HID;
  float* hh = dz + (7 * c_max + 1) * SDF_HID;
  float* X = hh + (7 * c_max + 1) * SDF_HID;
  float* din = X + (7 * c_max + 1) * kXld;
  float* dsb = din + (7 * c_max + 1) * kXld;
  const size_t lds = (size_t)kLdsTotal * sizeof(float);
  static std::atomic<unsigned long long> attr_set{0};
  int attr_dev;
  if (tf_once_needed(attr_set, &attr_dev)) {
    hipError_t e1 = hipFuncSetAttribute((const void*)sdf_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e2 = hipFuncSetAttribute((const void*)sdf_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    TF_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, TF_EHIP, "tf_sdf_alpha_bwd: hipFuncSetAttribute failed");
    tf_once_done(attr_set, attr_dev);
  }
  A.packed = packed; A.b2 = mlp->b2; A.inv_s = inv_s; A.cos_anneal = cos_anneal;
  for (int k = 0; k < 3; ++k) A.units[k] = units_host[k];
  for (long long c0 = 0; c0 < n; c0 += kBwdChunk) {
    const long long c = (n - c0) < kBwdChunk ? (n - c0) : kBwdChunk;
    const float* gf = g_feat ? g_feat + c0 * SDF_APP : nullptr;
    if (gf) {     // dh_app = g_feat . W2[1:]   (gZ [c,128], W [128,256])
      if (int rc = tf_linear_products(nullptr, mlp->w2 + SDF_HID, gf, c, SDF_HID, SDF_APP, TF_PREC_BF16X3, dh_app, nullptr, nullptr, stream)) return rc;
    }
    A.pts = pts + 3 * c0; A.level = level ? level + c0 : nullptr; A.n = c; A.dists = dists + c0; A.dirs = dirs + 3 * c0;
    SdfBwdArgs B = {sdf + c0, taps + 6 * c0, g_alpha ? g_alpha + c0 : nullptr, g_grad ? g_grad + 3 * c0 : nullptr,
                    g_sdf ? g_sdf + c0 : nullptr, g_nhess ? g_nhess + c0 : nullptr, gf ? dh_app : nullptr, dz, hh, X, dsb, scal};
    long long blocks = (c + 127) / 128;
    if (blocks > 256) blocks = 256;
    if (precision == TF_PREC_F16X3) sdf_bwd_kernel<true><<<(unsigned)blocks, 256, lds, stream>>>(A, B);
    else sdf_bwd_kernel<false><<<(unsigned)blocks, 256, lds, stream>>>(A, B);
    TF_LAUNCH_CHECK("tf_sdf_alpha_bwd(recompute)");
    // first layer: din = dz . W1p, [dW1 | db1] += dz^T . [X | 1]
    if (int rc = tf_linear_products(X, w1p, dz, 7 * c, kXld, SDF_HID, TF_PREC_BF16X3, din, gw1p, nullptr, stream)) return rc;
    // second layer: row 0 (the sdf) over all taps, rows 1.. (appearance features) on the centre tap (rows [0, c) of h)
    if (int rc = tf_linear_products(hh, mlp->w2, dsb, 7 * c, SDF_HID, 1, TF_PREC_BF16X3, nullptr, g_w2, nullptr, stream)) return rc;
    if (gf) {
      if (int rc = tf_linear_products(hh, mlp->w2 + SDF_HID, gf, c, SDF_HID, SDF_APP, TF_PREC_BF16X3, nullptr, g_w2 + SDF_HID, nullptr, stream)) return rc;
      sdf_colsum_kernel<<<tf_blocks(c, 512), 256, 0, stream>>>(gf, c, SDF_APP, g_b2 + 1);
    }
    if (int rc = tf_vm_scatter_taps(A.g, packed, A.pts, A.level, c, A.units, din, kXld, gpacked, stream)) return rc;
  }
  sdf_unpad_gw1_kernel<<<tf_blocks(SDF_HID * kXld, 256), 256, 0, stream>>>(gw1p, g_w1, g_b1);
  sdf_finish_scalars_kernel<<<1, 64, 0, stream>>>(scal, g_inv_s, g_b2);
  TF_LAUNCH_CHECK("tf_sdf_alpha_bwd");
  (void)K1;
  return TF_OK;
}
