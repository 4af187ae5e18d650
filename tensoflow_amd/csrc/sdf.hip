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
#include "mfma_mlp.h"
#include "tf_common.h"

#define SDF_C 36
#define SDF_HID 256
#define SDF_APP 128
#define SDF_KSTEPS 56  // (108 features + xyz + pad) / 2

// workspace layout (floats)
static constexpr int kW1f = 0;                                   // [8][56][64]
static constexpr int kB1a = kW1f + 8 * SDF_KSTEPS * 64;          // [8][16][2]
static constexpr int kW2r0 = kB1a + 256;                         // sdf row of W2, accumulator order [8][16][2]
static constexpr int kB2a = kW2r0 + 256;                         // b2[1:], accumulator order [4][16][2]
static constexpr int kLdsFloats = kB2a + 128;                    // everything above is copied to LDS
static constexpr int kStream = kLdsFloats;                       // 2 x 4096 floats: W2 streaming double buffer
static constexpr int kGeo = kStream + 2 * 4096;                  // 12 x 8 ints: per (plane, level) geometry
static constexpr int kLdsTotal = kGeo + 96;
static constexpr int kW2f = kLdsFloats;                          // [4][128][64]  (global only)
static constexpr int kSdfWsFloats = kW2f + 4 * 128 * 64;

extern "C" size_t tf_sdf_workspace_floats(void) { return kSdfWsFloats; }

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
  r.s0 = vm_texel4<TEX16>(packed, lb + (unsigned)(z0 * SDF_C));
  r.s1 = vm_texel4<TEX16>(packed, lb + (unsigned)(z1 * SDF_C));
}

template <int NL>
__device__ __forceinline__ float4 blend_chunk(const ChunkRaw (&r)[NL], float fl) {
  float4 pv = make_float4(0, 0, 0, 0), lv = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int li = 0; li < NL; ++li) {
    const float wl = NL == 1 ? 1.f : (li ? fl : 1.f - fl);
    const float4 pl = f4_lerp(f4_lerp(r[li].t00, r[li].t10, r[li].fx), f4_lerp(r[li].t01, r[li].t11, r[li].fx), r[li].fy);
    const float4 ln = f4_lerp(r[li].s0, r[li].s1, r[li].fz);
    pv.x += wl * pl.x; pv.y += wl * pl.y; pv.z += wl * pl.z; pv.w += wl * pl.w;
    lv.x += wl * ln.x; lv.y += wl * ln.y; lv.z += wl * ln.z; lv.w += wl * ln.w;
  }
  return make_float4(pv.x * lv.x, pv.y * lv.y, pv.z * lv.z, pv.w * lv.w);
}

// hidden layer for this lane's sample at world position x: fills acc[8] (post-softplus) and returns the sdf.
// NL = number of mip levels fetched (1 when no lane of the wave has a fractional LOD, else 2).
// Layer 1 runs as 7 feature groups of 2 chunks (8 k-steps x 8 unit tiles = 64 MFMAs each): while group g's MFMAs
// execute, the 12*NL texel loads of group g+1 are already in flight.
template <int NL, bool H3, bool TEX16>
__device__ __forceinline__ float sdf_hidden_nl(const SdfArgs& A, const float* lds, const float (&x)[3], int l0,
                                               int l1, float fl, int lane, f32x16 (&acc)[8]) {
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
    for (int j = 0; j < 16; ++j) acc[t][j] = lds[kB1a + (t * 16 + j) * 2 + h];
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
    for (int j = 0; j < 16; ++j) {
      acc[t][j] = softplus100(acc[t][j]);
      part += acc[t][j] * lds[kW2r0 + (t * 16 + j) * 2 + h];
    }
  part += __shfl_xor(part, 32);
  return part + A.b2[0];
}

template <bool H3, bool TEX16>
__device__ __forceinline__ float sdf_hidden(const SdfArgs& A, const float* lds, const float (&x)[3], int l0,
                                            int l1, float fl, int lane, f32x16 (&acc)[8]) {
  // wave-uniform choice: one mip level is enough when no lane has a fractional LOD
  if (__any(fl != 0.f)) return sdf_hidden_nl<2, H3, TEX16>(A, lds, x, l0, l1, fl, lane, acc);
  return sdf_hidden_nl<1, H3, TEX16>(A, lds, x, l0, l1, fl, lane, acc);
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
        for (int j = 0; j < 16; ++j) o[t][j] = lds[kB2a + (t * 16 + j) * 2 + h];
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
  if (precision == TF_PREC_F16X3)
    tf_pack_wfrag_h3_kernel<<<tf_blocks(8 * 7 * 64, 256), 256, 0, stream>>>(mlp->w1, SDF_HID, K, 0, K, 8, 7,
                                                                            reinterpret_cast<_Float16*>(workspace + kW1f));
  else
    tf_pack_wfrag_kernel<<<tf_blocks(8 * SDF_KSTEPS * 64, 256), 256, 0, stream>>>(mlp->w1, SDF_HID, K, 0, K, 8, SDF_KSTEPS,
                                                                                  workspace + kW1f);
  tf_pack_bias_kernel<<<1, 256, 0, stream>>>(mlp->b1, SDF_HID, 8, workspace + kB1a);
  tf_pack_bias_kernel<<<1, 256, 0, stream>>>(mlp->w2, SDF_HID, 8, workspace + kW2r0);           // row 0 of W2
  tf_pack_bias_kernel<<<1, 256, 0, stream>>>(mlp->b2 + 1, SDF_APP, 4, workspace + kB2a);
  if (precision == TF_PREC_F16X3)
    tf_pack_wfrag_h3_kernel<<<tf_blocks(4 * 16 * 64, 256), 256, 0, stream>>>(mlp->w2 + SDF_HID, SDF_APP, SDF_HID, 0, SDF_HID, 4, 16,
                                                                             reinterpret_cast<_Float16*>(workspace + kW2f));
  else
    tf_pack_wfrag_kernel<<<tf_blocks(4 * 128 * 64, 256), 256, 0, stream>>>(mlp->w2 + SDF_HID, SDF_APP, SDF_HID, 0, SDF_HID, 4,
                                                                           128, workspace + kW2f, 1);
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
                                float* feat, float* sdf, float* nhess, int32_t precision, float* workspace,
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
  A.alpha = alpha; A.grad = grad; A.nhess = nhess;
  if (A.g.texel_f16) return sdf_launch<2, true, true>(A, mlp->b2, stream, "tf_sdf_alpha_fwd");
  return precision == TF_PREC_F16X3 ? sdf_launch<2, true>(A, mlp->b2, stream, "tf_sdf_alpha_fwd")
                                    : sdf_launch<2, false>(A, mlp->b2, stream, "tf_sdf_alpha_fwd");
}
