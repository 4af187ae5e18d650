// Per-surface-point preparation of the rendering integral, one launch:
//   * MCShadingNetwork.tenso_feature + predict_materials (network/fields.py:776-810, :1010-1017):
//       VM gather (C = 36, level 0) -> 3 weight-norm MLPs 108-128-{1,1,3} (ReLU, sigmoid), roughness remap;
//   * TensoFlow.tenso_feature for the diffuse and the specular flow (network/flow.py:709-744):
//       VM gather (C = 12) ++ embed3(xyz) -> Linear(57,64) -> Softplus(100) -> Linear(64,16), and the condition
//       row cat[feat16, embed3(view_angles) 14, 0*embed3(rough) 7] of TensoFlow.sample / .forward (:836-848, :803-815).
// The reference runs ~30 small launches here (3 full-plane permute copies + mip rebuilds per field included); as
// separate library GEMMs this stage cost 0.78 ms per 16 384 points, i.e. pure launch latency.
//
// Mapping: one wave = 32 points x one of 5 tasks (blockIdx.y: metallic, roughness, albedo, flow_diffuse,
// flow_specular).  The 64 lanes gather the points' texel segments (one lane per (point, 16-byte chunk)) into an LDS
// row per point; the MLPs run on the exact-fp32 MFMA (mfma_mlp.h), points on the MFMA column index, layer-1 B operands
// read from the LDS rows (odd row stride: conflict-free), hidden activations kept in accumulator registers.
#include "mfma_mlp.h"
#include "tf_common.h"

#define PT_MAT_C 36
#define PT_NIS_C 12
#define PT_MAT_KS 56     // k-steps covering 108 inputs (3 full 32-blocks + 8 steps of the last)
#define PT_NIS_KS 32     // k-steps covering 57 inputs (padded to 64)

// workspace (floats): fragment-ordered weights, written once by tf_point_pack
static constexpr int kPmW1 = 0;                               // [4][56][64]
static constexpr int kPmB1 = kPmW1 + 4 * PT_MAT_KS * 64;      // [4][16][2]
static constexpr int kPmW2 = kPmB1 + 128;                     // [1][64][64]
static constexpr int kPmB2 = kPmW2 + 64 * 64;                 // [16][2]
static constexpr int kPmSize = kPmB2 + 32;
static constexpr int kPnW1 = 0;                               // [2][32][64]
static constexpr int kPnB1 = kPnW1 + 2 * PT_NIS_KS * 64;      // [2][16][2]
static constexpr int kPnW2 = kPnB1 + 64;                      // [1][32][64]
static constexpr int kPnB2 = kPnW2 + 32 * 64;                 // [16][2]
static constexpr int kPnSize = kPnB2 + 32;
static constexpr int kPointWsFloats = 3 * kPmSize + 2 * kPnSize;

extern "C" size_t tf_point_workspace_floats(void) { return kPointWsFloats; }

struct PointArgs {
  VmGeom gm, gd, gs;
  const float* mat_packed;
  const float* nis_packed[2];
  const float* ws;
  const float* pts;
  const float* va;
  long long pn;
  float rough_min;
  float* metallic;
  float* roughness;
  float* albedo;
  float* cond[2];
};

__device__ __forceinline__ float4 pt_lerp4(float4 a, float4 b, float t) {
  const float s = 1.f - t;
  return make_float4(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t, a.w * s + b.w * t);
}

// level-0 VM gather of `C` channels x 3 plane/line pairs for the wave's 32 points into X[r][0 .. 3C) (row stride KP);
// same tap arithmetic as vm_gather_kernel (vm_field.hip).
template <int C, int KP>
__device__ __forceinline__ void gather_rows(const VmGeom& g, const float* __restrict__ packed, const float* __restrict__ pts,
                                            long long row0, long long pn, int lane, float* __restrict__ X) {
  constexpr int CPL = C / 4, CPP = 3 * CPL;
  for (int e = lane; e < 32 * CPP; e += 64) {
    const int r = e / CPP, q = e % CPP;
    const int i = q / CPL, j = q % CPL;
    long long row = row0 + r;
    if (row >= pn) row = pn - 1;
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = (pts[row * 3 + k] - g.aabb_lo[k]) / g.aabb_size[k];
    const float u = i == 2 ? p[1] : p[0], v = i == 0 ? p[1] : p[2], w = i == 0 ? p[2] : (i == 1 ? p[1] : p[0]);
    const int H = g.ph[i], W = g.pw[i], L = g.ll[i];
    int x0, x1, y0, y1, z0, z1;
    float fx, fy, fz;
    axis_taps(u, W, x0, x1, fx);
    axis_taps(v, H, y0, y1, fy);
    axis_taps(w, L, z0, z1, fz);
    const long long pb = g.poff[i][0] + 4 * j, lb = g.loff[i][0] + 4 * j;
    const int f16 = g.texel_f16;                       // wave-uniform: half pyramid (TfVmDesc.texel_f16), widened on load
    const float4 t00 = vm_texel4(packed, pb + ((long long)y0 * W + x0) * C, f16);
    const float4 t10 = vm_texel4(packed, pb + ((long long)y0 * W + x1) * C, f16);
    const float4 t01 = vm_texel4(packed, pb + ((long long)y1 * W + x0) * C, f16);
    const float4 t11 = vm_texel4(packed, pb + ((long long)y1 * W + x1) * C, f16);
    const float4 s0 = vm_texel4(packed, lb + (long long)z0 * C, f16);
    const float4 s1 = vm_texel4(packed, lb + (long long)z1 * C, f16);
    const float4 pl = pt_lerp4(pt_lerp4(t00, t10, fx), pt_lerp4(t01, t11, fx), fy), ln = pt_lerp4(s0, s1, fz);
    float* x = X + r * KP + 4 * q;
    x[0] = pl.x * ln.x; x[1] = pl.y * ln.y; x[2] = pl.z * ln.z; x[3] = pl.w * ln.w;
  }
}

template <int KSTEPS, int TOUT, int KP>
__device__ __forceinline__ void layer_from_lds(const float* __restrict__ wf, const float* __restrict__ X, int r, int h,
                                               f32x16 (&acc)[TOUT]) {
#pragma unroll 8
  for (int s = 0; s < KSTEPS; ++s) {
    const float b = X[r * KP + tf_kmap(s, h)];
#pragma unroll
    for (int t = 0; t < TOUT; ++t) acc[t] = tf_mfma(wf[(t * KSTEPS + s) * 64], b, acc[t]);
  }
}

__global__ void __launch_bounds__(64) point_prep_kernel(PointArgs A) {
  constexpr int KPM = 129, KPN = 65;
  __shared__ __attribute__((aligned(16))) float X[32 * KPM];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int task = blockIdx.y;
  const long long row0 = (long long)blockIdx.x * 32;
  const long long row = row0 + r;
  const bool valid = row < A.pn;
  if (task < 3) {
    // ---------------- material predictor `task` (0 metallic, 1 roughness, 2 albedo)
    for (int e = lane; e < 32 * (KPM - 108); e += 64) X[(e / (KPM - 108)) * KPM + 108 + e % (KPM - 108)] = 0.f;
    gather_rows<PT_MAT_C, KPM>(A.gm, A.mat_packed, A.pts, row0, A.pn, lane, X);
    __syncthreads();
    const float* ws = A.ws + task * kPmSize;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[t][j] = ws[kPmB1 + (t * 16 + j) * 2 + h];
    layer_from_lds<PT_MAT_KS, 4, KPM>(ws + kPmW1 + lane, X, r, h, acc);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[t][j] = tf_relu(acc[t][j]);
    f32x16 o[1];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[0][j] = ws[kPmB2 + j * 2 + h];
    tf_layer<64, 1, 4>(ws + kPmW2 + lane, acc, o);
    if (valid && h == 0) {
      if (task == 0) A.metallic[row] = 1.f / (1.f + expf(-o[0][0]));
      else if (task == 1) A.roughness[row] = (1.f / (1.f + expf(-o[0][0]))) * (1.f - A.rough_min * A.rough_min) + A.rough_min * A.rough_min;
      else {
#pragma unroll
        for (int c = 0; c < 3; ++c) A.albedo[3 * row + c] = 1.f / (1.f + expf(-o[0][c]));
      }
    }
  } else {
    // ---------------- flow feature net + condition row of flow `f`
    const int f = task - 3;
    const VmGeom& g = f ? A.gs : A.gd;
    gather_rows<PT_NIS_C, KPN>(g, A.nis_packed[f], A.pts, row0, A.pn, lane, X);
    if (h == 0) {
      const long long rr = valid ? row : A.pn - 1;
      float* x = X + r * KPN + 36;
      const float p[3] = {A.pts[3 * rr], A.pts[3 * rr + 1], A.pts[3 * rr + 2]};
#pragma unroll
      for (int k = 0; k < 3; ++k) x[k] = p[k];
#pragma unroll
      for (int fq = 0; fq < 3; ++fq)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float a = p[k] * (float)(1 << fq);
          x[3 + 6 * fq + k] = sinf(a);
          x[3 + 6 * fq + 3 + k] = cosf(a);
        }
#pragma unroll
      for (int k = 57; k < KPN; ++k) X[r * KPN + k] = 0.f;
    }
    __syncthreads();
    const float* ws = A.ws + 3 * kPmSize + f * kPnSize;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[t][j] = ws[kPnB1 + (t * 16 + j) * 2 + h];
    layer_from_lds<PT_NIS_KS, 2, KPN>(ws + kPnW1 + lane, X, r, h, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[t][j] = softplus100(acc[t][j]);
    f32x16 o[1];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[0][j] = ws[kPnB2 + j * 2 + h];
    tf_layer<32, 1, 2>(ws + kPnW2 + lane, acc, o);
    if (valid) {
      float* c = A.cond[f] + row * 37;
      // outputs 0..15 sit in accumulator registers 0..7: unit = (reg & 3) + 8 (reg >> 2) + 4 h
#pragma unroll
      for (int reg = 0; reg < 8; ++reg) c[(reg & 3) + 8 * (reg >> 2) + 4 * h] = o[0][reg];
      if (h == 0) {
        const float a0 = A.va[2 * row], a1 = A.va[2 * row + 1];
        c[16] = a0; c[17] = a1;
#pragma unroll
        for (int fq = 0; fq < 3; ++fq) {
          const float s = (float)(1 << fq);
          c[18 + 4 * fq] = sinf(a0 * s); c[19 + 4 * fq] = sinf(a1 * s);
          c[20 + 4 * fq] = cosf(a0 * s); c[21 + 4 * fq] = cosf(a1 * s);
        }
      } else {
#pragma unroll
        for (int k = 30; k < 37; ++k) c[k] = 0.f;   // 0 * embed3(roughness): the reference zeroes it (flow.py:847-848)
      }
    }
  }
}

extern "C" int tf_point_pack(const TfPointNets* nets, float* workspace, size_t workspace_floats, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(nets && workspace, TF_EINVAL, "tf_point_pack: null pointer");
  TF_REQUIRE(workspace_floats >= (size_t)kPointWsFloats, TF_ESHAPE, "tf_point_pack: workspace too small (%zu < %d floats)",
             workspace_floats, kPointWsFloats);
  static const int mat_out[3] = {1, 1, 3};
  TfPackBatch PB(stream);           // the 20 packs of the five per-point nets: ONE launch
  for (int n = 0; n < 3; ++n) {
    TF_REQUIRE(nets->mat_w1[n] && nets->mat_b1[n] && nets->mat_w2[n] && nets->mat_b2[n], TF_EINVAL,
               "tf_point_pack: null material weight pointer (net %d)", n);
    float* ws = workspace + n * kPmSize;
    PB.wfrag(nets->mat_w1[n], 128, 108, 0, 108, 4, PT_MAT_KS, ws + kPmW1);
    PB.bias(nets->mat_b1[n], 128, 4, ws + kPmB1);
    PB.wfrag(nets->mat_w2[n], mat_out[n], 128, 0, 128, 1, 64, ws + kPmW2);
    PB.bias(nets->mat_b2[n], mat_out[n], 1, ws + kPmB2);
  }
  for (int f = 0; f < 2; ++f) {
    TF_REQUIRE(nets->nis_w1[f] && nets->nis_b1[f] && nets->nis_w2[f] && nets->nis_b2[f], TF_EINVAL,
               "tf_point_pack: null flow feature-net weight pointer (flow %d)", f);
    float* ws = workspace + 3 * kPmSize + f * kPnSize;
    PB.wfrag(nets->nis_w1[f], 64, 57, 0, 57, 2, PT_NIS_KS, ws + kPnW1);
    PB.bias(nets->nis_b1[f], 64, 2, ws + kPnB1);
    PB.wfrag(nets->nis_w2[f], 16, 64, 0, 64, 1, 32, ws + kPnW2);
    PB.bias(nets->nis_b2[f], 16, 1, ws + kPnB2);
  }
  PB.flush();
  TF_LAUNCH_CHECK("tf_point_pack");
  return TF_OK;
}

extern "C" int tf_point_fwd(const float* workspace, const TfVmDesc* mat_desc, const float* mat_packed,
                            const TfVmDesc* flow_d_desc, const float* flow_d_packed, const TfVmDesc* flow_s_desc,
                            const float* flow_s_packed, const float* aabb_host, const float* pts, const float* view_angles,
                            int64_t pn, float rough_min, float* metallic, float* roughness, float* albedo, float* cond_d,
                            float* cond_s, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(pn >= 0, TF_ESHAPE, "tf_point_fwd: pn < 0");
  if (pn == 0) return TF_OK;
  TF_REQUIRE(workspace && mat_packed && flow_d_packed && flow_s_packed && aabb_host && pts && view_angles && metallic &&
                 roughness && albedo && cond_d && cond_s, TF_EINVAL, "tf_point_fwd: null pointer");
  PointArgs A;
  const TfVmDesc* descs[3] = {mat_desc, flow_d_desc, flow_s_desc};
  VmGeom* geoms[3] = {&A.gm, &A.gd, &A.gs};
  const int want_c[3] = {PT_MAT_C, PT_NIS_C, PT_NIS_C};
  for (int k = 0; k < 3; ++k) {
    int rc = vm_geom_init(descs[k], aabb_host, geoms[k]);
    TF_REQUIRE(rc != -1, TF_EINVAL, "tf_point_fwd: bad TfVmDesc %d", k);
    TF_REQUIRE(rc != -2, TF_ESHAPE, "tf_point_fwd: plane/line sizes > 1 must be divisible by 2^(n_levels-1) (field %d)", k);
    TF_REQUIRE(descs[k]->C == want_c[k], TF_ESHAPE, "tf_point_fwd: field %d has C=%d, this build instantiates C=%d", k,
               descs[k]->C, want_c[k]);
  }
  A.mat_packed = mat_packed; A.nis_packed[0] = flow_d_packed; A.nis_packed[1] = flow_s_packed;
  A.ws = workspace; A.pts = pts; A.va = view_angles; A.pn = pn; A.rough_min = rough_min;
  A.metallic = metallic; A.roughness = roughness; A.albedo = albedo; A.cond[0] = cond_d; A.cond[1] = cond_s;
  dim3 grid(tf_blocks(pn, 32), 5);
  point_prep_kernel<<<grid, 64, 0, stream>>>(A);
  TF_LAUNCH_CHECK("tf_point_fwd");
  return TF_OK;
}
