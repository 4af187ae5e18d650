// Split-sum shading of the shape stage, one launch per batch of live march samples:
// ShapeShadingNetwork.forward (network/fields.py:448-567, predict_specular_lights :419-439) with EnvLight.__call__
// (network/light.py:72-80, :95-122) over the pre-filtered cube-map stack of EnvLight.build_mips.
// Per sample:  mat_mlp 128-128-128-5 (sigmoid) -> albedo / roughness / metallic;  diffuse = (1-metal) albedo exp(diff(n));
//   direct = exp(lerp of two specular mips at reflect(v,n), mip from roughness);  inner_light [pos_enc8(p) 51, IDE5(refl, rough)
//   72] -> 123-128-128-3, exp(min(.,exp_max));  inner_weight [pos_enc8(p) 51, pos_enc6(refl) 39] -> 90-128-128-1 (occlusion
//   probability);  FG LUT (bilinear, clamp);  colour = sRGB(diffuse + (spec_albedo FG.x + FG.y) light).
// In the reference this is ~60 PyTorch launches over [N,123]-wide concatenations (the build's first version ran the
// three MLPs as library GEMMs: torch.cat alone cost more than the fused march kernel).
//
// Mapping: as the inner-light kernel of the material stage (inner_light.hip): a wave owns 32 samples on the MFMA column
// index, 4 waves per workgroup in lockstep, the three 128-wide nets run in f16x3 (3 x v_mfma_f32_32x32x16_f16 per fp32
// product term, fp32 accumulate) with their 374 KB of fragment-ordered weights streamed L2 -> LDS by LDS-DMA through the
// software-pipelined 4-slab ring of mfma_mlp.h; activations stay in accumulator registers; encodings are computed in
// registers (short-reduction sincos, IDE polynomial table in constant cache) and selected per lane half with v_cndmask.
#include <cmath>

#include "cube.h"
#include "mfma_mlp.h"
#include "tf_common.h"

// workspace (floats): f16x3 fragments [s16][tout][hi|lo][lane][8 halves] (512 floats per (s16, tout) pair), then biases
static constexpr int kSM1 = 0;                        // mat   128 -> 128 : 8 k16 x 4 tiles
static constexpr int kSM2 = kSM1 + 8 * 4 * 512;
static constexpr int kSM3 = kSM2 + 8 * 4 * 512;       // mat   128 -> 5   : 8 k16 x 1 tile
static constexpr int kSL1 = kSM3 + 8 * 512;           // light 123 -> 128 : 8 k16 x 4
static constexpr int kSL2 = kSL1 + 8 * 4 * 512;
static constexpr int kSL3 = kSL2 + 8 * 4 * 512;       // light 128 -> 3
static constexpr int kSW1 = kSL3 + 8 * 512;           // weight 90 -> 128 : 6 k16 x 4
static constexpr int kSW2 = kSW1 + 6 * 4 * 512;
static constexpr int kSW3 = kSW2 + 8 * 4 * 512;       // weight 128 -> 1
static constexpr int kSBias = kSW3 + 8 * 512;         // 3 nets x {b1 [128], b2 [128], b3 [32]} in accumulator order
static constexpr int kSIde = kSBias + 3 * (128 + 128 + 32);   // [17][36] IDE polynomial coefficients
static constexpr int kShapeWsFloats = ((kSIde + 17 * 36 + 1023) / 1024) * 1024;

extern "C" size_t tf_shape_shade_workspace_floats(void) { return kShapeWsFloats; }

struct ShapeArgs {
  const float* ws;
  const float* spec[8];
  int spec_res[8];
  int n_spec;
  const float* diff;
  int diff_res;
  const float* fg;     // [H,W,2]
  int fg_h, fg_w;
  float min_r, max_r, exp_max;
  const float* pts;
  const float* normals;
  const float* view;
  const float* feat;   // [n,128]
  long long n;
  float* color;
  float* occ;
  float* rough;
  float* refl;
};

// Ref-NeRF IDE tables, as in inner_light.hip (utils/ref_utils.py:8-78)
static void shape_ide_tables_host(float* mat /*[17][36]*/) {
  auto fact = [](int n) { double r = 1; for (int i = 2; i <= n; ++i) r *= i; return r; };
  int col = 0;
  for (int i = 0; i < 17 * 36; ++i) mat[i] = 0.f;
  for (int d = 0; d < 5; ++d) {
    const int l = 1 << d;
    for (int m = 0; m <= l; ++m, ++col) {
      for (int k = 0; k <= l - m; ++k) {
        const double a = 0.5 * (l + k + m - 1.0);
        double gb = 1.0;
        for (int j = 0; j < l; ++j) gb *= (a - j);
        gb /= fact(l);
        const double leg = std::pow(-1.0, m) * std::pow(2.0, l) * fact(l) / fact(k) / fact(l - k - m) * gb;
        mat[k * 36 + col] = (float)(std::sqrt((2.0 * l + 1.0) * fact(l - m) / (4.0 * M_PI * fact(l + m))) * leg);
      }
    }
  }
}

// The nine layers' fragment images are one continuous weight stream (mfma_mlp.h TfStream), consumed in workspace order:
//   mat 4 + 4 + 1 | light 4 + 4 + 1 | weight 3 + 4 + 1 slabs = 26 per tile.  P0 = parity of the layer's first slab.
// hidden layer: out = relu(W in + b), 128 outputs (4 tiles)
// `bias`: this lane half's biases in accumulator order, in LDS (broadcast b128 reads; see inner_light.hip)
template <int K16, int TIN, int P0>
__device__ __forceinline__ void ss_hidden(TfStream& S, TfFrag& FA, TfFrag& FB, const float* __restrict__ bias, int h,
                                          const f32x16 (&in)[TIN], f32x16 (&out)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v4 = *reinterpret_cast<const float4*>(bias + t * 16 + 4 * q);
      out[t][4 * q] = v4.x; out[t][4 * q + 1] = v4.y; out[t][4 * q + 2] = v4.z; out[t][4 * q + 3] = v4.w;
    }
  tf_layer_h3s<K16, 4, TIN, P0>(S, FA, FB, in, out);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) out[t][j] = tf_relu(out[t][j]);
}

template <int P0>
__device__ __forceinline__ void ss_out(TfStream& S, TfFrag& FA, TfFrag& FB, const float* __restrict__ bias, int h,
                                       const f32x16 (&in)[4], f32x16 (&o)[1]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) o[0][j] = bias[j];
  tf_layer_h3s<8, 1, 4, P0>(S, FA, FB, in, o);
}

// operand slots of this lane half: in1[t][j] <- enc[32 t + (j & 3) + 8 (j >> 2) + 4 h]   (explicit v_cndmask: see inner_light.hip)
template <int TIN>
__device__ __forceinline__ void ss_select(const float (&enc)[32 * TIN], f32x16 (&in1)[TIN]) {
  const unsigned long long upper_half = 0xFFFFFFFF00000000ULL;
#pragma unroll
  for (int t = 0; t < TIN; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k0 = 32 * t + (j & 3) + 8 * (j >> 2);
      asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(in1[t][j]) : "v"(enc[k0]), "v"(enc[k0 + 4]), "s"(upper_half));
    }
}

__global__ void __launch_bounds__(256) shape_shade_kernel(ShapeArgs A) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 4096];   // weight-slab ring
  const int tid = threadIdx.x, lane = threadIdx.x & 63, h = lane >> 5;
  const long long n_groups = (A.n + 127) / 128;   // a workgroup advances 4 tiles (128 samples) in lockstep
  // biases, re-laid-out per lane half: lbias[net][half][b1 64 | b2 64 | (pad) | b3 16 at 256] <- packed[(n) * 2 + half]
  __shared__ __attribute__((aligned(16))) float lbias[3 * 2 * 272];
  for (int i = tid; i < 3 * 288; i += 256) {
    const int net = i / 288, r = i % 288;
    const int seg = r < 256 ? 0 : 1, rr = seg ? r - 256 : r;       // rr = n * 2 + half within the segment
    lbias[net * 544 + (rr & 1) * 272 + (seg ? 256 : 0) + (rr >> 1)] = A.ws[kSBias + i];
  }
  TfStream S;
  TfFrag FA, FB;
  tf_stream_begin(S, reinterpret_cast<const _Float16*>(A.ws + kSM1), 26, lds, tid, lane, FA);   // contains a barrier
  for (long long tg = blockIdx.x; tg < n_groups; tg += gridDim.x) {
    const float* ws = A.ws;
    asm volatile("" : "+s"(ws));   // keep biases / tables / slab addresses from being hoisted out of the tile loop and spilled
    const long long tile = tg * 4 + (threadIdx.x >> 6);
    long long row = tile * 32 + (lane & 31);
    const bool valid = row < A.n;
    if (!valid) row = A.n - 1;
    // ---------------- geometry (fields.py:453-461)
    float nx = A.normals[3 * row], ny = A.normals[3 * row + 1], nz = A.normals[3 * row + 2];
    float inv = 1.f / fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
    nx *= inv; ny *= inv; nz *= inv;
    if (nx + ny == 0.f) { nx = 0.f; ny = 1e-6f; nz = 1.f; }
    float vx = A.view[3 * row], vy = A.view[3 * row + 1], vz = A.view[3 * row + 2];
    inv = 1.f / fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
    vx *= inv; vy *= inv; vz *= inv;
    const float NoV = nx * vx + ny * vy + nz * vz;
    const float rx = NoV * nx * 2.f - vx, ry = NoV * ny * 2.f - vy, rz = NoV * nz * 2.f - vz;
    const float p[3] = {A.pts[3 * row], A.pts[3 * row + 1], A.pts[3 * row + 2]};

    // ---------------- material MLP on the 128 appearance features
    f32x16 a[4], b[4], o[1];
    {
      f32x16 in1[4];
      const float* frow = A.feat + row * 128 + 4 * h;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
          const float4 v4 = *reinterpret_cast<const float4*>(frow + 32 * t + 8 * jq);
          in1[t][4 * jq] = v4.x; in1[t][4 * jq + 1] = v4.y; in1[t][4 * jq + 2] = v4.z; in1[t][4 * jq + 3] = v4.w;
        }
      ss_hidden<8, 4, 0>(S, FA, FB, lbias + h * 272, h, in1, a);
    }
    ss_hidden<8, 4, 0>(S, FA, FB, lbias + h * 272 + 64, h, a, b);
    ss_out<0>(S, FA, FB, lbias + h * 272 + 256, h, b, o);
    // units 0..3 sit in registers 0..3 of lane half 0, unit 4 in register 0 of lane half 1
    float albedo[3], rough, metal;
    {
      float s[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) s[c] = 1.f / (1.f + expf(-o[0][c]));
      const float x0 = __shfl_xor(s[0], 32), x3 = __shfl_xor(s[3], 32);
      metal = h ? s[0] : x0;
      rough = (h ? x3 : s[3]) * 0.9f + 0.09f;
#pragma unroll
      for (int c = 0; c < 3; ++c) albedo[c] = s[c] * 0.77f + 0.03f;   // meaningful in lane half 0 (the half that writes)
    }

    // ---------------- environment light (light.py:72-80, :95-122)
    float diffuse[3], direct[3];
    {
      float r, g, bb;
      cube_fetch_rgb(A.diff, A.diff_res, nx, ny, nz, r, g, bb);
      diffuse[0] = (1.f - metal) * albedo[0] * expf(r);
      diffuse[1] = (1.f - metal) * albedo[1] * expf(g);
      diffuse[2] = (1.f - metal) * albedo[2] * expf(bb);
      const int ns = A.n_spec;
      float mip = rough < A.max_r ? (fminf(fmaxf(rough, A.min_r), A.max_r) - A.min_r) / (A.max_r - A.min_r) * (float)(ns - 2)
                                  : (fminf(fmaxf(rough, A.max_r), 1.f) - A.max_r) / (1.f - A.max_r) + (float)(ns - 2);
      mip = fminf(fmaxf(mip, 0.f), (float)(ns - 1));
      const float fl0 = fminf(floorf(mip), (float)(ns - 1));
      float f = mip - fl0;
      const int l0 = (int)fl0, l1 = min(l0 + 1, ns - 1);
      if (l1 == l0) f = 0.f;
      float acc3[3] = {0.f, 0.f, 0.f};
      for (int li = 0; li < ns; ++li) {   // wave-uniform loop over the stack; a lane takes the (at most two) levels it touches
        const float w = (l0 == li ? 1.f - f : 0.f) + ((l1 == li && l0 != li) ? f : 0.f);
        if (__any(w != 0.f)) {
          cube_fetch_rgb(A.spec[li], A.spec_res[li], rx, ry, rz, r, g, bb);
          acc3[0] += w * r; acc3[1] += w * g; acc3[2] += w * bb;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) direct[c] = expf(acc3[c]);
    }

    // ---------------- encodings shared by the two light nets
    float enc[128];
#pragma unroll
    for (int k = 0; k < 3; ++k) enc[k] = p[k];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int k = 0; k < 3; ++k) tf_sincos_small(p[k] * (float)(1 << f), enc[3 + 6 * f + k], enc[3 + 6 * f + 3 + k]);
    f32x16 il_in[4], iw_in[3];
    {
      // inner_weight input: [pos_enc8(p) 51 | pos_enc6(refl) 39 | pad 6]
      float e2[96];
#pragma unroll
      for (int k = 0; k < 51; ++k) e2[k] = enc[k];
      const float rr[3] = {rx, ry, rz};
#pragma unroll
      for (int k = 0; k < 3; ++k) e2[51 + k] = rr[k];
#pragma unroll
      for (int f = 0; f < 6; ++f)
#pragma unroll
        for (int k = 0; k < 3; ++k) tf_sincos_small(rr[k] * (float)(1 << f), e2[54 + 6 * f + k], e2[54 + 6 * f + 3 + k]);
#pragma unroll
      for (int k = 90; k < 96; ++k) e2[k] = 0.f;
      ss_select<3>(e2, iw_in);
    }
    {
      // inner_light input: [pos_enc8(p) 51 | IDE5(refl, rough) 72 | pad 5];  IDE: sph_i = (rx + i ry)^m_i sum_k mat[k][i] rz^k,
      // attenuated by exp(-l(l+1)/2 * rough)  (ref_utils.py:103-117)
      float zp[17], cre[17], cim[17];
      zp[0] = 1.f; cre[0] = 1.f; cim[0] = 0.f;
#pragma unroll
      for (int k = 1; k < 17; ++k) {
        zp[k] = zp[k - 1] * rz;
        cre[k] = cre[k - 1] * rx - cim[k - 1] * ry;
        cim[k] = cre[k - 1] * ry + cim[k - 1] * rx;
      }
      // wave-uniform table: scalar-cache reads (see inner_light.hip)
      const __attribute__((address_space(4))) float* mat = (const __attribute__((address_space(4))) float*)(unsigned long long)(ws + kSIde);
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        const float att = expf(-0.5f * (float)((1 << d) * ((1 << d) + 1)) * rough);
#pragma unroll
        for (int mm = 0; mm <= (1 << d); ++mm) {
          const int col = (1 << d) - 1 + d + mm;
          float poly = 0.f;
#pragma unroll
          for (int k = 0; k <= (1 << d) - mm; ++k) poly += zp[k] * mat[k * 36 + col];
          enc[51 + col] = cre[mm] * poly * att;
          enc[51 + 36 + col] = cim[mm] * poly * att;
        }
      }
#pragma unroll
      for (int k = 123; k < 128; ++k) enc[k] = 0.f;
      ss_select<4>(enc, il_in);
    }

    // ---------------- inner_light -> indirect radiance
    float indirect[3];
    ss_hidden<8, 4, 1>(S, FA, FB, lbias + 544 + h * 272, h, il_in, a);
    ss_hidden<8, 4, 1>(S, FA, FB, lbias + 544 + h * 272 + 64, h, a, b);
    ss_out<1>(S, FA, FB, lbias + 544 + h * 272 + 256, h, b, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) indirect[c] = expf(fminf(o[0][c], A.exp_max));
    // ---------------- inner_weight -> occlusion probability
    ss_hidden<6, 3, 0>(S, FA, FB, lbias + 1088 + h * 272, h, iw_in, a);
    ss_hidden<8, 4, 1>(S, FA, FB, lbias + 1088 + h * 272 + 64, h, a, b);
    ss_out<1>(S, FA, FB, lbias + 1088 + h * 272 + 256, h, b, o);
    const float occ = o[0][0] * 0.5f + 0.5f;

    // ---------------- combine (lane half 0 holds albedo / indirect / occ)
    if (valid && h == 0) {
      const float occ_c = fminf(fmaxf(occ, 0.f), 1.f);
      // FG LUT: F.grid_sample(bilinear, border, align_corners=False) at (NoV, roughness)   (fields.py:346, :528-531)
      const float u = fminf(fmaxf(fminf(fmaxf(NoV, 0.f), 1.f) * (float)A.fg_w - 0.5f, 0.f), (float)(A.fg_w - 1));
      const float v = fminf(fmaxf(fminf(fmaxf(rough, 0.f), 1.f) * (float)A.fg_h - 0.5f, 0.f), (float)(A.fg_h - 1));
      const float fu0 = floorf(u), fv0 = floorf(v);
      const float fu = u - fu0, fv = v - fv0;
      const int x0 = (int)fu0, y0 = (int)fv0, x1 = min(x0 + 1, A.fg_w - 1), y1 = min(y0 + 1, A.fg_h - 1);
      const float2 t00 = *reinterpret_cast<const float2*>(A.fg + 2LL * (y0 * A.fg_w + x0));
      const float2 t10 = *reinterpret_cast<const float2*>(A.fg + 2LL * (y0 * A.fg_w + x1));
      const float2 t01 = *reinterpret_cast<const float2*>(A.fg + 2LL * (y1 * A.fg_w + x0));
      const float2 t11 = *reinterpret_cast<const float2*>(A.fg + 2LL * (y1 * A.fg_w + x1));
      const float w00 = (1.f - fu) * (1.f - fv), w10 = fu * (1.f - fv), w01 = (1.f - fu) * fv, w11 = fu * fv;
      const float fg0 = t00.x * w00 + t10.x * w10 + t01.x * w01 + t11.x * w11;
      const float fg1 = t00.y * w00 + t10.y * w10 + t01.y * w01 + t11.y * w11;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float light = indirect[c] * occ_c + direct[c] * (1.f - occ_c);
        const float spec_alb = 0.04f * (1.f - metal) + metal * albedo[c];
        const float lin = diffuse[c] + (spec_alb * fg0 + fg1) * light;
        // linear_to_srgb (utils/raw_utils.py:4-17), then clamp to [0,1]
        const float srgb = lin <= 0.0031308f ? (323.f / 25.f) * lin
                                             : (211.f * powf(fmaxf(lin, 1.1920928955078125e-07f), 5.f / 12.f) - 11.f) / 200.f;
        A.color[3 * row + c] = fminf(fmaxf(srgb, 0.f), 1.f);
      }
      if (A.occ) A.occ[row] = occ;
      if (A.rough) A.rough[row] = rough;
      if (A.refl) { A.refl[3 * row] = rx; A.refl[3 * row + 1] = ry; A.refl[3 * row + 2] = rz; }
    }
  }
  tf_stream_end();
}

extern "C" int tf_shape_shade_pack(const TfShapeNets* nets, float* workspace, size_t workspace_floats, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(nets && workspace, TF_EINVAL, "tf_shape_shade_pack: null pointer");
  TF_REQUIRE(workspace_floats >= (size_t)kShapeWsFloats, TF_ESHAPE, "tf_shape_shade_pack: workspace too small (%zu < %d floats)",
             workspace_floats, kShapeWsFloats);
  const TfMlp3* net[3] = {&nets->mat_mlp, &nets->inner_light, &nets->inner_weight};
  const int kin[3] = {128, 123, 90}, k16[3] = {8, 8, 6}, nout[3] = {5, 3, 1};
  const int off1[3] = {kSM1, kSL1, kSW1}, off2[3] = {kSM2, kSL2, kSW2}, off3[3] = {kSM3, kSL3, kSW3};
  _Float16* hw = reinterpret_cast<_Float16*>(workspace);
  for (int i = 0; i < 3; ++i) {
    for (int l = 0; l < 3; ++l)
      TF_REQUIRE(net[i]->w[l] && net[i]->b[l], TF_EINVAL, "tf_shape_shade_pack: null weight pointer (net %d layer %d)", i, l);
    tf_pack_wfrag_h3_kernel<<<tf_blocks(4 * k16[i] * 64, 256), 256, 0, stream>>>(net[i]->w[0], 128, kin[i], 0, kin[i], 4, k16[i],
                                                                                hw + 2 * (size_t)off1[i]);
    tf_pack_wfrag_h3_kernel<<<tf_blocks(4 * 8 * 64, 256), 256, 0, stream>>>(net[i]->w[1], 128, 128, 0, 128, 4, 8, hw + 2 * (size_t)off2[i]);
    tf_pack_wfrag_h3_kernel<<<tf_blocks(1 * 8 * 64, 256), 256, 0, stream>>>(net[i]->w[2], nout[i], 128, 0, 128, 1, 8, hw + 2 * (size_t)off3[i]);
    float* bb = workspace + kSBias + i * 288;
    tf_pack_bias_kernel<<<1, 256, 0, stream>>>(net[i]->b[0], 128, 4, bb);
    tf_pack_bias_kernel<<<1, 256, 0, stream>>>(net[i]->b[1], 128, 4, bb + 128);
    tf_pack_bias_kernel<<<1, 256, 0, stream>>>(net[i]->b[2], nout[i], 1, bb + 256);
  }
  // computed once per process (C++11 magic static: thread-safe), read-only afterwards
  struct IdeTable { float v[17 * 36]; IdeTable() { shape_ide_tables_host(v); } };
  static const IdeTable ide_table;
  hipError_t e = hipMemcpyAsync(workspace + kSIde, ide_table.v, sizeof(ide_table.v), hipMemcpyHostToDevice, stream);
  TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_shape_shade_pack: hipMemcpyAsync failed: %s", hipGetErrorString(e));
  TF_LAUNCH_CHECK("tf_shape_shade_pack");
  return TF_OK;
}

extern "C" int tf_shape_shade_fwd(const float* workspace, const float* const* spec_mips, const int32_t* spec_res, int32_t n_spec,
                                  const float* diffuse_map, int32_t diffuse_res, const float* fg_lut, int32_t fg_h, int32_t fg_w,
                                  float min_roughness, float max_roughness, float light_exp_max, const float* pts,
                                  const float* normals, const float* view, const float* feat, int64_t n, float* color,
                                  float* occ, float* roughness, float* refl, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_shape_shade_fwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(workspace && spec_mips && spec_res && diffuse_map && fg_lut && pts && normals && view && feat && color, TF_EINVAL,
             "tf_shape_shade_fwd: null pointer");
  TF_REQUIRE(n_spec >= 2 && n_spec <= 8, TF_ESHAPE, "tf_shape_shade_fwd: need 2..8 specular mips (got %d)", n_spec);
  TF_REQUIRE(diffuse_res > 0 && fg_h > 0 && fg_w > 0, TF_ESHAPE, "tf_shape_shade_fwd: bad texture size");
  TF_REQUIRE((((uintptr_t)feat) & 15) == 0, TF_EINVAL, "tf_shape_shade_fwd: feat must be 16-byte aligned");
  ShapeArgs A = {};
  A.ws = workspace;
  for (int i = 0; i < n_spec; ++i) {
    TF_REQUIRE(spec_mips[i] && spec_res[i] > 0, TF_EINVAL, "tf_shape_shade_fwd: null / empty specular mip %d", i);
    A.spec[i] = spec_mips[i]; A.spec_res[i] = spec_res[i];
  }
  A.n_spec = n_spec; A.diff = diffuse_map; A.diff_res = diffuse_res; A.fg = fg_lut; A.fg_h = fg_h; A.fg_w = fg_w;
  A.min_r = min_roughness; A.max_r = max_roughness; A.exp_max = light_exp_max;
  A.pts = pts; A.normals = normals; A.view = view; A.feat = feat; A.n = n;
  A.color = color; A.occ = occ; A.rough = roughness; A.refl = refl;
  long long blocks = (n + 127) / 128;
  if (blocks > 1024) blocks = 1024;
  shape_shade_kernel<<<(unsigned)blocks, 256, 0, stream>>>(A);
  TF_LAUNCH_CHECK("tf_shape_shade_fwd");
  return TF_OK;
}
