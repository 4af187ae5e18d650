// The element-wise algebra of ShapeShadingNetwork.forward in the TRAINING direction (network/fields.py:448-567), as two differentiable
// stages around the nets / cube lookups / encodings, which are ops of their own (tf_linear_*, tf_cube_lookup*, tf_ide5, tf_posenc):
//   pre :  normals, view -> unit normals (with the reference's degenerate-row patch, :455-457), unit view, NoV, reflective (:458-459);
//          mat -> roughness (:463)
//   post:  mat (5 sigmoid outputs), NoV, diffuse / direct / indirect light, occlusion logit -> colour (:460-561: albedo / metallic /
//          roughness remap, diffuse and specular albedo, occlusion blend, FG LUT fetch, sRGB, clamp) and occ_prob
// and their adjoints.  Round 5's profile of the shape training step: ~385 of 526 launches were 4-10 us torch element-wise kernels, about
// 150 of them this composition and autograd's mirror image of it; here it is 4 launches.  Derivative conventions are torch autograd's of
// the composition this replaces (F.normalize, clamp masks, F.grid_sample(bilinear, border, align_corners=False) incl. its zero gradient on
// clipped coordinates, the sRGB branch taken): the reference-run gradient goldens are the test.
#include "tf_common.h"

struct GluePre {
  const float* n_raw; const float* v_raw; const float* mat;       // [N,3] [N,3] [N,5]
  long long n;
  float* n_u; float* v_u; float* nov; float* refl; float* rough;  // [N,3] [N,3] [N] [N,3] [N]
  float* mip; float min_r, max_r; int n_levels;                   // optional: EnvLight.get_mip(roughness).clamp(0, n - 1) (light.py:72-80, :101)
};

// get_mip: roughness -> coordinate in the specular stack (two linear pieces meeting at max_roughness), clamped to [0, n - 1]; *d = its
// derivative under torch's conventions (torch.where routes the gradient to the branch taken, clamp passes it on [min, max] inclusive)
__device__ __forceinline__ float glue_mip(float r, float min_r, float max_r, int n, float* d) {
  float m, dm;
  if (r < max_r) {
    m = (fminf(fmaxf(r, min_r), max_r) - min_r) / (max_r - min_r) * (float)(n - 2);
    dm = (r >= min_r && r <= max_r) ? (float)(n - 2) / (max_r - min_r) : 0.f;
  } else {
    m = (fminf(fmaxf(r, max_r), 1.f) - max_r) / (1.f - max_r) + (float)(n - 2);
    dm = (r >= max_r && r <= 1.f) ? 1.f / (1.f - max_r) : 0.f;
  }
  if (!(m >= 0.f && m <= (float)(n - 1))) dm = 0.f;
  if (d) *d = dm;
  return fminf(fmaxf(m, 0.f), (float)(n - 1));
}

__device__ __forceinline__ float glue_norm3(float x, float y, float z) { return fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f); }

__global__ void __launch_bounds__(256) shape_glue_pre_fwd_kernel(GluePre A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n) return;
  float nx = A.n_raw[3 * i], ny = A.n_raw[3 * i + 1], nz = A.n_raw[3 * i + 2];
  const float ln = glue_norm3(nx, ny, nz);
  nx /= ln; ny /= ln; nz /= ln;
  if (nx + ny == 0.f) { nx = 0.f; ny = 1e-6f; nz = 1.f; }      // torch.where(normals[:, :2].sum(-1) == 0, (0, 1e-6, 1), normals)
  float vx = A.v_raw[3 * i], vy = A.v_raw[3 * i + 1], vz = A.v_raw[3 * i + 2];
  const float lv = glue_norm3(vx, vy, vz);
  vx /= lv; vy /= lv; vz /= lv;
  const float nov = nx * vx + ny * vy + nz * vz;
  A.n_u[3 * i] = nx; A.n_u[3 * i + 1] = ny; A.n_u[3 * i + 2] = nz;
  A.v_u[3 * i] = vx; A.v_u[3 * i + 1] = vy; A.v_u[3 * i + 2] = vz;
  A.nov[i] = nov;
  A.refl[3 * i] = nov * nx * 2.f - vx; A.refl[3 * i + 1] = nov * ny * 2.f - vy; A.refl[3 * i + 2] = nov * nz * 2.f - vz;
  const float rough = A.mat[5 * i + 3] * 0.9f + 0.09f;
  A.rough[i] = rough;
  if (A.mip) A.mip[i] = glue_mip(rough, A.min_r, A.max_r, A.n_levels, nullptr);
}

struct GluePreBwd {
  const float* n_raw; const float* v_raw;
  const float* g_nu; const float* g_nov; const float* g_refl; const float* g_rough;   // any may be NULL (no gradient arrived)
  const float* g_mip; const float* mat; float min_r, max_r; int n_levels;             // g_mip non-NULL: the mip coordinate's gradient (needs mat)
  long long n;
  float* g_n_raw; float* g_mat;     // [N,3]; [N,5]: column 3 written, the others zeroed
};

__global__ void __launch_bounds__(256) shape_glue_pre_bwd_kernel(GluePreBwd A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n) return;
  const float rx = A.n_raw[3 * i], ry = A.n_raw[3 * i + 1], rz = A.n_raw[3 * i + 2];
  const float ln = glue_norm3(rx, ry, rz);
  float nx = rx / ln, ny = ry / ln, nz = rz / ln;
  const bool bad = nx + ny == 0.f;
  if (bad) { nx = 0.f; ny = 1e-6f; nz = 1.f; }
  float vx = A.v_raw[3 * i], vy = A.v_raw[3 * i + 1], vz = A.v_raw[3 * i + 2];
  const float lv = glue_norm3(vx, vy, vz);
  vx /= lv; vy /= lv; vz /= lv;
  const float nov = nx * vx + ny * vy + nz * vz;
  float gx = 0.f, gy = 0.f, gz = 0.f, gnov = A.g_nov ? A.g_nov[i] : 0.f;
  if (A.g_nu) { gx = A.g_nu[3 * i]; gy = A.g_nu[3 * i + 1]; gz = A.g_nu[3 * i + 2]; }
  if (A.g_refl) {
    const float ax = A.g_refl[3 * i], ay = A.g_refl[3 * i + 1], az = A.g_refl[3 * i + 2];
    gnov += 2.f * (ax * nx + ay * ny + az * nz);       // reflective = 2 NoV n - v
    gx += 2.f * nov * ax; gy += 2.f * nov * ay; gz += 2.f * nov * az;
  }
  gx += gnov * vx; gy += gnov * vy; gz += gnov * vz;   // NoV = n . v
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (!bad) {                                          // a patched row is a constant: nothing flows back; F.normalize: (g - n (n . g)) / |x|
    const float d = gx * nx + gy * ny + gz * nz;
    // (|x| below the eps of F.normalize: x / eps, derivative g / eps)
    const bool tiny = sqrtf(rx * rx + ry * ry + rz * rz) < 1e-12f;
    ox = tiny ? gx / ln : (gx - nx * d) / ln; oy = tiny ? gy / ln : (gy - ny * d) / ln; oz = tiny ? gz / ln : (gz - nz * d) / ln;
  }
  A.g_n_raw[3 * i] = ox; A.g_n_raw[3 * i + 1] = oy; A.g_n_raw[3 * i + 2] = oz;
  A.g_mat[5 * i] = 0.f; A.g_mat[5 * i + 1] = 0.f; A.g_mat[5 * i + 2] = 0.f; A.g_mat[5 * i + 4] = 0.f;
  float gr = A.g_rough ? A.g_rough[i] : 0.f;
  if (A.g_mip) {
    float dm;
    glue_mip(A.mat[5 * i + 3] * 0.9f + 0.09f, A.min_r, A.max_r, A.n_levels, &dm);
    gr += A.g_mip[i] * dm;
  }
  A.g_mat[5 * i + 3] = 0.9f * gr;
}

struct GluePost {
  const float* mat; const float* nov; const float* diffuse; const float* direct; const float* indirect; const float* occ_raw;
  const float* fg; int fg_h, fg_w;
  long long n;
  float* color; float* occ_prob;                       // forward outputs
  const float* g_color; const float* g_occ_prob;       // backward inputs (g_occ_prob may be NULL)
  float* g_mat; float* g_nov; float* g_diffuse; float* g_direct; float* g_indirect; float* g_occ_raw;
};

struct GlueFg {
  float fg0, fg1;        // the two LUT channels at (NoV, roughness)
  float d0u, d1u, d0v, d1v;   // their derivatives wrt NoV / roughness (0 where grid_sample's border clip or the [0,1] clamp cuts them)
};

__device__ __forceinline__ GlueFg glue_fg(const float* __restrict__ fg, int H, int W, float nov, float rough) {
  // uv = (clamp(NoV, 0, 1), clamp(rough, 0, 1)); grid = 2 uv - 1; grid_sample(align_corners=False): ix = uv.x W - 0.5, clipped to [0, W - 1]
  const float cu = fminf(fmaxf(nov, 0.f), 1.f), cv = fminf(fmaxf(rough, 0.f), 1.f);
  const float iu = cu * (float)W - 0.5f, iv = cv * (float)H - 0.5f;
  const float u = fminf(fmaxf(iu, 0.f), (float)(W - 1)), v = fminf(fmaxf(iv, 0.f), (float)(H - 1));
  // clip_coordinates_set_grad: zero at and beyond the border texel centres; torch.clamp passes the gradient on [min, max] inclusive
  const float mu = (iu > 0.f && iu < (float)(W - 1) && nov >= 0.f && nov <= 1.f) ? (float)W : 0.f;
  const float mv = (iv > 0.f && iv < (float)(H - 1) && rough >= 0.f && rough <= 1.f) ? (float)H : 0.f;
  const float fu0 = floorf(u), fv0 = floorf(v);
  const float fu = u - fu0, fv = v - fv0;
  const int x0 = (int)fu0, y0 = (int)fv0, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  const float2 t00 = *reinterpret_cast<const float2*>(fg + 2LL * (y0 * W + x0)), t10 = *reinterpret_cast<const float2*>(fg + 2LL * (y0 * W + x1));
  const float2 t01 = *reinterpret_cast<const float2*>(fg + 2LL * (y1 * W + x0)), t11 = *reinterpret_cast<const float2*>(fg + 2LL * (y1 * W + x1));
  const float w00 = (1.f - fu) * (1.f - fv), w10 = fu * (1.f - fv), w01 = (1.f - fu) * fv, w11 = fu * fv;
  GlueFg R;
  R.fg0 = t00.x * w00 + t10.x * w10 + t01.x * w01 + t11.x * w11;
  R.fg1 = t00.y * w00 + t10.y * w10 + t01.y * w01 + t11.y * w11;
  R.d0u = ((t10.x - t00.x) * (1.f - fv) + (t11.x - t01.x) * fv) * mu;
  R.d1u = ((t10.y - t00.y) * (1.f - fv) + (t11.y - t01.y) * fv) * mu;
  R.d0v = ((t01.x - t00.x) * (1.f - fu) + (t11.x - t10.x) * fu) * mv;
  R.d1v = ((t01.y - t00.y) * (1.f - fu) + (t11.y - t10.y) * fu) * mv;
  return R;
}

template <bool BWD>
__global__ void __launch_bounds__(256) shape_glue_post_kernel(GluePost A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n) return;
  const float m0 = A.mat[5 * i], m1 = A.mat[5 * i + 1], m2 = A.mat[5 * i + 2], m3 = A.mat[5 * i + 3], metal = A.mat[5 * i + 4];
  const float alb[3] = {m0 * 0.77f + 0.03f, m1 * 0.77f + 0.03f, m2 * 0.77f + 0.03f};
  const float rough = m3 * 0.9f + 0.09f;
  const float nov = A.nov[i];
  const float occ_p = A.occ_raw[i] * 0.5f + 0.5f;
  const float occ = fminf(fmaxf(occ_p, 0.f), 1.f);
  const GlueFg F = glue_fg(A.fg, A.fg_h, A.fg_w, nov, rough);
  const float eps = 1.1920928955078125e-07f;
  float g_m[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, g_nov = 0.f, g_occ = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dl = A.diffuse[3 * i + c], dr = A.direct[3 * i + c], il = A.indirect[3 * i + c];
    const float dalb = (1.f - metal) * alb[c];
    const float salb = 0.04f * (1.f - metal) + metal * alb[c];
    const float light = il * occ + dr * (1.f - occ);
    const float sref = salb * F.fg0 + F.fg1;
    const float lin = dalb * dl + sref * light;
    const float y = lin <= 0.0031308f ? 12.92f * lin : (211.f * powf(fmaxf(lin, eps), 5.f / 12.f) - 11.f) / 200.f;
    if (!BWD) {
      A.color[3 * i + c] = fminf(fmaxf(y, 0.f), 1.f);
    } else {
      float d = lin <= 0.0031308f ? 12.92f : (lin >= eps ? (211.f / 200.f) * (5.f / 12.f) * powf(lin, -7.f / 12.f) : 0.f);
      if (!(y >= 0.f && y <= 1.f)) d = 0.f;
      const float gl = A.g_color[3 * i + c] * d;           // d loss / d lin
      A.g_diffuse[3 * i + c] = gl * dalb;
      const float g_light = gl * sref;
      A.g_indirect[3 * i + c] = g_light * occ;
      A.g_direct[3 * i + c] = g_light * (1.f - occ);
      g_occ += g_light * (il - dr);
      const float g_sref = gl * light;
      const float g_salb = g_sref * F.fg0;
      const float g_dalb = gl * dl;
      // d / d albedo_c, d / d metallic
      g_m[c] = 0.77f * (g_dalb * (1.f - metal) + g_salb * metal);
      g_m[4] += -g_dalb * alb[c] + g_salb * (alb[c] - 0.04f);
      // through the LUT: fg0 (scaled by the specular albedo) and fg1
      g_nov += g_sref * (salb * F.d0u + F.d1u);
      g_m[3] += g_sref * (salb * F.d0v + F.d1v) * 0.9f;
    }
  }
  if (!BWD) {
    A.occ_prob[i] = occ_p;
  } else {
    float g_raw = (occ_p >= 0.f && occ_p <= 1.f) ? 0.5f * g_occ : 0.f;
    if (A.g_occ_prob) g_raw += 0.5f * A.g_occ_prob[i];
    A.g_occ_raw[i] = g_raw;
    A.g_nov[i] = g_nov;
#pragma unroll
    for (int k = 0; k < 5; ++k) A.g_mat[5 * i + k] = g_m[k];
  }
}

extern "C" int tf_shape_glue_pre_fwd(const float* normals, const float* view, const float* mat, int64_t n, float* normals_u, float* view_u,
                                     float* nov, float* reflective, float* roughness, float* mip, float min_roughness, float max_roughness,
                                     int32_t n_levels, tf_stream_t stream) {
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_shape_glue_pre_fwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(normals && view && mat && normals_u && view_u && nov && reflective && roughness, TF_EINVAL, "tf_shape_glue_pre_fwd: null pointer");
  TF_REQUIRE(!mip || (n_levels >= 2 && min_roughness < max_roughness && max_roughness < 1.f), TF_EINVAL,
             "tf_shape_glue_pre_fwd: mip needs n_levels >= 2 and min_roughness < max_roughness < 1");
  GluePre A{normals, view, mat, n, normals_u, view_u, nov, reflective, roughness, mip, min_roughness, max_roughness, n_levels};
  shape_glue_pre_fwd_kernel<<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(A);
  TF_LAUNCH_CHECK("tf_shape_glue_pre_fwd");
  return TF_OK;
}

extern "C" int tf_shape_glue_pre_bwd(const float* normals, const float* view, const float* g_normals_u, const float* g_nov, const float* g_reflective,
                                     const float* g_roughness, const float* g_mip, const float* mat, float min_roughness, float max_roughness,
                                     int32_t n_levels, int64_t n, float* g_normals, float* g_mat, tf_stream_t stream) {
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_shape_glue_pre_bwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(normals && view && g_normals && g_mat, TF_EINVAL, "tf_shape_glue_pre_bwd: null pointer");
  TF_REQUIRE(!g_mip || (mat && n_levels >= 2 && min_roughness < max_roughness && max_roughness < 1.f), TF_EINVAL,
             "tf_shape_glue_pre_bwd: g_mip needs mat, n_levels >= 2 and min_roughness < max_roughness < 1");
  GluePreBwd A{normals, view, g_normals_u, g_nov, g_reflective, g_roughness, g_mip, mat, min_roughness, max_roughness, n_levels, n, g_normals, g_mat};
  shape_glue_pre_bwd_kernel<<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(A);
  TF_LAUNCH_CHECK("tf_shape_glue_pre_bwd");
  return TF_OK;
}

extern "C" int tf_shape_glue_post_fwd(const float* mat, const float* nov, const float* diffuse_light, const float* direct_light,
                                      const float* indirect_light, const float* occ_raw, const float* fg_lut, int32_t fg_h, int32_t fg_w, int64_t n,
                                      float* color, float* occ_prob, tf_stream_t stream) {
  TF_REQUIRE(n >= 0 && fg_h > 0 && fg_w > 0, TF_ESHAPE, "tf_shape_glue_post_fwd: n < 0 or an empty LUT");
  if (n == 0) return TF_OK;
  TF_REQUIRE(mat && nov && diffuse_light && direct_light && indirect_light && occ_raw && fg_lut && color && occ_prob, TF_EINVAL,
             "tf_shape_glue_post_fwd: null pointer");
  GluePost A{mat, nov, diffuse_light, direct_light, indirect_light, occ_raw, fg_lut, fg_h, fg_w, n, color, occ_prob,
             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  shape_glue_post_kernel<false><<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(A);
  TF_LAUNCH_CHECK("tf_shape_glue_post_fwd");
  return TF_OK;
}

extern "C" int tf_shape_glue_post_bwd(const float* mat, const float* nov, const float* diffuse_light, const float* direct_light,
                                      const float* indirect_light, const float* occ_raw, const float* fg_lut, int32_t fg_h, int32_t fg_w,
                                      const float* g_color, const float* g_occ_prob, int64_t n, float* g_mat, float* g_nov, float* g_diffuse_light,
                                      float* g_direct_light, float* g_indirect_light, float* g_occ_raw, tf_stream_t stream) {
  TF_REQUIRE(n >= 0 && fg_h > 0 && fg_w > 0, TF_ESHAPE, "tf_shape_glue_post_bwd: n < 0 or an empty LUT");
  if (n == 0) return TF_OK;
  TF_REQUIRE(mat && nov && diffuse_light && direct_light && indirect_light && occ_raw && fg_lut && g_color && g_mat && g_nov && g_diffuse_light &&
             g_direct_light && g_indirect_light && g_occ_raw, TF_EINVAL, "tf_shape_glue_post_bwd: null pointer");
  GluePost A{mat, nov, diffuse_light, direct_light, indirect_light, occ_raw, fg_lut, fg_h, fg_w, n, nullptr, nullptr,
             g_color, g_occ_prob, g_mat, g_nov, g_diffuse_light, g_direct_light, g_indirect_light, g_occ_raw};
  shape_glue_post_kernel<true><<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(A);
  TF_LAUNCH_CHECK("tf_shape_glue_post_bwd");
  return TF_OK;
}

// ---- F.normalize(x, dim=-1) on [n,3] rows, optionally with the eikonal residual (|x| - 1)^2 (shapeRenderer.py:1137, :1145) and optionally
// on the blend x = v a + (1 - a) c with a constant c (the composited ray normal, :1207-1208): one launch each way instead of the 3 + 15
// (normalize), 3 + 8 (residual) and 7 + 15 (blend + normalize) element-wise launches of torch and its autograd.
struct Norm3 {
  const float* x; const float* acc; float c[3];      // acc non-NULL: rows are x acc + (1 - acc) c
  long long n;
  float* y; float* err;                              // forward (err may be NULL)
  const float* g_y; const float* g_err;              // backward inputs (either may be NULL)
  float* g_x; float* g_acc;                          // backward outputs (g_acc with acc only)
};

template <bool BWD>
__global__ void __launch_bounds__(256) normalize3_kernel(Norm3 A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n) return;
  float x0 = A.x[3 * i], x1 = A.x[3 * i + 1], x2 = A.x[3 * i + 2];
  const float v0 = x0, v1 = x1, v2 = x2;
  float a = 1.f;
  if (A.acc) { a = A.acc[i]; x0 = x0 * a + (1.f - a) * A.c[0]; x1 = x1 * a + (1.f - a) * A.c[1]; x2 = x2 * a + (1.f - a) * A.c[2]; }
  const float len = sqrtf(x0 * x0 + x1 * x1 + x2 * x2), den = fmaxf(len, 1e-12f);
  const float y0 = x0 / den, y1 = x1 / den, y2 = x2 / den;
  if (!BWD) {
    A.y[3 * i] = y0; A.y[3 * i + 1] = y1; A.y[3 * i + 2] = y2;
    if (A.err) A.err[i] = (len - 1.f) * (len - 1.f);
    return;
  }
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (A.g_y) {
    const float a0 = A.g_y[3 * i], a1 = A.g_y[3 * i + 1], a2 = A.g_y[3 * i + 2];
    if (len < 1e-12f) { g0 = a0 / den; g1 = a1 / den; g2 = a2 / den; }      // x / eps
    else { const float d = a0 * y0 + a1 * y1 + a2 * y2; g0 = (a0 - y0 * d) / den; g1 = (a1 - y1 * d) / den; g2 = (a2 - y2 * d) / den; }
  }
  if (A.g_err && len > 0.f) {      // d (|x| - 1)^2 = 2 (|x| - 1) x / |x|   (torch's norm backward: 0 at x = 0)
    const float k = A.g_err[i] * 2.f * (len - 1.f) / len;
    g0 += k * x0; g1 += k * x1; g2 += k * x2;
  }
  if (A.acc) {
    A.g_acc[i] = g0 * (v0 - A.c[0]) + g1 * (v1 - A.c[1]) + g2 * (v2 - A.c[2]);
    g0 *= a; g1 *= a; g2 *= a;
  }
  A.g_x[3 * i] = g0; A.g_x[3 * i + 1] = g1; A.g_x[3 * i + 2] = g2;
}

extern "C" int tf_normalize3_fwd(const float* x, const float* acc, const float* blend_c, int64_t n, float* y, float* err, tf_stream_t stream) {
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_normalize3_fwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(x && y && (!acc || blend_c), TF_EINVAL, "tf_normalize3_fwd: null pointer (acc needs blend_c)");
  Norm3 A{x, acc, {0.f, 0.f, 0.f}, n, y, err, nullptr, nullptr, nullptr, nullptr};
  if (acc) for (int k = 0; k < 3; ++k) A.c[k] = blend_c[k];
  normalize3_kernel<false><<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(A);
  TF_LAUNCH_CHECK("tf_normalize3_fwd");
  return TF_OK;
}

extern "C" int tf_normalize3_bwd(const float* x, const float* acc, const float* blend_c, const float* g_y, const float* g_err, int64_t n,
                                 float* g_x, float* g_acc, tf_stream_t stream) {
  TF_REQUIRE(n >= 0, TF_ESHAPE, "tf_normalize3_bwd: n < 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(x && g_x && (!acc || (blend_c && g_acc)), TF_EINVAL, "tf_normalize3_bwd: null pointer (acc needs blend_c and g_acc)");
  Norm3 A{x, acc, {0.f, 0.f, 0.f}, n, nullptr, nullptr, g_y, g_err, g_x, g_acc};
  if (acc) for (int k = 0; k < 3; ++k) A.c[k] = blend_c[k];
  normalize3_kernel<true><<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(A);
  TF_LAUNCH_CHECK("tf_normalize3_bwd");
  return TF_OK;
}
