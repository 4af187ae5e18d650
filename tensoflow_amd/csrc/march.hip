// Sample generation for the shape-stage march.
//   tf_alpha_mask_sample : AlphaGridMask.sample_alpha(...) > 0 (network/shapeRenderer.py:78-97, used at :1120-1121) on a
//                          binary occupancy volume -- trilinear F.grid_sample(align_corners=True, zeros padding) > 0,
//                          evaluated as "some in-range corner voxel is set and its three 1-D weights are non-zero"
//                          (all terms of the trilinear sum are >= 0), bit-exact with the reference mask.
//   tf_march_uniform     : fixed-step sampler + occupancy culling + packing -- the role nerfacc's
//                          OccGridEstimator.sampling plays at shapeRenderer.py:950-959 and BASELINE configs[1]'s
//                          "256 uniform steps in the aabb slab".  One wave per ray, 64 steps per pass; live samples
//                          are compacted per wavefront with ballot + prefix popcount so the packed list stays
//                          ordered by (ray, t) as render_weight_from_alpha requires.  Two passes (count, write) around a
//                          device prefix sum keep the output dense and deterministic.
// The binary volume is 2 MB at 128^3 (u8) and stays L2-resident; rays are read once.
#include "tf_common.h"

#pragma clang fp contract(off)

struct MaskGeom {
  const unsigned char* vol;   // [D,H,W], W <- x
  int D, H, W;
  float lo[3], inv[3];        // normalize_coord: (x - lo) * inv - 1, inv = 1/size*2
  int cells;                  // 1: `vol` is an occupancy grid [rx = D, ry = H, rz = W] looked up by the CELL that holds the point
};

// nerfacc-style occupancy grid (OccGridEstimator.binaries, flattened (x * ry + y) * rz + z): the cell containing the point
__device__ __forceinline__ bool cell_alive(const MaskGeom& M, float px, float py, float pz) {
  const float ux = (px - M.lo[0]) * M.inv[0] * 0.5f, uy = (py - M.lo[1]) * M.inv[1] * 0.5f, uz = (pz - M.lo[2]) * M.inv[2] * 0.5f;
  const int ix = min(max((int)floorf(ux * (float)M.D), 0), M.D - 1);
  const int iy = min(max((int)floorf(uy * (float)M.H), 0), M.H - 1);
  const int iz = min(max((int)floorf(uz * (float)M.W), 0), M.W - 1);
  return M.vol[((long long)ix * M.H + iy) * M.W + iz] != 0;
}

__device__ __forceinline__ bool axis_corners(float g, int n, int& i0, float& w0, float& w1) {
  // grid_sample align_corners=True unnormalisation: ((g + 1) / 2) * (n - 1)
  const float x = ((g + 1.f) / 2.f) * (float)(n - 1);
  const float f = floorf(x);
  i0 = (int)f;
  w1 = x - f;            // weight of corner i0+1  (torch: ix - ix_tnw)
  w0 = (f + 1.f) - x;    // weight of corner i0    (torch: ix_tse - ix)
  return true;
}

__device__ __forceinline__ bool mask_alive(const MaskGeom& M, float px, float py, float pz) {
  int ix, iy, iz;
  float wx0, wx1, wy0, wy1, wz0, wz1;
  axis_corners((px - M.lo[0]) * M.inv[0] - 1.f, M.W, ix, wx0, wx1);
  axis_corners((py - M.lo[1]) * M.inv[1] - 1.f, M.H, iy, wy0, wy1);
  axis_corners((pz - M.lo[2]) * M.inv[2] - 1.f, M.D, iz, wz0, wz1);
  bool any = false;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int x = ix + (c & 1), y = iy + ((c >> 1) & 1), z = iz + (c >> 2);
    const float w = ((c & 1) ? wx1 : wx0) * ((c & 2) ? wy1 : wy0) * ((c & 4) ? wz1 : wz0);
    const bool in = x >= 0 && x < M.W && y >= 0 && y < M.H && z >= 0 && z < M.D;
    if (in && w > 0.f && M.vol[((long long)z * M.H + y) * M.W + x]) any = true;
  }
  return any;
}

__global__ void __launch_bounds__(256) alpha_mask_kernel(MaskGeom M, const float* __restrict__ pts, long long n,
                                                         unsigned char* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = mask_alive(M, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]) ? 1 : 0;
}

struct MarchBox { float lo[3], hi[3]; };

template <int WRITE>
__global__ void __launch_bounds__(256) march_uniform_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                            const float* __restrict__ near, const float* __restrict__ far,
                                                            long long rn, int n_steps, float step_size, MarchBox B, MaskGeom M,
                                                            const float* __restrict__ t_jitter,
                                                            const long long* __restrict__ offsets, long long* __restrict__ counts,
                                                            float* __restrict__ t0_out, float* __restrict__ t1_out,
                                                            long long* __restrict__ ridx_out) {
  const int lane = threadIdx.x & 63;
  const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= rn) return;
  const float ox = o[3 * ray], oy = o[3 * ray + 1], oz = o[3 * ray + 2];
  const float dx = d[3 * ray], dy = d[3 * ray + 1], dz = d[3 * ray + 2];
  const float nr = near[ray], fr = far[ray];
  // slab test as ShapeRenderer.sample_ray (shapeRenderer.py:878-884): zero components replaced by 1e-6
  const float vx = dx == 0.f ? 1e-6f : dx, vy = dy == 0.f ? 1e-6f : dy, vz = dz == 0.f ? 1e-6f : dz;
  const float ax = (B.hi[0] - ox) / vx, bx = (B.lo[0] - ox) / vx;
  const float ay = (B.hi[1] - oy) / vy, by = (B.lo[1] - oy) / vy;
  const float az = (B.hi[2] - oz) / vz, bz = (B.lo[2] - oz) / vz;
  float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  tmin = fminf(fmaxf(tmin, nr), fr);
  tmax = fminf(fmaxf(tmax, nr), fr);
  if (t_jitter) tmin += t_jitter[ray];            // stratified start (OccGridEstimator.sampling(stratified=True): + U[0,1) * step)
  const float step = step_size > 0.f ? step_size : (tmax - tmin) / (float)n_steps;
  long long base_out = WRITE ? offsets[ray] : 0;
  long long cnt = 0;
  if (step > 0.f && tmax > tmin) {
    for (int b0 = 0; b0 < n_steps; b0 += 64) {
      const int i = b0 + lane;
      const float t0 = tmin + step * (float)i;
      const float t1 = t0 + step;
      bool alive = i < n_steps && t0 < tmax;
      if (alive) {
        const float mid = (t0 + t1) * 0.5f;
        const float px = ox + dx * mid, py = oy + dy * mid, pz = oz + dz * mid;
        alive = !(B.lo[0] > px || px > B.hi[0] || B.lo[1] > py || py > B.hi[1] || B.lo[2] > pz || pz > B.hi[2]);
        if (alive && M.vol) alive = M.cells ? cell_alive(M, px, py, pz) : mask_alive(M, px, py, pz);
      }
      const unsigned long long m = __ballot(alive);
      if (WRITE && alive) {
        const long long dst = base_out + cnt + __popcll(m & ((1ull << lane) - 1ull));
        t0_out[dst] = t0; t1_out[dst] = t1; ridx_out[dst] = ray;
      }
      cnt += __popcll(m);
      if (__ballot(i < n_steps && t0 < tmax) == 0ull) break;      // past tmax for every lane
    }
  }
  if (!WRITE && lane == 0) counts[ray] = cnt;
}

static int mask_geom(const unsigned char* vol, int D, int H, int W, const float* aabb_host, MaskGeom* M) {
  M->vol = vol; M->D = D; M->H = H; M->W = W; M->cells = 0;
  for (int k = 0; k < 3; ++k) {
    M->lo[k] = aabb_host[k];
    const float size = aabb_host[3 + k] - aabb_host[k];
    M->inv[k] = 1.0f / size * 2.f;                  // invgridSize of AlphaGridMask (shapeRenderer.py:85-86)
  }
  return 0;
}

extern "C" int tf_alpha_mask_sample(const uint8_t* volume, int32_t D, int32_t H, int32_t W, const float* aabb_host,
                                    const float* pts, int64_t n, uint8_t* alive, tf_stream_t stream) {
  TF_REQUIRE(n >= 0 && D >= 2 && H >= 2 && W >= 2, TF_ESHAPE, "tf_alpha_mask_sample: bad sizes n=%lld vol=%dx%dx%d", (long long)n, D, H, W);
  if (n == 0) return TF_OK;
  TF_REQUIRE(volume && aabb_host && pts && alive, TF_EINVAL, "tf_alpha_mask_sample: null pointer");
  MaskGeom M;
  mask_geom(volume, D, H, W, aabb_host, &M);
  alpha_mask_kernel<<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(M, pts, n, alive);
  TF_LAUNCH_CHECK("tf_alpha_mask_sample");
  return TF_OK;
}

extern "C" int tf_march_uniform(const float* rays_o, const float* rays_d, const float* near, const float* far, int64_t rn,
                                int32_t n_steps, float step_size, const float* aabb_host, const uint8_t* volume, int32_t D,
                                int32_t H, int32_t W, const float* mask_aabb_host, int32_t occupancy_mode, const float* t_jitter,
                                const int64_t* offsets, int64_t* counts, float* t_starts, float* t_ends, int64_t* ray_indices,
                                tf_stream_t stream) {
  TF_REQUIRE(occupancy_mode == 0 || occupancy_mode == 1, TF_EINVAL, "tf_march_uniform: occupancy_mode %d", occupancy_mode);
  TF_REQUIRE(rn >= 0 && n_steps > 0, TF_ESHAPE, "tf_march_uniform: rn=%lld n_steps=%d", (long long)rn, n_steps);
  if (rn == 0) return TF_OK;
  TF_REQUIRE(rays_o && rays_d && near && far && aabb_host, TF_EINVAL, "tf_march_uniform: null pointer");
  TF_REQUIRE(!volume || (D >= 2 && H >= 2 && W >= 2 && mask_aabb_host), TF_ESHAPE, "tf_march_uniform: bad occupancy volume");
  MarchBox B;
  for (int k = 0; k < 3; ++k) { B.lo[k] = aabb_host[k]; B.hi[k] = aabb_host[3 + k]; }
  MaskGeom M;
  if (volume) mask_geom(volume, D, H, W, mask_aabb_host, &M); else { M.vol = nullptr; M.D = M.H = M.W = 2; M.cells = 0; for (int k = 0; k < 3; ++k) { M.lo[k] = 0; M.inv[k] = 1; } }
  if (volume) M.cells = occupancy_mode;
  const unsigned blocks = tf_blocks(rn, 4);
  if (!offsets) {
    TF_REQUIRE(counts, TF_EINVAL, "tf_march_uniform: count pass needs `counts`");
    march_uniform_kernel<0><<<blocks, 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, near, far, rn, n_steps, step_size, B, M, t_jitter, nullptr,
                                                                     (long long*)counts, nullptr, nullptr, nullptr);
  } else {
    TF_REQUIRE(t_starts && t_ends && ray_indices, TF_EINVAL, "tf_march_uniform: write pass needs the three outputs");
    march_uniform_kernel<1><<<blocks, 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, near, far, rn, n_steps, step_size, B, M, t_jitter,
                                                                     (const long long*)offsets, nullptr, t_starts, t_ends,
                                                                     (long long*)ray_indices);
  }
  TF_LAUNCH_CHECK("tf_march_uniform");
  return TF_OK;
}


// =====================================================================================================================
// ShapeRenderer.sample_ray (network/shapeRenderer.py:871-932) with its helpers upsample / cat_z_vals (:820-869) and sample_pdf
// (utils/network_utils.py:117-147, det=True) as THREE kernels per ray batch instead of ~100 torch launches per up-sampling round
// (419 launches / 1.5 ms of a 1 024-ray training step): the slab test + the 64 uniform t (init), one up-sampling round (NeuS
// weights of the current samples -> inverse-CDF resampling of n_imp new t, one WAVE per ray: scans by shuffles, the ray's z / sdf /
// cdf rows in LDS), and the stable merge of the new samples into the sorted row.  The field evaluations in between stay
// tf_sdf_forward.  Arithmetic follows the torch expressions term by term (no fma contraction in this file).
__device__ __forceinline__ float ball_radius(float t, float radii, float c) {           // shapeRenderer.py:966-970
  const float inv = 1.0f / c;
  const float tmp = sqrtf(inv * inv - 1.f) - radii;
  return t * radii * c / sqrtf(tmp * tmp + 1.0f);
}

__global__ void __launch_bounds__(256) sample_ray_init_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                              const float* __restrict__ near, const float* __restrict__ far,
                                                              const float* __restrict__ radiis, const float* __restrict__ rays_cos,
                                                              float lo0, float lo1, float lo2, float hi0, float hi1, float hi2,
                                                              const float* __restrict__ lin, const float* __restrict__ t_rand, long long rn,
                                                              int S, float base_radii, float* __restrict__ z, float* __restrict__ pts,
                                                              float* __restrict__ lv) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rn * S) return;
  const long long r = e / S;
  const int k = (int)(e - r * S);
  const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
  float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float dv = d[3 * r + a], ov = o[3 * r + a];
    const float vec = dv == 0.f ? 1e-6f : dv;
    const float ra = (hi[a] - ov) / vec, rb = (lo[a] - ov) / vec;
    tmin = fmaxf(tmin, fminf(ra, rb));
    tmax = fminf(tmax, fmaxf(ra, rb));
  }
  const float nr = near[r], fr = far[r];
  tmin = fminf(fmaxf(tmin, nr), fr);
  tmax = fminf(fmaxf(tmax, nr), fr);
  float t = tmin + (tmax - tmin) * lin[k];
  if (t_rand) t = t + t_rand[r] * 2.0f / (float)S;
  z[e] = t;
#pragma unroll
  for (int a = 0; a < 3; ++a) pts[3 * e + a] = o[3 * r + a] + d[3 * r + a] * t;
  lv[e] = log2f(ball_radius(t, radiis[r], rays_cos[r]) / base_radii);
}

// exclusive scans over 128 values held two per lane (elements 2 l, 2 l + 1)
__device__ __forceinline__ void wave_excl_prod2(float a, float b, int lane, float& ea, float& eb) {
  float p = a * b;                                   // inclusive product over lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float q = __shfl_up(p, o);
    if (lane >= o) p = q * p;
  }
  float ex = __shfl_up(p, 1);
  if (lane == 0) ex = 1.f;
  ea = ex; eb = ex * a;
}
__device__ __forceinline__ void wave_incl_sum2(float a, float b, int lane, float& ia, float& ib, float& total) {
  float p = a + b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float q = __shfl_up(p, o);
    if (lane >= o) p = q + p;
  }
  float ex = __shfl_up(p, 1);
  if (lane == 0) ex = 0.f;
  ia = ex + a; ib = ia + b;
  total = __shfl(p, 63);
}

__global__ void __launch_bounds__(256) sample_ray_upsample_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                                  const float* __restrict__ radiis, const float* __restrict__ rays_cos,
                                                                  const float* __restrict__ z, const float* __restrict__ sdf, long long rn,
                                                                  int S, int n_imp, float inv_s, const float* __restrict__ u_lin,
                                                                  float base_radii, float* __restrict__ new_t, float* __restrict__ npts,
                                                                  float* __restrict__ nlv) {
  __shared__ float sz[4][128], ss[4][128], sc[4][128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wv;
  if (r >= rn) return;                                 // wave-uniform
  float* Z = sz[wv]; float* Sd = ss[wv]; float* Cd = sc[wv];
  for (int i = lane; i < 128; i += 64) { Z[i] = i < S ? z[r * S + i] : 0.f; Sd[i] = i < S ? sdf[r * S + i] : 0.f; }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2], dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
  const int NI = S - 1;                                // intervals
  // per interval k: alpha (upsample, :820-857).  lane l holds intervals 2 l and 2 l + 1
  float al[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = 2 * lane + q;
    float a = 0.f;
    if (k < NI) {
      const float pz_ = Z[k], nz_ = Z[k + 1], ps = Sd[k], ns = Sd[k + 1];
      auto rad = [&](float t) {
        const float px = ox + dx * t, py = oy + dy * t, pzc = oz + dz * t;
        return sqrtf(px * px + py * py + pzc * pzc);
      };
      const bool inside = (rad(pz_) < 1.0f) | (rad(nz_) < 1.0f);
      const float mid = (ps + ns) * 0.5f;
      float cs = (ns - ps) / (nz_ - pz_ + 1e-5f);
      float prev = 0.f;
      if (k > 0) prev = (Sd[k] - Sd[k - 1]) / (Z[k] - Z[k - 1] + 1e-5f);
      cs = fminf(prev, cs);
      cs = fminf(fmaxf(cs, -1e3f), 0.0f) * (inside ? 1.f : 0.f);
      const float dist = nz_ - pz_;
      const float pc = 1.f / (1.f + expf(-((mid - cs * dist * 0.5f) * inv_s)));
      const float nc = 1.f / (1.f + expf(-((mid + cs * dist * 0.5f) * inv_s)));
      a = (pc - nc + 1e-5f) / (pc + 1e-5f);
    }
    al[q] = a;
  }
  // weights = alpha * exclusive cumprod(1 - alpha + 1e-7); padding intervals contribute a factor that nobody reads
  float e0, e1;
  wave_excl_prod2(2 * lane < NI ? 1.f - al[0] + 1e-7f : 1.f, 2 * lane + 1 < NI ? 1.f - al[1] + 1e-7f : 1.f, lane, e0, e1);
  // sample_pdf (det): weights + 1e-5 -> pdf -> cdf = [0, cumsum]
  const float w0 = 2 * lane < NI ? al[0] * e0 + 1e-5f : 0.f, w1 = 2 * lane + 1 < NI ? al[1] * e1 + 1e-5f : 0.f;
  float c0, c1, tot;
  wave_incl_sum2(w0, w1, lane, c0, c1, tot);
  // torch: pdf = w / sum, cdf = cumsum(pdf): the cumulative sum of the QUOTIENTS (not the quotient of the cumulative sum)
  float q0, q1, qt;
  wave_incl_sum2(w0 / tot, w1 / tot, lane, q0, q1, qt);
  if (lane == 0) Cd[0] = 0.f;
  if (2 * lane < NI) Cd[2 * lane + 1] = q0;
  if (2 * lane + 1 < NI) Cd[2 * lane + 2] = q1;
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < n_imp) {
    const float u = u_lin[lane];
    int lo_ = 0, hi_ = S;                              // searchsorted(cdf, u, right=True) = #{cdf <= u}: cdf is non-decreasing
    while (lo_ < hi_) {
      const int m = (lo_ + hi_) >> 1;
      if (Cd[m] <= u) lo_ = m + 1; else hi_ = m;
    }
    const int inds = lo_;
    const int below = max(inds - 1, 0), above = min(inds, S - 1);
    const float cb = Cd[below], ca = Cd[above], b0 = Z[below], b1 = Z[above];
    float den = ca - cb;
    den = den < 1e-5f ? 1.0f : den;
    const float t = b0 + (u - cb) / den * (b1 - b0);
    const long long e = r * n_imp + lane;
    new_t[e] = t;
    if (npts) {
      npts[3 * e] = ox + dx * t; npts[3 * e + 1] = oy + dy * t; npts[3 * e + 2] = oz + dz * t;
      nlv[e] = log2f(ball_radius(t, radiis[r], rays_cos[r]) / base_radii);
    }
  }
}

// torch.sort(cat([z, new_t])) (stable: on ties the old samples come first) + the gather of the sdf row, by rank counting
__global__ void __launch_bounds__(256) sample_ray_merge_kernel(const float* __restrict__ z, const float* __restrict__ sdf,
                                                               const float* __restrict__ new_t, const float* __restrict__ nsdf, long long rn,
                                                               int S, int n_imp, float* __restrict__ z_out, float* __restrict__ sdf_out) {
  __shared__ float sz[4][128], sn[4][32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wv;
  if (r >= rn) return;
  float* Z = sz[wv]; float* Nn = sn[wv];
  for (int i = lane; i < S; i += 64) Z[i] = z[r * S + i];
  if (lane < n_imp) Nn[lane] = new_t[r * n_imp + lane];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int So = S + n_imp;
  for (int i = lane; i < S; i += 64) {
    const float v = Z[i];
    int c = 0;
    for (int j = 0; j < n_imp; ++j) c += Nn[j] < v ? 1 : 0;
    z_out[r * So + i + c] = v;
    if (sdf_out) sdf_out[r * So + i + c] = sdf[r * S + i];
  }
  if (lane < n_imp) {
    // rank by FULL counting, as a stable sort orders them: the inverse-CDF value b0 + frac * fl(b1 - b0) can exceed b1 by an ulp while
    // the next sample equals b1, so the new samples are not assumed to arrive sorted among themselves (every output slot is written
    // exactly once whatever the order)
    const float v = Nn[lane];
    int c = 0;
    for (int i = 0; i < S; ++i) c += Z[i] <= v ? 1 : 0;
    for (int j = 0; j < n_imp; ++j) c += (Nn[j] < v || (Nn[j] == v && j < lane)) ? 1 : 0;
    z_out[r * So + c] = v;
    if (sdf_out) sdf_out[r * So + c] = nsdf[r * n_imp + lane];
  }
}

extern "C" int tf_sample_ray_init(const float* rays_o, const float* rays_d, const float* near, const float* far, const float* radiis,
                                  const float* rays_cos, const float* aabb_host, const float* lin, const float* t_rand, int64_t rn,
                                  int32_t n_samples, float base_radii, float* z, float* pts, float* level, tf_stream_t stream) {
  TF_REQUIRE(rn >= 0 && n_samples >= 2 && n_samples <= 128, TF_ESHAPE, "tf_sample_ray_init: rn < 0 or n_samples outside [2, 128]");
  if (rn == 0) return TF_OK;
  TF_REQUIRE(rays_o && rays_d && near && far && radiis && rays_cos && aabb_host && lin && z && pts && level, TF_EINVAL, "tf_sample_ray_init: null pointer");
  sample_ray_init_kernel<<<tf_blocks(rn * n_samples, 256), 256, 0, (hipStream_t)stream>>>(
      rays_o, rays_d, near, far, radiis, rays_cos, aabb_host[0], aabb_host[1], aabb_host[2], aabb_host[3], aabb_host[4], aabb_host[5], lin,
      t_rand, rn, n_samples, base_radii, z, pts, level);
  TF_LAUNCH_CHECK("tf_sample_ray_init");
  return TF_OK;
}

extern "C" int tf_sample_ray_upsample(const float* rays_o, const float* rays_d, const float* radiis, const float* rays_cos, const float* z,
                                      const float* sdf, int64_t rn, int32_t n_cur, int32_t n_imp, float inv_s, const float* u_lin,
                                      float base_radii, float* new_t, float* new_pts, float* new_level, tf_stream_t stream) {
  TF_REQUIRE(rn >= 0 && n_cur >= 2 && n_cur <= 128 && n_imp >= 1 && n_imp <= 32, TF_ESHAPE,
             "tf_sample_ray_upsample: n_cur outside [2, 128] or n_imp outside [1, 32]");
  if (rn == 0) return TF_OK;
  TF_REQUIRE(rays_o && rays_d && radiis && rays_cos && z && sdf && u_lin && new_t && (!new_pts || new_level), TF_EINVAL,
             "tf_sample_ray_upsample: null pointer");
  sample_ray_upsample_kernel<<<tf_blocks(rn, 4), 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, radiis, rays_cos, z, sdf, rn, n_cur, n_imp,
                                                                               inv_s, u_lin, base_radii, new_t, new_pts, new_level);
  TF_LAUNCH_CHECK("tf_sample_ray_upsample");
  return TF_OK;
}

extern "C" int tf_sample_ray_merge(const float* z, const float* sdf, const float* new_t, const float* new_sdf, int64_t rn, int32_t n_cur,
                                   int32_t n_imp, float* z_out, float* sdf_out, tf_stream_t stream) {
  TF_REQUIRE(rn >= 0 && n_cur >= 1 && n_cur <= 128 && n_imp >= 1 && n_imp <= 32, TF_ESHAPE,
             "tf_sample_ray_merge: n_cur outside [1, 128] or n_imp outside [1, 32]");
  if (rn == 0) return TF_OK;
  TF_REQUIRE(z && new_t && z_out && (!sdf_out || (sdf && new_sdf)), TF_EINVAL, "tf_sample_ray_merge: null pointer");
  sample_ray_merge_kernel<<<tf_blocks(rn, 4), 256, 0, (hipStream_t)stream>>>(z, sdf, new_t, new_sdf, rn, n_cur, n_imp, z_out, sdf_out);
  TF_LAUNCH_CHECK("tf_sample_ray_merge");
  return TF_OK;
}

// ---- the element-wise algebra between sample_ray and compute_sdf_alpha (round 5: ~30 launches per shape training step)
// sample_ray's tail (shapeRenderer.py:921-932): per sample of the merged grid t [rn,S] the interval [t, t + dist] (dist to the next
// sample, the last one repeats) and whether the interval's midpoint lies inside the aabb -> t0, t1 [rn*S], inner [rn*S] (bytes)
#pragma clang fp contract(off)
__global__ void __launch_bounds__(256) sample_ray_intervals_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ t,
                                                                   long long rn, int S, MarchBox A, float* __restrict__ t0, float* __restrict__ t1,
                                                                   unsigned char* __restrict__ inner) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rn * S) return;
  const long long r = e / S;
  const int j = (int)(e - r * S);
  const float tj = t[e];
  const float dist = j + 1 < S ? t[e + 1] - tj : (S > 1 ? tj - t[e - 1] : 0.f);
  const float mid = tj + dist * 0.5f;
  bool in = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float p = o[3 * r + k] + d[3 * r + k] * mid;
    in = in && !(A.lo[k] > p) && !(p > A.hi[k]);
  }
  t0[e] = tj;
  t1[e] = tj + dist;
  inner[e] = in ? 1 : 0;
}

// render_core's prelude (shapeRenderer.py:1118-1131): per packed sample i of ray ridx[i] with interval [t0, t1]:
// mid = (t0 + t1) / 2, dists = t1 - t0, viewdir = dirs[ridx], points = o[ridx] + viewdir * mid,
// level = log2(ball_radii(mid, radiis[ridx], cos[ridx]) / base_radii)   (compute_ball_radii, :1038-1044)
__global__ void __launch_bounds__(256) sample_points_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ radiis,
                                                            const float* __restrict__ rcos, const long long* __restrict__ ridx,
                                                            const float* __restrict__ t0, const float* __restrict__ t1, long long n,
                                                            float base_radii, float* __restrict__ mid, float* __restrict__ dists,
                                                            float* __restrict__ viewdir, float* __restrict__ points, float* __restrict__ level) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long r = ridx[i];
  const float a = t0[i], b = t1[i];
  const float m = (a + b) * 0.5f;
  mid[i] = m;
  dists[i] = b - a;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float dk = d[3 * r + k];
    viewdir[3 * i + k] = dk;
    points[3 * i + k] = o[3 * r + k] + dk * m;
  }
  const float rad = radiis[r], c = rcos[r];
  const float inv = 1.0f / c;
  const float tmp = sqrtf(inv * inv - 1.f) - rad;
  const float ball = m * rad * c / sqrtf(tmp * tmp + 1.0f);
  level[i] = log2f(ball / base_radii);
}
#pragma clang fp contract(fast)

extern "C" int tf_sample_ray_intervals(const float* rays_o, const float* dirs, const float* t, int64_t rn, int32_t n_samples,
                                       const float* aabb_host, float* t0, float* t1, uint8_t* inner, tf_stream_t stream) {
  TF_REQUIRE(rn >= 0 && n_samples >= 1, TF_ESHAPE, "tf_sample_ray_intervals: bad sizes");
  if (rn == 0) return TF_OK;
  TF_REQUIRE(rays_o && dirs && t && aabb_host && t0 && t1 && inner, TF_EINVAL, "tf_sample_ray_intervals: null pointer");
  MarchBox A;
  for (int k = 0; k < 3; ++k) { A.lo[k] = aabb_host[k]; A.hi[k] = aabb_host[3 + k]; }
  sample_ray_intervals_kernel<<<tf_blocks(rn * (int64_t)n_samples, 256), 256, 0, (hipStream_t)stream>>>(rays_o, dirs, t, rn, n_samples, A, t0, t1, inner);
  TF_LAUNCH_CHECK("tf_sample_ray_intervals");
  return TF_OK;
}

extern "C" int tf_sample_points(const float* rays_o, const float* dirs, const float* radiis, const float* rays_cos, const int64_t* ray_indices,
                                const float* t0, const float* t1, int64_t n, float base_radii, float* mid, float* dists, float* viewdir,
                                float* points, float* level, tf_stream_t stream) {
  TF_REQUIRE(n >= 0 && base_radii > 0.f, TF_ESHAPE, "tf_sample_points: n < 0 or base_radii <= 0");
  if (n == 0) return TF_OK;
  TF_REQUIRE(rays_o && dirs && radiis && rays_cos && ray_indices && t0 && t1 && mid && dists && viewdir && points && level, TF_EINVAL,
             "tf_sample_points: null pointer");
  sample_points_kernel<<<tf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(rays_o, dirs, radiis, rays_cos, (const long long*)ray_indices, t0, t1, n,
                                                                           base_radii, mid, dists, viewdir, points, level);
  TF_LAUNCH_CHECK("tf_sample_points");
  return TF_OK;
}
