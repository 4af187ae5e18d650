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
