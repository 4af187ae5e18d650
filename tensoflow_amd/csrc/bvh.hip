// First-hit ray/mesh intersection (visibility rays of the rendering integral).
// Replaces the third-party `raytracing` BVH extension as used by MaterialRenderer.trace
// (network/materialRenderer.py:149,221-223,253-263; wrapper raytracing/raytracer.py:19-54):
//   depth = nearest t (10.0 = miss), pos = o + t d, normal = normalize(-face_normal), hit = depth < 10.
// Host: binned-SAH binary BVH (leaves <= 4 triangles, children stored as adjacent pairs, 32-byte
// nodes).  Device: one lane per ray, near-child-first stack traversal; the per-triangle test is the
// Moeller-Trumbore/iq form with the acceptance window u>=0, v>=0, u+v<=1, t>=0.
#include <algorithm>
#include <cmath>
#include <vector>

#include "tf_common.h"

#define BVH_MAX_DIST 10.0f
#define BVH_LEAF 4
#ifndef BVH_REFILL
#define BVH_REFILL 48   // idle lanes per wave that trigger a refill from the ray pool
#endif
#define BVH_STACK 64

// ----------------------------------------------------------------------------- host build
namespace {
struct Box {
  float lo[3], hi[3];
  void reset() { for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
  void grow(const float* p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
  void grow(const Box& b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
  float area() const {
    float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
    if (d0 < 0) return 0.f;
    return 2.f * (d0 * d1 + d1 * d2 + d2 * d0);
  }
};

struct Builder {
  const float* v;
  const int32_t* f;
  std::vector<int32_t> order;
  std::vector<Box> tbox;
  std::vector<float> cen;  // [nf,3]
  TfBvhNode* nodes;
  int64_t n_nodes = 0;

  void build(int64_t node, int64_t begin, int64_t end) {
    Box b; b.reset();
    Box cb; cb.reset();
    for (int64_t i = begin; i < end; ++i) { b.grow(tbox[order[i]]); cb.grow(&cen[3 * order[i]]); }
    TfBvhNode& nd = nodes[node];
    for (int k = 0; k < 3; ++k) { nd.lo[k] = b.lo[k]; nd.hi[k] = b.hi[k]; }
    const int64_t n = end - begin;
    if (n <= BVH_LEAF) { nd.left = (int32_t)begin; nd.count = (int32_t)n; return; }
    // binned SAH over the widest centroid axis candidates
    int best_axis = -1, best_bin = -1;
    float best_cost = INFINITY;
    const int NB = 16;
    for (int ax = 0; ax < 3; ++ax) {
      float lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
      if (!(ext > 0.f)) continue;
      Box bb[NB]; int cnt[NB];
      for (int i = 0; i < NB; ++i) { bb[i].reset(); cnt[i] = 0; }
      for (int64_t i = begin; i < end; ++i) {
        int t = order[i];
        int bi = std::min(NB - 1, (int)((cen[3 * t + ax] - lo) / ext * NB));
        bb[bi].grow(tbox[t]); cnt[bi]++;
      }
      float la[NB], ra[NB]; int lc[NB], rc[NB];
      Box acc; acc.reset(); int c = 0;
      for (int i = 0; i < NB; ++i) { acc.grow(bb[i]); c += cnt[i]; la[i] = acc.area(); lc[i] = c; }
      acc.reset(); c = 0;
      for (int i = NB - 1; i >= 0; --i) { acc.grow(bb[i]); c += cnt[i]; ra[i] = acc.area(); rc[i] = c; }
      for (int i = 0; i < NB - 1; ++i) {
        if (lc[i] == 0 || rc[i + 1] == 0) continue;
        float cost = la[i] * lc[i] + ra[i + 1] * rc[i + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = i; }
      }
    }
    int64_t mid;
    if (best_axis < 0) {
      mid = begin + n / 2;  // all centroids coincide
    } else {
      float lo = cb.lo[best_axis], ext = cb.hi[best_axis] - cb.lo[best_axis];
      auto it = std::partition(order.begin() + begin, order.begin() + end, [&](int32_t t) {
        int bi = std::min(NB - 1, (int)((cen[3 * t + best_axis] - lo) / ext * NB));
        return bi <= best_bin;
      });
      mid = it - order.begin();
      if (mid == begin || mid == end) mid = begin + n / 2;
    }
    const int64_t left = n_nodes;
    n_nodes += 2;
    nd.left = (int32_t)left;
    nd.count = 0;
    build(left, begin, mid);
    build(left + 1, mid, end);
  }
};
}  // namespace

extern "C" int64_t tf_bvh_build_host(const float* verts_host, int64_t nv, const int32_t* faces_host, int64_t nf,
                                     TfBvhNode* nodes_host, float* tris_host) {
  TF_REQUIRE(verts_host && faces_host && nodes_host && tris_host, TF_EINVAL, "tf_bvh_build_host: null pointer");
  TF_REQUIRE(nv > 0 && nf > 0 && nf < (1LL << 30), TF_ESHAPE, "tf_bvh_build_host: need 0 < nf < 2^30 and nv > 0");
  Builder B;
  B.v = verts_host; B.f = faces_host; B.nodes = nodes_host;
  B.order.resize(nf); B.tbox.resize(nf); B.cen.resize(3 * nf);
  for (int64_t t = 0; t < nf; ++t) {
    B.order[t] = (int32_t)t;
    B.tbox[t].reset();
    float c[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
      int32_t vi = faces_host[3 * t + k];
      TF_REQUIRE(vi >= 0 && vi < nv, TF_ESHAPE, "tf_bvh_build_host: face %lld references vertex %d (nv=%lld)", (long long)t, vi, (long long)nv);
      B.tbox[t].grow(verts_host + 3 * vi);
      for (int a = 0; a < 3; ++a) c[a] += verts_host[3 * vi + a];
    }
    for (int a = 0; a < 3; ++a) B.cen[3 * t + a] = c[a] / 3.f;
  }
  B.n_nodes = 1;
  B.build(0, 0, nf);
  for (int64_t i = 0; i < nf; ++i) {
    int32_t t = B.order[i];
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a) tris_host[9 * i + 3 * k + a] = verts_host[3 * faces_host[3 * t + k] + a];
  }
  return B.n_nodes;
}

// ----------------------------------------------------------------------------- device trace
#ifdef BVH_STATS
__device__ unsigned long long g_bvh_stats[4];
extern "C" void tf_bvh_stats(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bvh_stats), 32); unsigned long long z[4] = {0,0,0,0}; hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_stats), z, 32); }
#endif
__device__ __forceinline__ bool box_hit(const TfBvhNode& nd, float ox, float oy, float oz, float ix, float iy, float iz,
                                        float tmax, float& tnear) {
  float t0 = (nd.lo[0] - ox) * ix, t1 = (nd.hi[0] - ox) * ix;
  float tmin = fminf(t0, t1), tmx = fmaxf(t0, t1);
  t0 = (nd.lo[1] - oy) * iy; t1 = (nd.hi[1] - oy) * iy;
  tmin = fmaxf(tmin, fminf(t0, t1)); tmx = fminf(tmx, fmaxf(t0, t1));
  t0 = (nd.lo[2] - oz) * iz; t1 = (nd.hi[2] - oz) * iz;
  tmin = fmaxf(tmin, fminf(t0, t1)); tmx = fminf(tmx, fmaxf(t0, t1));
  tnear = tmin;
  // conservative: widen by a few ulps so that a hit the exact triangle test accepts is never culled
  return tmx * 1.0000004f + 1e-6f >= fmaxf(tmin, 0.f) - 1e-6f && tmin <= tmax;
}

__global__ void __launch_bounds__(256) bvh_trace_kernel(const TfBvhNode* __restrict__ nodes, const float* __restrict__ tris,
                                                        const float* __restrict__ o, const float* __restrict__ d,
                                                        float off0, float off1, const unsigned char* __restrict__ live,
                                                        long long m, float* __restrict__ pos,
                                                        float* __restrict__ nrm, float* __restrict__ depth,
                                                        unsigned char* __restrict__ hit) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const float dx = d[3 * i], dy = d[3 * i + 1], dz = d[3 * i + 2];
  // origin = (o + d*off0) + off1*d, with the reference's two separate roundings (no fma contraction)
  float ox = __fadd_rn(__fadd_rn(o[3 * i], __fmul_rn(dx, off0)), __fmul_rn(off1, dx));
  float oy = __fadd_rn(__fadd_rn(o[3 * i + 1], __fmul_rn(dy, off0)), __fmul_rn(off1, dy));
  float oz = __fadd_rn(__fadd_rn(o[3 * i + 2], __fmul_rn(dz, off0)), __fmul_rn(off1, dz));
  const float ix = 1.f / dx, iy = 1.f / dy, iz = 1.f / dz;
  float best = BVH_MAX_DIST;
  int best_tri = -1;
  int stack[BVH_STACK];
  int sp = 0;
  int cur = 0;
  float tn;
  if (!box_hit(nodes[0], ox, oy, oz, ix, iy, iz, best, tn)) cur = -1;
  if (live && !live[i]) cur = -1;   // ray carries zero weight in the integral: reported as a miss, never traversed
#ifdef BVH_STATS
  unsigned n_inner = 0, n_leaf = 0, n_tri = 0, n_iter = 0;
#endif
  while (cur >= 0) {
    const TfBvhNode nd = nodes[cur];
#ifdef BVH_STATS
    n_iter++;
    if (nd.count > 0) { n_leaf++; n_tri += nd.count; } else n_inner++;
#endif
    if (nd.count > 0) {
      for (int k = 0; k < nd.count; ++k) {
        const float* T = tris + 9LL * (nd.left + k);
        const float ax = T[0], ay = T[1], az = T[2];
        const float e1x = T[3] - ax, e1y = T[4] - ay, e1z = T[5] - az;
        const float e2x = T[6] - ax, e2y = T[7] - ay, e2z = T[8] - az;
        const float rx = ox - ax, ry = oy - ay, rz = oz - az;
        const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
        const float qx = ry * dz - rz * dy, qy = rz * dx - rx * dz, qz = rx * dy - ry * dx;
        const float det = 1.f / (dx * nx + dy * ny + dz * nz);
        const float u = det * -(qx * e2x + qy * e2y + qz * e2z);
        const float v = det * (qx * e1x + qy * e1y + qz * e1z);
        const float t = det * -(nx * rx + ny * ry + nz * rz);
        if (u >= 0.f && u <= 1.f && v >= 0.f && u + v <= 1.f && t >= 0.f && t < best) {
          best = t;
          best_tri = nd.left + k;
        }
      }
      cur = sp > 0 ? stack[--sp] : -1;
    } else {
      float tl, tr;
      const bool hl = box_hit(nodes[nd.left], ox, oy, oz, ix, iy, iz, best, tl);
      const bool hr = box_hit(nodes[nd.left + 1], ox, oy, oz, ix, iy, iz, best, tr);
      if (hl && hr) {
        const bool left_first = tl <= tr;
        if (sp < BVH_STACK) stack[sp++] = left_first ? nd.left + 1 : nd.left;
        cur = left_first ? nd.left : nd.left + 1;
      } else if (hl) {
        cur = nd.left;
      } else if (hr) {
        cur = nd.left + 1;
      } else {
        cur = sp > 0 ? stack[--sp] : -1;
      }
    }
  }
#ifdef BVH_STATS
  atomicAdd(&g_bvh_stats[0], (unsigned long long)n_inner); atomicAdd(&g_bvh_stats[1], (unsigned long long)n_leaf);
  atomicAdd(&g_bvh_stats[2], (unsigned long long)n_tri);
  { unsigned mx = n_iter; for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o)); if ((threadIdx.x & 63) == 0) atomicAdd(&g_bvh_stats[3], (unsigned long long)mx * 64ULL); }
#endif
  depth[i] = best;
  if (hit) hit[i] = best < BVH_MAX_DIST ? 1 : 0;
  if (pos) { pos[3 * i] = ox + best * dx; pos[3 * i + 1] = oy + best * dy; pos[3 * i + 2] = oz + best * dz; }
  if (nrm) {
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (best_tri >= 0) {
      const float* T = tris + 9LL * best_tri;
      const float e1x = T[3] - T[0], e1y = T[4] - T[1], e1z = T[5] - T[2];
      const float e2x = T[6] - T[0], e2y = T[7] - T[1], e2z = T[8] - T[2];
      float fx = e1y * e2z - e1z * e2y, fy = e1z * e2x - e1x * e2z, fz = e1x * e2y - e1y * e2x;
      float inv = 1.f / fmaxf(sqrtf(fx * fx + fy * fy + fz * fz), 1e-12f);   // face normal (raytracing)
      fx = -fx * inv; fy = -fy * inv; fz = -fz * inv;                         // materialRenderer.py:256
      inv = 1.f / fmaxf(sqrtf(fx * fx + fy * fy + fz * fz), 1e-12f);          // F.normalize (:257)
      nx = fx * inv; ny = fy * inv; nz = fz * inv;
    }
    nrm[3 * i] = nx; nrm[3 * i + 1] = ny; nrm[3 * i + 2] = nz;
  }
}

// ---- persistent variant with dynamic ray fetch.  Measured on the bench scene: a ray visits 22 inner nodes and 0.9
// leaves on average, but the slowest lane of a statically assigned wave needs 70 steps -- 2/3 of the lanes idle.
// Here every lane pulls a new ray from a global counter as soon as >= 16 lanes of its wave are idle.
__global__ void __launch_bounds__(256) bvh_trace_dyn_kernel(const TfBvhNode* __restrict__ nodes, const float* __restrict__ tris,
                                                            const float* __restrict__ o, const float* __restrict__ d,
                                                            float off0, float off1, const unsigned char* __restrict__ live,
                                                            long long m, unsigned long long* __restrict__ counter,
                                                            float* __restrict__ pos, float* __restrict__ nrm,
                                                            float* __restrict__ depth, unsigned char* __restrict__ hit) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ULL << lane) - 1ULL;
  long long rid = -1;
  int cur = -1, sp = 0, best_tri = -1;
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, best = BVH_MAX_DIST;
  int stack[BVH_STACK];
  bool exhausted = false;
  while (true) {
    // ---- retire finished rays
    if (rid >= 0 && cur < 0) {
      depth[rid] = best;
      if (hit) hit[rid] = best < BVH_MAX_DIST ? 1 : 0;
      if (pos) { pos[3 * rid] = ox + best * dx; pos[3 * rid + 1] = oy + best * dy; pos[3 * rid + 2] = oz + best * dz; }
      if (nrm) {
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (best_tri >= 0) {
          const float* T = tris + 9LL * best_tri;
          const float e1x = T[3] - T[0], e1y = T[4] - T[1], e1z = T[5] - T[2];
          const float e2x = T[6] - T[0], e2y = T[7] - T[1], e2z = T[8] - T[2];
          float fx = e1y * e2z - e1z * e2y, fy = e1z * e2x - e1x * e2z, fz = e1x * e2y - e1y * e2x;
          float inv = 1.f / fmaxf(sqrtf(fx * fx + fy * fy + fz * fz), 1e-12f);
          fx = -fx * inv; fy = -fy * inv; fz = -fz * inv;
          inv = 1.f / fmaxf(sqrtf(fx * fx + fy * fy + fz * fz), 1e-12f);
          nx = fx * inv; ny = fy * inv; nz = fz * inv;
        }
        nrm[3 * rid] = nx; nrm[3 * rid + 1] = ny; nrm[3 * rid + 2] = nz;
      }
      rid = -1;
    }
    // ---- fetch new rays for idle lanes
    const unsigned long long active_b = __ballot(rid >= 0);
    if (!exhausted) {
      const bool want = rid < 0;
      const unsigned long long wb = __ballot(want);
      const int nw = __popcll(wb);
      if (nw >= BVH_REFILL || active_b == 0ULL) {
        unsigned long long base = 0;
        const int leader = __ffsll((long long)wb) - 1;
        if (lane == leader) base = atomicAdd(counter, (unsigned long long)nw);
        base = __shfl(base, leader);
        if (base + (unsigned long long)nw >= (unsigned long long)m) exhausted = true;
        if (want) {
          const long long id = (long long)base + __popcll(wb & lt_mask);
          if (id < m) {
            rid = id;
            dx = d[3 * id]; dy = d[3 * id + 1]; dz = d[3 * id + 2];
            ox = __fadd_rn(__fadd_rn(o[3 * id], __fmul_rn(dx, off0)), __fmul_rn(off1, dx));
            oy = __fadd_rn(__fadd_rn(o[3 * id + 1], __fmul_rn(dy, off0)), __fmul_rn(off1, dy));
            oz = __fadd_rn(__fadd_rn(o[3 * id + 2], __fmul_rn(dz, off0)), __fmul_rn(off1, dz));
            ix = 1.f / dx; iy = 1.f / dy; iz = 1.f / dz;
            best = BVH_MAX_DIST; best_tri = -1; sp = 0;
            float tn;
            cur = box_hit(nodes[0], ox, oy, oz, ix, iy, iz, best, tn) ? 0 : -1;
            if (live && !live[id]) cur = -1;
          }
        }
      }
    }
    if (__ballot(rid >= 0) == 0ULL) {
      if (exhausted) break;
      continue;
    }
    // ---- traverse until enough lanes have finished to make a refill worthwhile
    while (true) {
      if (cur >= 0) {
        const TfBvhNode nd = nodes[cur];
        if (nd.count > 0) {
          for (int k = 0; k < nd.count; ++k) {
            const float* T = tris + 9LL * (nd.left + k);
            const float ax = T[0], ay = T[1], az = T[2];
            const float e1x = T[3] - ax, e1y = T[4] - ay, e1z = T[5] - az;
            const float e2x = T[6] - ax, e2y = T[7] - ay, e2z = T[8] - az;
            const float rx = ox - ax, ry = oy - ay, rz = oz - az;
            const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
            const float qx = ry * dz - rz * dy, qy = rz * dx - rx * dz, qz = rx * dy - ry * dx;
            const float det = 1.f / (dx * nx + dy * ny + dz * nz);
            const float u = det * -(qx * e2x + qy * e2y + qz * e2z);
            const float v = det * (qx * e1x + qy * e1y + qz * e1z);
            const float t = det * -(nx * rx + ny * ry + nz * rz);
            if (u >= 0.f && u <= 1.f && v >= 0.f && u + v <= 1.f && t >= 0.f && t < best) { best = t; best_tri = nd.left + k; }
          }
          cur = sp > 0 ? stack[--sp] : -1;
        } else {
          float tl, tr;
          const bool hl = box_hit(nodes[nd.left], ox, oy, oz, ix, iy, iz, best, tl);
          const bool hr = box_hit(nodes[nd.left + 1], ox, oy, oz, ix, iy, iz, best, tr);
          if (hl && hr) {
            const bool left_first = tl <= tr;
            if (sp < BVH_STACK) stack[sp++] = left_first ? nd.left + 1 : nd.left;
            cur = left_first ? nd.left : nd.left + 1;
          } else if (hl) {
            cur = nd.left;
          } else if (hr) {
            cur = nd.left + 1;
          } else {
            cur = sp > 0 ? stack[--sp] : -1;
          }
        }
      }
      const int n_act = __popcll(__ballot(cur >= 0));
      if (n_act == 0 || (!exhausted && n_act <= 64 - BVH_REFILL)) break;
    }
  }
}

extern "C" int tf_bvh_trace(const TfBvhNode* nodes, const float* tris, int64_t n_nodes, const float* o, const float* d,
                            float origin_offset0, float origin_offset1, const uint8_t* live, int64_t m, float* pos,
                            float* nrm, float* depth, uint8_t* hit, int64_t* work_counter, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(m >= 0 && n_nodes > 0, TF_ESHAPE, "tf_bvh_trace: m < 0 or empty BVH");
  if (m == 0) return TF_OK;
  TF_REQUIRE(nodes && tris && o && d && depth, TF_EINVAL, "tf_bvh_trace: null pointer");
  if (work_counter) {
    hipError_t e = hipMemsetAsync(work_counter, 0, sizeof(int64_t), stream);
    TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_bvh_trace: hipMemsetAsync failed: %s", hipGetErrorString(e));
    long long blocks = (m + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;   // 8 resident 256-thread blocks per CU pull rays until the pool is empty
    bvh_trace_dyn_kernel<<<(unsigned)blocks, 256, 0, stream>>>(nodes, tris, o, d, origin_offset0, origin_offset1, live, m,
                                                              (unsigned long long*)work_counter, pos, nrm, depth, hit);
  } else {
    bvh_trace_kernel<<<tf_blocks(m, 256), 256, 0, stream>>>(nodes, tris, o, d, origin_offset0, origin_offset1, live, m,
                                                          pos, nrm, depth, hit);
  }
  TF_LAUNCH_CHECK("tf_bvh_trace");
  return TF_OK;
}
