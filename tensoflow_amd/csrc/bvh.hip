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
#include "tf_internal.h"

#define BVH_MAX_DIST 10.0f
#ifndef BVH_LEAF
#define BVH_LEAF 2         // triangles per leaf at most.  Every triangle test is three 16-byte fetches of which most miss the L2 (12.7 MB of
                           // records on the bench mesh against 4 MB): one more pair step is cheaper than two more triangles
                           // (201 M rays, 7 waves per SIMD: 1: 17.0, 2: 15.5, 3: 15.9, 4: 16.5, 6: 17.2 ms)
#endif
#ifndef BVH_REFILL
#define BVH_REFILL 32   // idle lanes per wave that trigger a refill from the ray pool (16: 4.92, 24: 4.79, 32: 4.76, 40: 4.77, 48: 4.92 ms per 50 M rays)
#endif
#define BVH_CHUNK_MAX 512  // rays a wave takes from the global pool per atomic (shrinks towards BVH_CHUNK_MIN at the end)
#define BVH_CHUNK_MIN 64
#ifdef BVH_WIDE
#define BVH_STACK 40       // 4-wide nodes: <= 13 levels (binary depth <= BVH_MAX_DEPTH = 25) x up to 3 postponed children per level
#else
#define BVH_STACK 32       // per-ray traversal stack entries; the builder bounds the tree depth to match
#endif
#ifndef BVH_LDS_STACK
#define BVH_LDS_STACK 10   // of which in LDS (the rest is a per-lane scratch array, touched by the rare deep pile-ups only; 10 vs 12: 15.2 vs 15.4 ms)
#endif
#ifndef BVH_WAVES
#define BVH_WAVES 7        // resident 256-thread blocks per CU the kernel is compiled for (register budget): 72 registers hold the step without
                           // spilling into the hot loop; at 8 (64 registers, 16 spilled) the spill traffic costs more than the eighth wave hides
                           // (17.1 vs 16.5 ms).  5 and 6 need -mllvm -disable-promote-alloca-to-vector (the private stack tail is otherwise
                           // promoted to registers and 118 values spill: 51 ms) and are slower with it too (6: 16.5 vs 15.8 at 7, BVH_LEAF 2)
#endif
#ifdef BVH_WIDE
#define BVH_MAX_DEPTH 25   // deepest node level the builder may create (root = 0): 13 four-wide levels, <= 39 postponed children
#else
#define BVH_MAX_DEPTH 31   // deepest node level the builder may create (root = 0)
#endif
#ifndef BVH_TOP
#define BVH_TOP 127        // pair records of the top of the tree kept in LDS per workgroup (4 KB; numbered breadth-first by the packer)
#endif
#ifndef BVH_LEAF_PAIRED
#define BVH_LEAF_PAIRED 0  // 1: the two triangle records of a leaf are requested together.  Measured slower at every occupancy (7 waves:
                           // 19.6 vs 15.8 ms, 25 spilled registers; 6: 17.1 vs 16.5; 5, no spill: 17.9): the second record's registers
                           // cost more than its overlapped round trip returns
#endif
#ifndef BVH_LEAF_TOUCH
#define BVH_LEAF_TOUCH 0   // 1 (round 4, measured, not adopted): a lane arriving at a leaf touches its triangle records at once, iterations before
                           // the batched leaf step reads them.  16.9 vs 14.5 ms per 201 M rays: vector-memory results return IN ORDER, so the
                           // L2-missing touch sits in front of the next pair fetch of every lane of the wave, and its register costs 10 spills
#endif
#ifndef BVH_LEAF_W
#define BVH_LEAF_W 2       // a wave runs a leaf step once (lanes at a leaf) * BVH_LEAF_W >= (lanes at an inner node)
#endif

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
  int max_depth = 0;

  // levels a balanced subdivision of n triangles still needs below this node
  static int balanced_levels(int64_t n) { int l = 0; while (n > BVH_LEAF) { n = (n + 1) / 2; ++l; } return l; }

  void build(int64_t node, int64_t begin, int64_t end, int depth) {
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
#ifndef BVH_SAH_BINS
#define BVH_SAH_BINS 16     // 32 / 64 bins: traversal time within noise (15.05 / 14.76 / 14.95 ms)
#endif
    const int NB = BVH_SAH_BINS;
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
    // depth bound (the device keeps a BVH_STACK-entry traversal stack per ray): fall back to the median split as soon
    // as the SAH split could push the subtree below depth BVH_MAX_DEPTH; median halving keeps depth + balanced_levels
    // non-increasing, so the bound holds by induction.
    if (depth + 1 + balanced_levels(std::max(mid - begin, end - mid)) > BVH_MAX_DEPTH) {
      const int ax = best_axis < 0 ? 0 : best_axis;
      mid = begin + (n + 1) / 2;
      std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end,
                       [&](int32_t a, int32_t b) { return cen[3 * a + ax] < cen[3 * b + ax]; });
    }
    max_depth = std::max(max_depth, depth + 1);
    const int64_t left = n_nodes;
    n_nodes += 2;
    nd.left = (int32_t)left;
    nd.count = 0;
    build(left, begin, mid, depth + 1);
    build(left + 1, mid, end, depth + 1);
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
  B.build(0, 0, nf, 0);
  for (int64_t i = 0; i < nf; ++i) {
    int32_t t = B.order[i];
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a) tris_host[9 * i + 3 * k + a] = verts_host[3 * faces_host[3 * t + k] + a];
  }
  return B.n_nodes;
}

// ----------------------------------------------------------------------------- traversal layout (host)
// The device does not walk TfBvhNode: one traversal step there costs two dependent memory round trips (the node, then
// its two children), and profiling the first pair layout (64-byte records, full-precision boxes) showed the kernel bound by
// the per-CU vector L1, which serves ONE divergent 16-byte access per clock (rocprofv3: TCP_TOTAL_CACHE_ACCESSES = rays x
// steps x 4).  tf_bvh_pack_host therefore lays the tree out as 32-byte PAIRS -- one record per inner node holding BOTH
// child boxes, quantised to 16 bits per coordinate on ONE global grid, and both child references -- so a step is two
// dwordx4 fetches per lane and the whole inner tree (2.6 MB on the bench mesh) fits an XCD's L2.
//   pair (8 dwords): child0 {lo.x|lo.y<<16, lo.z|hi.x<<16, hi.y|hi.z<<16}, child1 {same}, c0, c1
//   grid:  x = frame.origin + q * frame.scale per axis; boxes are rounded OUTWARD after being inflated by `margin`
//          (8e-6 of the scene extent), which also absorbs the rounding of the kernel's slab test t = fma(q, s/d, (o0-o)/d)
//          -- the quantised box test never culls a box the exact one accepts.  Triangles stay full precision, so hits
//          and depths are unchanged.
//   child reference:  >= 0 pair index;  < -1 leaf = ~((first_triangle << 3) | count), count 1..4;  -1 none
// Pairs are numbered in depth-first order (a subtree is contiguous; the top of the tree shares cache lines).
// Triangles are stored as 48-byte (a, e1, e2) records (3 x dwordx4; e1 = b - a, e2 = c - a).
namespace {
struct Packer {
  const TfBvhNode* nodes;
  uint32_t* pairs;
  int64_t n_pairs = 0;
  double org[3], scl[3], margin[3];
  static int32_t leaf_ref(const TfBvhNode& n) { return ~(int32_t)(((uint32_t)n.left << 3) | (uint32_t)n.count); }
  void quantise(const float* lo, const float* hi, uint32_t* q /*[3]*/) const {
    uint32_t ql[3], qh[3];
    for (int k = 0; k < 3; ++k) {
      if (!(hi[k] >= lo[k])) { ql[k] = 65535; qh[k] = 0; continue; }   // empty box: never hit
      double a = std::floor(((double)lo[k] - margin[k] - org[k]) / scl[k]);
      double b = std::ceil(((double)hi[k] + margin[k] - org[k]) / scl[k]);
      a = std::min(std::max(a, 0.0), 65535.0);
      b = std::min(std::max(b, 0.0), 65535.0);
      ql[k] = (uint32_t)a; qh[k] = (uint32_t)b;
    }
    q[0] = ql[0] | (ql[1] << 16);
    q[1] = ql[2] | (qh[0] << 16);
    q[2] = qh[1] | (qh[2] << 16);
  }
#ifdef BVH_WIDE
  // 4-wide record (16 dwords): an inner node's children are replaced by THEIR children where those are inner nodes too, so one
  // step descends two levels of the binary tree: half the dependent fetches per ray.
  //   dwords 3k .. 3k+2: quantised box of child k (k = 0..3, empty box for an unused slot); dwords 12..15: child references
  int32_t emit(int64_t node) {
    const TfBvhNode& nd = nodes[node];
    const int64_t me = n_pairs++;
    int64_t kids[4];
    int nk = 0;
    for (int c = 0; c < 2; ++c) {
      const int64_t ch = nd.left + c;
      if (nodes[ch].count == 0) { kids[nk++] = nodes[ch].left; kids[nk++] = nodes[ch].left + 1; }
      else kids[nk++] = ch;
    }
    int32_t refs[4] = {-1, -1, -1, -1};
    for (int k = 0; k < 4; ++k) {
      if (k < nk) {
        const TfBvhNode& ch = nodes[kids[k]];
        quantise(ch.lo, ch.hi, pairs + 16 * me + 3 * k);
        if (ch.count > 0) refs[k] = leaf_ref(ch);
      } else {
        const float elo[3] = {1.f, 1.f, 1.f}, ehi[3] = {0.f, 0.f, 0.f};
        quantise(elo, ehi, pairs + 16 * me + 3 * k);
      }
    }
    for (int k = 0; k < nk; ++k)
      if (nodes[kids[k]].count == 0) refs[k] = emit(kids[k]);
    for (int k = 0; k < 4; ++k) pairs[16 * me + 12 + k] = (uint32_t)refs[k];
    return (int32_t)me;
  }
#else
  // Numbering: the first BVH_TOP records are the top of the tree in BREADTH-first order (every ray walks them: the kernel serves
  // them from LDS), the rest depth-first (a subtree is contiguous).  A child's index is always larger than its parent's.
  std::vector<int32_t> index_of;
  void dfs_number(int64_t node) {
    if (index_of[node] < 0) index_of[node] = (int32_t)n_pairs++;
    for (int c = 0; c < 2; ++c)
      if (nodes[nodes[node].left + c].count == 0) dfs_number(nodes[node].left + c);
  }
  void fill(int64_t node) {
    const TfBvhNode& nd = nodes[node];
    const int64_t me = index_of[node];
    for (int c = 0; c < 2; ++c) {
      const TfBvhNode& ch = nodes[nd.left + c];
      quantise(ch.lo, ch.hi, pairs + 8 * me + 3 * c);
      pairs[8 * me + 6 + c] = (uint32_t)(ch.count > 0 ? leaf_ref(ch) : index_of[nd.left + c]);
    }
    for (int c = 0; c < 2; ++c)
      if (nodes[nd.left + c].count == 0) fill(nd.left + c);
  }
  int32_t emit(int64_t root, int64_t n_nodes_total) {   // root is an inner node
    index_of.assign((size_t)n_nodes_total, -1);
    std::vector<int64_t> queue{root};
    for (size_t head = 0; head < queue.size() && n_pairs < BVH_TOP; ++head) {
      const int64_t nd = queue[head];
      index_of[nd] = (int32_t)n_pairs++;
      for (int c = 0; c < 2; ++c)
        if (nodes[nodes[nd].left + c].count == 0) queue.push_back(nodes[nd].left + c);
    }
    dfs_number(root);
    fill(root);
    return 0;
  }
#endif
};
}  // namespace

extern "C" int32_t tf_bvh_record_dwords(void) {
#ifdef BVH_WIDE
  return 16;
#else
  return 8;
#endif
}

extern "C" int64_t tf_bvh_pack_host(const TfBvhNode* nodes_host, int64_t n_nodes, const float* tris_host, int64_t nf,
                                    uint32_t* pairs_host, float* tris12_host, float* frame_host) {
  TF_REQUIRE(nodes_host && tris_host && pairs_host && tris12_host && frame_host, TF_EINVAL, "tf_bvh_pack_host: null pointer");
  TF_REQUIRE(n_nodes > 0 && nf > 0 && nf < (1LL << 28), TF_ESHAPE, "tf_bvh_pack_host: need n_nodes > 0 and 0 < nf < 2^28");
  for (int64_t i = 0; i < n_nodes; ++i) {
    const TfBvhNode& n = nodes_host[i];
    if (n.count > 0) TF_REQUIRE(n.count <= 7 && n.left >= 0 && (int64_t)n.left + n.count <= nf, TF_ESHAPE,
                                "tf_bvh_pack_host: leaf %lld out of range", (long long)i);
    else TF_REQUIRE(n.count == 0 && n.left > 0 && (int64_t)n.left + 1 < n_nodes, TF_ESHAPE,
                    "tf_bvh_pack_host: inner node %lld has bad children", (long long)i);
  }
  Packer P;
  P.nodes = nodes_host; P.pairs = pairs_host;
  const TfBvhNode& root = nodes_host[0];
  for (int k = 0; k < 3; ++k) {
    const double ext = std::max((double)root.hi[k] - (double)root.lo[k], 1e-6);
    P.margin[k] = 8e-6 * ext;
    P.org[k] = (double)root.lo[k] - 2.0 * P.margin[k];
    P.scl[k] = (ext + 4.0 * P.margin[k]) / 65535.0;
    // the kernel evaluates origin + q * scale in fp32: store fp32 values and quantise against exactly those
    frame_host[k] = (float)P.org[k]; frame_host[3 + k] = (float)P.scl[k];
    P.org[k] = frame_host[k]; P.scl[k] = frame_host[3 + k];
  }
  if (root.count > 0) {
    // the whole mesh is one leaf: a single record whose other children are empty boxes
    P.quantise(root.lo, root.hi, pairs_host);
    const float elo[3] = {1.f, 1.f, 1.f}, ehi[3] = {0.f, 0.f, 0.f};
#ifdef BVH_WIDE
    for (int k = 1; k < 4; ++k) P.quantise(elo, ehi, pairs_host + 3 * k);
    pairs_host[12] = (uint32_t)Packer::leaf_ref(root);
    for (int k = 1; k < 4; ++k) pairs_host[12 + k] = (uint32_t)-1;
#else
    P.quantise(elo, ehi, pairs_host + 3);
    pairs_host[6] = (uint32_t)Packer::leaf_ref(root); pairs_host[7] = (uint32_t)-1;
#endif
    P.n_pairs = 1;
  } else {
#ifdef BVH_WIDE
    P.emit(0);
#else
    P.emit(0, n_nodes);
#endif
  }
  for (int64_t t = 0; t < nf; ++t) {
    const float* T = tris_host + 9 * t;
    float* o = tris12_host + 12 * t;
    for (int k = 0; k < 3; ++k) { o[k] = T[k]; o[3 + k] = T[3 + k] - T[k]; o[6 + k] = T[6 + k] - T[k]; o[9 + k] = 0.f; }
  }
  return P.n_pairs;
}

// ----------------------------------------------------------------------------- device trace
#ifdef BVH_STATS
__device__ unsigned long long g_bvh_stats[8];   // inner lane-steps, leaf lane-steps, wave inner iterations x64, wave leaf iterations x64, spine entries, spine pushes, triangles tested, max steps of a ray
extern "C" void tf_bvh_stats(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bvh_stats), 64); unsigned long long z[8] = {0,0,0,0,0,0,0,0}; hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_stats), z, 64); }
#endif

// Slab test on a quantised box: plane coordinate x = org + q * scl, so t = (x - o) / d = q * (scl / d) + (org - o) / d
// = fma(q, A, B) with the per-ray constants A, B -- the de-quantisation costs nothing (one fma per plane instead of a
// subtract and a multiply).  w0, w1, w2 = {lo.x|lo.y<<16, lo.z|hi.x<<16, hi.y|hi.z<<16}.
// 16-bit field -> float in ONE instruction (sub-dword addressing: v_cvt_f32_u32 with src0_sel WORD_0 / WORD_1) instead of mask or
// shift + convert: 12 of the ~74 vector instructions of a pair step were field extraction.
__device__ __forceinline__ float lo16f(unsigned w) {
#ifdef BVH_NO_SDWA
  return (float)(w & 0xffffu);
#else
  float r;
  asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(w));
  return r;
#endif
}
__device__ __forceinline__ float hi16f(unsigned w) {
#ifdef BVH_NO_SDWA
  return (float)(w >> 16);
#else
  float r;
  asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(w));
  return r;
#endif
}

__device__ __forceinline__ bool box_hit(unsigned w0, unsigned w1, unsigned w2, float Ax, float Ay, float Az, float Bx, float By,
                                        float Bz, float tmax, float& tnear) {
  float t0 = fmaf(lo16f(w0), Ax, Bx), t1 = fmaf(hi16f(w1), Ax, Bx);
  float tmin = fminf(t0, t1), tmx = fmaxf(t0, t1);
  t0 = fmaf(hi16f(w0), Ay, By); t1 = fmaf(lo16f(w2), Ay, By);
  tmin = fmaxf(tmin, fminf(t0, t1)); tmx = fminf(tmx, fmaxf(t0, t1));
  t0 = fmaf(lo16f(w1), Az, Bz); t1 = fmaf(hi16f(w2), Az, Bz);
  tmin = fmaxf(tmin, fminf(t0, t1)); tmx = fminf(tmx, fmaxf(t0, t1));
  tnear = tmin;
  // conservative: widen by a few ulps so that a hit the exact triangle test accepts is never culled
  return tmx * 1.0000004f + 1e-6f >= fmaxf(tmin, 0.f) - 1e-6f && tmin <= tmax;
}

struct TraceArgs {
  float org[3], scl[3];        // quantisation grid of the pair boxes
  const uint4* pairs;          // two uint4 per pair
  const float4* tris;
  const float* o;
  const float* d;
  const unsigned char* live;
  long long m;
  int n_top;                   // pair records [0, n_top) are numbered breadth-first (top of the tree), the rest depth-first
  int n_pairs;
  long long rays_per_origin;   // o holds m / rays_per_origin rows; ray i starts at row i / rays_per_origin
  const int* order;            // [rays_per_origin] or null: the j-th ray traced of a point is its slot order[j]
  const int* origin_order;     // [n_origins] or null (SPINE): the u-th origin handed out is origin_order[u] (spatially sorted by the caller)
  int hits_only;               // 1: pos / nrm rows are written for rays that hit only (nobody reads a miss's row)
  int unit_parts, unit_size;   // SPINE: an origin's rays_per_origin rays are handed out in unit_parts units of <= unit_size rays
  float spine_radius;          // SPINE: ray origins lie within this distance of their origin row (|off0| + |off1| for unit directions)
  float off0, off1;
  unsigned long long* counter;
  float* pos;
  float* nrm;
  float* depth;
  unsigned char* hit;
};

#define BVH_NONE (-1)

#ifdef BVH_DEV_TMAX   // dev-only (tools/exp_bvh_tmax.py): a per-ray upper bound of the hit distance handed in from outside -- what a conservative
// occupancy pre-test would know before the first node fetch.  Results must not change; the time shows what such a test can be worth.
__device__ const float* g_bvh_tmax;
extern "C" void tf_bvh_dev_tmax(const float* p) { hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_tmax), &p, sizeof(p)); }
#endif

#ifdef BVH_CLOCK   // dev-only: where a wave's life goes.  [0] refill (ray fetch, spine, stores) [1] inner steps [2] leaf steps [3] wave lifetime,
// all in s_memrealtime ticks (100 MHz) summed over waves; [4] waves; [5] earliest wave end, [6] latest wave end, [7] earliest wave start
// [8] of the refill: result stores of finished rays (+ normal rows) [9] unit claim + spine build [10] ray fetch + spine walk
__device__ unsigned long long g_bvh_clock[12];
extern "C" void tf_bvh_clock(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bvh_clock), 96); unsigned long long z[12] = {0,0,0,0,0,~0ULL,0,~0ULL,0,0,0,0}; hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_clock), z, 96); }
#define BVH_TICK() __builtin_amdgcn_s_memrealtime()
#endif

// One lane = one ray.  A wave alternates between INNER steps (lanes whose current reference is a pair test both child
// boxes, descend into the nearer one and push the other on their LDS stack) and LEAF steps (lanes parked at a leaf
// intersect its <= 4 triangles and pop): lanes that reach a leaf wait until enough of the wave is at a leaf too, so the
// ~4x more expensive leaf body is not paid on every iteration for a couple of lanes.  DYN: persistent workgroups, a
// lane pulls a new ray from a global counter once BVH_REFILL lanes of its wave are idle (a ray needs 5..70 steps).
// The kernel is latency-bound (two thirds of a wave's life is s_waitcnt on the node fetch: rocprofv3 SQ_WAIT_ANY), so
// occupancy is the lever: only the BVH_LDS_STACK deepest-used entries of the stack live in LDS (12 KB per block instead of
// 32 KB -- a ray keeps ~4 entries pending on average, the tree is 24 deep on the bench mesh).
// SPINE (shared origins, rays_per_origin >= 64): three quarters of a ray's pair steps only re-discover where its ORIGIN sits in
// the tree -- the root-to-leaf chain of boxes that contain the origin, the same for all rays of a surface point (measured: 22.2
// pair steps per ray, ~17 of them on that chain; inner SIMD efficiency 0.52).  A wave therefore takes its rays one ORIGIN at a
// time (units of <= 512 rays of one origin), walks that chain ONCE per unit (wave-uniform: follow the child whose box contains
// the ball of possible ray origins, record the sibling's box and reference in LDS) and every ray of the unit starts with a
// uniform loop of single-box tests over the recorded siblings (no dependent fetch, no divergence: all starting lanes run the
// same list) that pushes the siblings it hits; the per-ray traversal proper starts at the chain's last node.  ANY root-to-node
// chain is a valid start (its siblings are box-tested, its end is traversed unconditionally), so the ball only steers speed:
// results are bit-identical to the plain traversal.
#define BVH_SPINE_MAX 28
// ... and the chain is followed (past the point where the ball straddles a split: any chain is valid) until the subtree below it
// has at most BVH_NEAR pair records -- contiguous in memory (the packer numbers everything below the breadth-first top of the
// tree depth-first) -- which the wave copies into LDS.  The kernel is bound by the per-CU vector L1 (one divergent 16-byte access
// per clock: ~57 accesses per ray before, two per pair step); the steps a ray spends around its own origin are now LDS reads.
// Measured on the bench (201 M rays, one box, ms per launch): no spine 17.6; spine while the ball fits (the default) 17.1;
// + units claimed 16 per atomic 18.3 (the counter alone costs 6 ms with the traversal switched off, 2.7 ms grouped -- and the
// whole kernel still gets slower); + chain followed to the origin's leaf region, no LDS subtree 17.6; + LDS subtree of 32 records
// 18.4.  Pair steps per ray fall 20.4 -> 11.2 and wave iterations 35 -> 21 with the full form while the time does not move:
// what is left is the latency of the deep, L2-missing fetches (TCC hit rate 66 % -> 60 %), not the count of steps.
#ifndef BVH_NEAR
#define BVH_NEAR 0          // records of the LDS subtree (0: off)
#endif
#ifndef BVH_SPINE_FOLLOW
#define BVH_SPINE_FOLLOW 0  // 1: keep following the child that holds the origin row once the ball straddles a split
#endif
#ifndef BVH_UNIT_GROUP
#define BVH_UNIT_GROUP 1    // units claimed per atomic
#endif
#ifndef BVH_XCD_POOLS
#define BVH_XCD_POOLS 0     // 1: one unit pool per XCD over Morton-sorted origins (origin_order).  Measured 18.0-18.3 ms vs 17.1-17.3: the
                            // traversal does not respond to L2 locality either (TCC hit rate 60-66 % in every variant)
#endif
#ifndef BVH_STAGE
#define BVH_STAGE 0         // 1 (dev switch): SPINE, rays of a point contiguous (no slot_order): the direction rows and live flags of the wave's
                            // next 64 rays are copied into LDS by LDS-DMA one refill AHEAD of their use, so that a refill (600 per wave on
                            // the bench) does not stall every lane of the wave on a fetch from HBM.  Results identical; measured SLOWER
                            // (17.3 vs 15.2 ms per 201 M rays): the fetch was not what a refill waits for (its spine walk and start-up
                            // arithmetic are), and the window re-fetch, 10 more spilled registers and the extra waits cost more
#endif
template <bool DYN, bool SPINE, bool STAGED = false>
__global__ void __launch_bounds__(256, BVH_WAVES) bvh_trace_kernel(TraceArgs A) {
  __shared__ int stack[(BVH_LDS_STACK + 1) * 256];   // + one dummy row: the target of predicated-off pushes
  __shared__ float dstage[STAGED ? 4 * 2 * 256 : 1];      // per wave: two windows of 64 direction rows (LDS-DMA of 12 bytes per lane lands at a 16-byte stride)
  __shared__ unsigned lstage[STAGED ? 4 * 2 * 64 : 1];    // ... and of 64 live flags (one dword each)
  long long st_base = -1;                            // wave-uniform: first ray of the window staged last
  int st_buf = 0;
  float uox = 0.f, uoy = 0.f, uoz = 0.f;             // wave-uniform: origin row of the current unit
  __shared__ uint4 spine[SPINE ? 4 * BVH_SPINE_MAX : 1];
  __shared__ uint4 near_tree[SPINE && BVH_NEAR > 0 ? 4 * 2 * BVH_NEAR : 1];   // per wave: the records of the subtree around the unit's origin
  int spine_n = 0, spine_end = 0;                    // wave-uniform
  int near_base = 0, near_size = 0;                  // wave-uniform: pair records [near_base, near_base + near_size) are in near_tree
  int deep[BVH_STACK - BVH_LDS_STACK];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef BVH_TOP_LDS   // dev-only experiment: a third of every ray's steps touch the same ~127 records at the top of the tree (the
  // packer numbers them breadth-first, first); serving them from LDS changed nothing (19.7 vs 19.2 ms): node fetches are not the bound
  __shared__ uint4 top[2 * BVH_TOP];
  for (int i = tid; i < 2 * A.n_top; i += 256) top[i] = A.pairs[i];
  __syncthreads();
#endif
  const unsigned long long lt_mask = (1ULL << lane) - 1ULL;
  long long rid = -1;
  int cur = BVH_NONE, sp = 0, best_tri = -1;
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, best = BVH_MAX_DIST;
  float Ax = 0, Ay = 0, Az = 0, Bx = 0, By = 0, Bz = 0;   // slab-test constants of the current ray
  bool exhausted = false;
#if BVH_LEAF_TOUCH
  float touch = 0.f;
#endif
  long long q_next = 0, q_end = 0;   // wave-uniform: this wave's private chunk of the ray pool
  long long u_next = 0, u_end = 0;   // SPINE: this wave's private range of units
  int ugrab = BVH_UNIT_GROUP;
#if BVH_XCD_POOLS
  int my_pool;
  { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); my_pool = (int)(xcc & 7u); }
#endif
  int grab = BVH_CHUNK_MAX;
#ifdef BVH_STATS
  unsigned st_inner = 0, st_leaf = 0, st_wi = 0, st_wl = 0, st_spine = 0, st_spush = 0, st_walks = 0, st_ray = 0, st_max = 0, st_tris = 0;
#endif
  auto start_ray = [&](long long seq) {
    // trace order -> ray id: consecutive lanes take the slots of one point in `order` (directions sorted along a
    // space-filling curve by the caller), so a wave's rays share an origin AND point the same way
    const long long oid = seq / A.rays_per_origin;
    const long long id = A.order ? oid * A.rays_per_origin + A.order[seq - oid * A.rays_per_origin] : seq;
    rid = id;
    dx = A.d[3 * id]; dy = A.d[3 * id + 1]; dz = A.d[3 * id + 2];
    // origin = (o + d*off0) + off1*d, with the reference's two separate roundings (no fma contraction)
    ox = __fadd_rn(__fadd_rn(A.o[3 * oid], __fmul_rn(dx, A.off0)), __fmul_rn(A.off1, dx));
    oy = __fadd_rn(__fadd_rn(A.o[3 * oid + 1], __fmul_rn(dy, A.off0)), __fmul_rn(A.off1, dy));
    oz = __fadd_rn(__fadd_rn(A.o[3 * oid + 2], __fmul_rn(dz, A.off0)), __fmul_rn(A.off1, dz));
    const float ix = 1.f / dx, iy = 1.f / dy, iz = 1.f / dz;
    Ax = A.scl[0] * ix; Ay = A.scl[1] * iy; Az = A.scl[2] * iz;
    Bx = (A.org[0] - ox) * ix; By = (A.org[1] - oy) * iy; Bz = (A.org[2] - oz) * iz;
    best = BVH_MAX_DIST; best_tri = -1; sp = 0;
#ifdef BVH_DEV_TMAX
    if (g_bvh_tmax) best = g_bvh_tmax[id];
#endif
    cur = 0;
    if (A.live && !A.live[id]) cur = BVH_NONE;   // zero weight in the integral: reported as a miss, never traversed
#ifdef BVH_ABLATE_TRAVERSE   // dev-only timing ablation: ray fetch + scheduling + result stores only
    cur = BVH_NONE;
#endif
  };
  constexpr bool staged = STAGED;
  auto stage_issue = [&](long long seq0) {           // window [seq0, seq0 + 64) -> the buffer not read at the moment
    st_buf ^= 1; st_base = seq0;
    const long long row = min(seq0 + lane, A.m - 1);
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) float*)(dstage + ((tid >> 6) * 2 + st_buf) * 256));
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx3 %1, off" ::"s"(la), "v"(A.d + 3 * row) : "memory");
    if (A.live) {
      const unsigned lb = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned*)(lstage + ((tid >> 6) * 2 + st_buf) * 64));
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_ubyte %1, off" ::"s"(lb), "v"(A.live + row) : "memory");
    }
  };
  auto start_ray_staged = [&](long long seq, int k) {   // ray `seq` = row k of the staged window; its origin row is the unit's
    rid = seq;
    const float* dsrc = dstage + ((tid >> 6) * 2 + st_buf) * 256 + 4 * k;
    dx = dsrc[0]; dy = dsrc[1]; dz = dsrc[2];
    ox = __fadd_rn(__fadd_rn(uox, __fmul_rn(dx, A.off0)), __fmul_rn(A.off1, dx));
    oy = __fadd_rn(__fadd_rn(uoy, __fmul_rn(dy, A.off0)), __fmul_rn(A.off1, dy));
    oz = __fadd_rn(__fadd_rn(uoz, __fmul_rn(dz, A.off0)), __fmul_rn(A.off1, dz));
    const float ix = 1.f / dx, iy = 1.f / dy, iz = 1.f / dz;
    Ax = A.scl[0] * ix; Ay = A.scl[1] * iy; Az = A.scl[2] * iz;
    Bx = (A.org[0] - ox) * ix; By = (A.org[1] - oy) * iy; Bz = (A.org[2] - oz) * iz;
    best = BVH_MAX_DIST; best_tri = -1; sp = 0;
    cur = 0;
    if (A.live && !lstage[((tid >> 6) * 2 + st_buf) * 64 + k]) cur = BVH_NONE;
#ifdef BVH_ABLATE_TRAVERSE
    cur = BVH_NONE;
#endif
  };
  auto push = [&](int ref) {
    if (sp < BVH_LDS_STACK) stack[sp * 256 + tid] = ref;
    else deep[sp - BVH_LDS_STACK] = ref;
    ++sp;
  };
  auto pop = [&]() -> int {
    if (sp == 0) return BVH_NONE;
    --sp;
    // always a plain LDS read; the scratch tail is a separate, rarely taken branch.  Written as one conditional expression the
    // two arrays become one generic pointer and the pop a flat_load behind `s_waitcnt vmcnt(0) lgkmcnt(0)` -- on the dependent
    // chain of every second traversal step.
    int v = stack[min(sp, BVH_LDS_STACK - 1) * 256 + tid];
    asm volatile("" : "+v"(v));                 // keeps the LDS read an instruction of its own (not a select of two pointers)
    if (sp >= BVH_LDS_STACK) v = deep[sp - BVH_LDS_STACK];
    return v;
  };
  auto retire = [&]() {
#ifdef BVH_STATS
    st_max = max(st_max, st_ray); st_ray = 0;
#endif
    // a ray hit iff a triangle was accepted: `best` may start below BVH_MAX_DIST (an upper bound of the hit distance known before the
    // traversal), so the distance alone no longer tells
    if (best_tri < 0) best = BVH_MAX_DIST;
#ifdef BVH_ABLATE_STORE   // dev-only timing ablation: results are not written
    if (best == -123.f) A.depth[rid] = best;
    rid = -1;
    return;
#endif
#ifdef BVH_ABLATE_MISS_DEPTH   // dev-only timing ablation: depth stored for the rays that hit only
    if (best < BVH_MAX_DIST) A.depth[rid] = best;
#else
    A.depth[rid] = best;
#endif
    if (A.hit) A.hit[rid] = best < BVH_MAX_DIST ? 1 : 0;
    // 85 % of the integral's secondary rays miss; their position / normal rows are never read, and writing them cost 12 %
    // of the kernel (scattered partial-line stores: 4x write amplification at the fabric)
    const bool rows = !A.hits_only || best < BVH_MAX_DIST;
    if (A.pos && rows) { A.pos[3 * rid] = ox + best * dx; A.pos[3 * rid + 1] = oy + best * dy; A.pos[3 * rid + 2] = oz + best * dz; }
    if (A.nrm && rows) {
      float nx = 0.f, ny = 0.f, nz = 0.f;
      if (best_tri >= 0) {
        const float4 t0 = A.tris[3 * (long long)best_tri], t1 = A.tris[3 * (long long)best_tri + 1], t2 = A.tris[3 * (long long)best_tri + 2];
        const float e1x = t0.w, e1y = t1.x, e1z = t1.y, e2x = t1.z, e2y = t1.w, e2z = t2.x;
        float fx = e1y * e2z - e1z * e2y, fy = e1z * e2x - e1x * e2z, fz = e1x * e2y - e1y * e2x;
        float inv = 1.f / fmaxf(sqrtf(fx * fx + fy * fy + fz * fz), 1e-12f);   // face normal (raytracing)
        fx = -fx * inv; fy = -fy * inv; fz = -fz * inv;                         // materialRenderer.py:256
        inv = 1.f / fmaxf(sqrtf(fx * fx + fy * fy + fz * fz), 1e-12f);          // F.normalize (:257)
        nx = fx * inv; ny = fy * inv; nz = fz * inv;
      }
      A.nrm[3 * rid] = nx; A.nrm[3 * rid + 1] = ny; A.nrm[3 * rid + 2] = nz;
    }
    rid = -1;
  };
  // ---- SPINE helpers
  auto build_spine = [&](long long oid) {             // wave-uniform: every lane runs it on the same origin
    const float px = A.o[3 * oid], py = A.o[3 * oid + 1], pz = A.o[3 * oid + 2];
    uint4* my = spine + (tid >> 6) * BVH_SPINE_MAX;
    int node = 0, n = 0, end = A.n_pairs;              // records of the subtree under `node`: [node, end) once node >= n_top
    near_size = 0;
    while (n < BVH_SPINE_MAX) {
      if (BVH_NEAR > 0 && node >= A.n_top && end - node <= BVH_NEAR) break;   // small enough: this subtree goes to LDS, per-ray traversal starts here
      const uint4 q0 = A.pairs[2LL * node], q1 = A.pairs[2LL * node + 1];
      const int c0 = (int)q1.z, c1 = (int)q1.w;
      auto inside = [&](unsigned w0, unsigned w1, unsigned w2, float r) {
        const float lx = fmaf((float)(w0 & 0xffffu), A.scl[0], A.org[0]), hx = fmaf((float)(w1 >> 16), A.scl[0], A.org[0]);
        const float ly = fmaf((float)(w0 >> 16), A.scl[1], A.org[1]), hy = fmaf((float)(w2 & 0xffffu), A.scl[1], A.org[1]);
        const float lz = fmaf((float)(w1 & 0xffffu), A.scl[2], A.org[2]), hz = fmaf((float)(w2 >> 16), A.scl[2], A.org[2]);
        return lx <= px - r && px + r <= hx && ly <= py - r && py + r <= hy && lz <= pz - r && pz + r <= hz;
      };
      // prefer the child that holds the whole ball of ray origins; where the ball straddles the split, the child that holds the
      // origin row itself (a ray whose own origin lies in the sibling finds it through the sibling's box test like any other hit)
      bool in0 = c0 != BVH_NONE && inside(q0.x, q0.y, q0.z, A.spine_radius), in1 = c1 != BVH_NONE && inside(q0.w, q1.x, q1.y, A.spine_radius);
      if (!in0 && !in1) {
#if BVH_SPINE_FOLLOW
        in0 = c0 != BVH_NONE && inside(q0.x, q0.y, q0.z, 0.f); in1 = c1 != BVH_NONE && inside(q0.w, q1.x, q1.y, 0.f);
        if (!in0 && !in1) break;
#else
        break;
#endif
      }
      const bool f0 = in0;                             // both contain it: either is valid, take child 0
      const int sib = f0 ? c1 : c0, next = f0 ? c0 : c1;
      if (next < 0) break;                             // the chain would end in a leaf: stop one level above (traversal from `node`)
      if (sib != BVH_NONE) {
        if (lane == 0) my[n] = f0 ? make_uint4(q0.w, q1.x, q1.y, (unsigned)sib) : make_uint4(q0.x, q0.y, q0.z, (unsigned)sib);
        ++n;
      }
      // depth-first numbering below the top: left subtree = [c0, c1), right subtree = [c1, end)
      if (node >= A.n_top && f0 && c1 >= 0) end = c1;
      else if (node < A.n_top) end = A.n_pairs;        // inside the breadth-first top the range is not contiguous yet
      node = next;
    }
    spine_n = n; spine_end = node;
    if (BVH_NEAR > 0 && node >= A.n_top && end - node <= BVH_NEAR && end > node) {
      near_base = node; near_size = end - node;
      uint4* nt = near_tree + (tid >> 6) * 2 * BVH_NEAR;
      for (int i = lane; i < 2 * near_size; i += 64) nt[i] = A.pairs[2LL * node + i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // list and subtree are in LDS before any lane of this wave reads them
  };
  auto spine_walk = [&]() {                            // by every lane that has just started a ray of the current unit
    const uint4* my = spine + (tid >> 6) * BVH_SPINE_MAX;
    for (int e = 0; e < spine_n; ++e) {
      const uint4 sb = my[e];
      float tn;
      if (box_hit(sb.x, sb.y, sb.z, Ax, Ay, Az, Bx, By, Bz, best, tn)) {
        push((int)sb.w);
#ifdef BVH_STATS
        st_spush++;
#endif
      }
    }
#ifdef BVH_STATS
    st_spine += spine_n; if (lane == __ffsll(__ballot(1)) - 1) st_walks++;
#endif
    cur = spine_end;
  };
  if (!DYN) {
    const long long i = (long long)blockIdx.x * 256 + tid;
    if (i < A.m) start_ray(i);
    exhausted = true;
  }
#ifdef BVH_CLOCK
  const unsigned long long ck_start = BVH_TICK();
  unsigned long long ck_refill = 0, ck_inner = 0, ck_leaf = 0, ck_t = ck_start, ck_retire = 0, ck_claim = 0, ck_startray = 0;
#endif
  while (true) {
#ifdef BVH_CLOCK
    ck_t = BVH_TICK();
#endif
    if (rid >= 0 && cur == BVH_NONE) retire();
#ifdef BVH_CLOCK
    { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ck_retire += BVH_TICK() - ck_t; }
#endif
    if (DYN && !exhausted) {
      // ---- fetch new rays for idle lanes from the wave's private chunk [q_next, q_end); the chunk itself comes from ONE
      // global atomic per BVH_CHUNK_MIN..BVH_CHUNK_MAX rays (a single counter word sustains only ~88 M atomics/s -- one
      // atomic per refill made the counter, not the traversal, the limit).  Dead rays are retired on the spot and the
      // lane asks again.
      for (int round = 0; round < 4 && !exhausted; ++round) {
        const bool want = rid < 0;
        const unsigned long long wb = __ballot(want);
        const int nw = __popcll(wb);
        if (nw == 0 || (round > 0 && nw < BVH_REFILL)) break;
        if (q_next >= q_end) {
#ifdef BVH_CLOCK
          const unsigned long long ck_c0 = BVH_TICK();
#endif
          unsigned long long base = 0;
          if (SPINE) {
            // one unit = the rays [part * unit_size, (part + 1) * unit_size) of ONE origin: all rays started from it share a spine.
            // Units are claimed several at a time: a single counter word sustains ~88 M atomics/s, and one atomic per 384-ray
            // unit (524 k of them per 201 M rays) held the whole kernel at 6 ms with the traversal switched off.
            const long long n_units = (A.m / A.rays_per_origin) * A.unit_parts;
            if (u_next >= u_end) {
#if BVH_XCD_POOLS
              // eight pools, one per XCD: the units (origins in the caller's spatial order) are cut into eight contiguous ranges and
              // a wave drains the range of the XCD it runs on first -- its XCD's L2 then holds one eighth of the mesh's tree and
              // triangles instead of all of it -- before helping the others out (placement only steers speed, never results)
              bool got = false;
              for (int t = 0; t < 8 && !got; ++t) {
                const int pool = (my_pool + t) & 7;
                const long long p_lo = n_units * pool / 8, p_hi = n_units * (pool + 1) / 8;
                if (lane == 0) base = atomicAdd(A.counter + pool, (unsigned long long)ugrab);
                base = __shfl(base, 0);
                if (p_lo + (long long)base < p_hi) {
                  u_next = p_lo + (long long)base;
                  u_end = min(u_next + ugrab, p_hi);
                  got = true;
                } else if (t == 0) {
                  my_pool = (my_pool + 1) & 7;        // own range is empty for good: start from the next one from now on
                }
              }
              if (!got) { exhausted = true; break; }
#else
              if (lane == 0) base = atomicAdd(A.counter, (unsigned long long)ugrab);
              base = __shfl(base, 0);
              if (base >= (unsigned long long)n_units) { exhausted = true; break; }
              u_next = (long long)base;
              u_end = min(u_next + ugrab, n_units);
              const long long share = (n_units - u_end) / (4LL * gridDim.x * 4);       // guided: shrinks as the pool drains
              ugrab = (int)min((long long)BVH_UNIT_GROUP, max(1LL, share));
#endif
            }
            base = (unsigned long long)u_next++;
            const long long ou = (long long)base / A.unit_parts, part = (long long)base - ou * A.unit_parts;
            const long long oid = A.origin_order ? (long long)A.origin_order[ou] : ou;
            q_next = oid * A.rays_per_origin + part * A.unit_size;
            q_end = min(q_next + A.unit_size, (oid + 1) * A.rays_per_origin);
            if (staged) {
              stage_issue(q_next);                   // in flight while the spine is walked
              // one origin row per unit, the same for every lane: kept in scalar registers
              uox = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(A.o[3 * oid])));
              uoy = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(A.o[3 * oid + 1])));
              uoz = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(A.o[3 * oid + 2])));
            }
            build_spine(oid);
          } else {
          if (lane == 0) base = atomicAdd(A.counter, (unsigned long long)grab);
          base = __shfl(base, 0);
          if (base >= (unsigned long long)A.m) { exhausted = true; break; }
          q_next = (long long)base;
          q_end = min((long long)base + grab, A.m);
          // guided self-scheduling: chunks shrink as the pool drains so the last chunk of the slowest wave stays short
          const long long share = (A.m - q_end) / (4LL * gridDim.x * 4);
          grab = (int)min((long long)BVH_CHUNK_MAX, max((long long)BVH_CHUNK_MIN, share));
          }
#ifdef BVH_CLOCK
          ck_claim += BVH_TICK() - ck_c0;
#endif
        }
#ifdef BVH_CLOCK
        const unsigned long long ck_s0 = BVH_TICK();
#endif
        if (staged) {
          // the window [q_next, q_next + 64) was requested at the end of the previous refill (or by the unit claim above, whose
          // spine build ends in a full wait); a second round of the same refill reads a window requested a moment ago
          if (st_base != q_next) stage_issue(q_next);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // free when nothing is in flight
          if (want) {
            const int k = __popcll(wb & lt_mask);
            const long long id = q_next + k;
            if (id < q_end) {
              start_ray_staged(id, k);
              if (cur == BVH_NONE) retire();
              else spine_walk();
            }
          }
          q_next = min(q_next + nw, q_end);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the window has been read before its buffer can be targeted again (two refills on)
          if (q_next < q_end) stage_issue(q_next);
        } else {
        if (want) {
          const long long id = q_next + __popcll(wb & lt_mask);
          if (id < q_end) {
            start_ray(id);
            if (cur == BVH_NONE) retire();
            else if (SPINE) spine_walk();
          }
        }
        q_next = min(q_next + nw, q_end);
        }
#ifdef BVH_CLOCK
        { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ck_startray += BVH_TICK() - ck_s0; }
#endif
      }
    }
#ifdef BVH_CLOCK
    { const unsigned long long t = BVH_TICK(); ck_refill += t - ck_t; ck_t = t; }
#endif
    if (__ballot(rid >= 0) == 0ULL) {
      if (exhausted) break;
      continue;
    }
    // ---- traverse until enough lanes have finished to make a refill worthwhile
    while (true) {
      const int n_inner = __popcll(__ballot(cur >= 0));
      const int n_leaf = __popcll(__ballot(cur < BVH_NONE));
      if (n_inner + n_leaf == 0) break;
#ifdef BVH_CLOCK
      const bool ck_is_leaf = n_leaf > 0 && (n_inner == 0 || n_leaf * BVH_LEAF_W >= n_inner);
#endif
      if (n_leaf > 0 && (n_inner == 0 || n_leaf * BVH_LEAF_W >= n_inner)) {
#ifdef BVH_STATS
        st_wl++;
#endif
        if (cur < BVH_NONE) {
#ifdef BVH_STATS
          st_leaf++; st_tris += (~cur) & 7;
#endif
          const int enc = ~cur;
          const int first = enc >> 3, cnt = enc & 7;
          // Moeller-Trumbore on a full-precision (a, e1, e2) record, the arithmetic of the brute-force oracle (hits and depths are exact)
          auto tri_test = [&](const float4 t0, const float4 t1, const float4 t2, int id) {
            const float ax = t0.x, ay = t0.y, az = t0.z;
            const float e1x = t0.w, e1y = t1.x, e1z = t1.y, e2x = t1.z, e2y = t1.w, e2z = t2.x;
            const float rx = ox - ax, ry = oy - ay, rz = oz - az;
            const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
            const float qx = ry * dz - rz * dy, qy = rz * dx - rx * dz, qz = rx * dy - ry * dx;
            const float det = 1.f / (dx * nx + dy * ny + dz * nz);
            const float u = det * -(qx * e2x + qy * e2y + qz * e2z);
            const float v = det * (qx * e1x + qy * e1y + qz * e1z);
            const float t = det * -(nx * rx + ny * ry + nz * rz);
            if (u >= 0.f && u <= 1.f && v >= 0.f && u + v <= 1.f && t >= 0.f && t < best) { best = t; best_tri = id; }
          };
#ifdef BVH_ABLATE_LEAF    // dev-only timing ablation: no triangle is tested
          const int cnt_t = 0;
#else
          const int cnt_t = cnt;
#endif
          const float4* T = A.tris + 3LL * first;
#if BVH_LEAF_PAIRED
          // leaves hold <= 2 triangles by default: both records are requested before either is tested (one memory round trip per leaf
          // instead of one per triangle: three quarters of the triangle fetches miss the L2 -- 12.7 MB of records against 4 MB)
          if (cnt_t > 0) {
            const float4 a0 = T[0], a1 = T[1], a2 = T[2];
            float4 b0 = a0, b1 = a1, b2 = a2;
            if (cnt_t > 1) { b0 = T[3]; b1 = T[4]; b2 = T[5]; }
            tri_test(a0, a1, a2, first);
            if (cnt_t > 1) tri_test(b0, b1, b2, first + 1);
          }
          for (int k = 2; k < cnt_t; ++k) tri_test(T[3 * k], T[3 * k + 1], T[3 * k + 2], first + k);
#else
          for (int k = 0; k < cnt_t; ++k) tri_test(T[3 * k], T[3 * k + 1], T[3 * k + 2], first + k);
#endif
          cur = pop();
        }
      } else {
#ifndef BVH_INNER_REP
#define BVH_INNER_REP 4    // inner steps per scheduling decision: the two ballots + the refill test cost ~30 scalar instructions (1: 19.2 ms, 2: 18.9, 3: 18.6, 4: 18.5 per 201 M rays)
#endif
#pragma unroll
        for (int rep = 0; rep < BVH_INNER_REP; ++rep)
#ifdef BVH_STATS
        { st_wi++;
#endif
        if (cur >= 0) {
#ifdef BVH_STATS
          st_inner++; st_ray++;
#endif
#ifdef BVH_WIDE
          const uint4* P = A.pairs + 4LL * cur;
          const uint4 q0 = P[0], q1 = P[1], q2 = P[2], q3 = P[3];
          float t0, t1, t2, t3;
          const bool h0 = box_hit(q0.x, q0.y, q0.z, Ax, Ay, Az, Bx, By, Bz, best, t0);
          const bool h1 = box_hit(q0.w, q1.x, q1.y, Ax, Ay, Az, Bx, By, Bz, best, t1);
          const bool h2 = box_hit(q1.z, q1.w, q2.x, Ax, Ay, Az, Bx, By, Bz, best, t2);
          const bool h3 = box_hit(q2.y, q2.z, q2.w, Ax, Ay, Az, Bx, By, Bz, best, t3);
          int r0 = (int)q3.x, r1 = (int)q3.y, r2 = (int)q3.z, r3 = (int)q3.w;
          // an unused slot (reference -1) holds an inverted box, which the symmetric slab test does NOT reject: mask it here
          float k0 = (h0 && r0 != BVH_NONE) ? t0 : INFINITY, k1 = (h1 && r1 != BVH_NONE) ? t1 : INFINITY;
          float k2 = (h2 && r2 != BVH_NONE) ? t2 : INFINITY, k3 = (h3 && r3 != BVH_NONE) ? t3 : INFINITY;
          // sort the four (entry distance, reference) pairs ascending: 5-comparator network
#define BVH_CSWAP(ka, ra, kb, rb) { const bool sw_ = kb < ka; const float tk_ = sw_ ? kb : ka; kb = sw_ ? ka : kb; ka = tk_; \
                                    const int tr_ = sw_ ? rb : ra; rb = sw_ ? ra : rb; ra = tr_; }
          BVH_CSWAP(k0, r0, k1, r1) BVH_CSWAP(k2, r2, k3, r3) BVH_CSWAP(k0, r0, k2, r2) BVH_CSWAP(k1, r1, k3, r3) BVH_CSWAP(k1, r1, k2, r2)
#undef BVH_CSWAP
          // postpone the farther hit children (farthest pushed first), descend into the nearest
          if (k3 < INFINITY) push(r3);
          if (k2 < INFINITY) push(r2);
          if (k1 < INFINITY) push(r1);
          cur = k0 < INFINITY ? r0 : pop();
#else
          uint4 q0, q1;
#ifdef BVH_TOP_LDS
          if (cur < A.n_top) {
            q0 = top[2 * cur]; q1 = top[2 * cur + 1];
          } else
#endif
          if (SPINE && BVH_NEAR > 0 && (unsigned)(cur - near_base) < (unsigned)near_size) {
            const uint4* P = near_tree + (tid >> 6) * 2 * BVH_NEAR + 2 * (cur - near_base);
            q0 = P[0]; q1 = P[1];
          } else {
            const uint4* P = A.pairs + 2LL * cur;
            q0 = P[0]; q1 = P[1];
          }
          const int c0 = (int)q1.z, c1 = (int)q1.w;
          float tl, tr;
          const bool hl = box_hit(q0.x, q0.y, q0.z, Ax, Ay, Az, Bx, By, Bz, best, tl);
          const bool hr = box_hit(q0.w, q1.x, q1.y, Ax, Ay, Az, Bx, By, Bz, best, tr);
#ifndef BVH_PREDICATED   // default: one branch per case
          if (hl && hr) {
            const bool left_first = tl <= tr;
            push(left_first ? c1 : c0);                             // depth <= BVH_MAX_DEPTH bounds sp < BVH_STACK
            cur = left_first ? c0 : c1;
          } else if (hl) {
            cur = c0;
          } else if (hr) {
            cur = c1;
          } else {
            cur = pop();
          }
#if BVH_LEAF_TOUCH
          // dev experiment (round 4): a lane that has just arrived at a leaf requests the first dword of its triangle records NOW -- the
          // leaf step proper runs iterations later, when enough lanes wait at leaves, and three quarters of the record fetches miss the L2
          if (cur < BVH_NONE) {
            const int enc_t = ~cur;
            const float* tp = reinterpret_cast<const float*>(A.tris + 3LL * (enc_t >> 3));
            touch += tp[0] * 0.f + tp[11 + 12 * (((enc_t & 7) > 1) ? 1 : 0)] * 0.f;
          }
#endif
#else
          // dev-only experiment (-DBVH_PREDICATED): the four cases as selects, the push an unconditional LDS write (to a dummy row
          // when predicated off), the pop an unconditional LDS read, one rare wave-uniform branch for the scratch tail.  Results
          // identical, time unchanged (19.7 vs 19.2 ms): the lane-mask algebra moves to the scalar unit instead of the branches
          // (55 vs 59 scalar, 84 vs 74 vector instructions per step).
          const bool both = hl && hr, any = hl || hr;
          const bool left_first = tl <= tr;
          const int near = (hl && (!hr || left_first)) ? c0 : c1;
          const int far = left_first ? c1 : c0;
          const int sp0 = sp;
          const bool deep_push = both && sp0 >= BVH_LDS_STACK, deep_pop = !any && sp0 > BVH_LDS_STACK;
          stack[((both && !deep_push) ? sp0 : BVH_LDS_STACK) * 256 + tid] = far;
          int popped = stack[min(max(sp0 - 1, 0), BVH_LDS_STACK - 1) * 256 + tid];
          if (__builtin_expect(__ballot(deep_push || deep_pop) != 0ULL, 0)) {
            if (deep_push) deep[sp0 - BVH_LDS_STACK] = far;
            if (deep_pop) popped = deep[sp0 - 1 - BVH_LDS_STACK];
          }
          cur = any ? near : (sp0 > 0 ? popped : BVH_NONE);
          sp = any ? sp0 + (both ? 1 : 0) : max(sp0 - 1, 0);
#endif
#endif
        }
#ifdef BVH_STATS
        }
#endif
      }
#ifdef BVH_CLOCK
      { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = BVH_TICK(); if (ck_is_leaf) ck_leaf += t - ck_t; else ck_inner += t - ck_t; ck_t = t; }
#endif
      if (DYN && !exhausted && __popcll(__ballot(cur == BVH_NONE)) >= BVH_REFILL) break;
    }
  }
#if BVH_LEAF_TOUCH
  if (touch == 123.456f) A.depth[0] = touch;      // (never true: keeps the touch loads alive)
#endif
#ifdef BVH_CLOCK
  if (lane == 0) {
    const unsigned long long t = BVH_TICK();
    atomicAdd(&g_bvh_clock[0], ck_refill); atomicAdd(&g_bvh_clock[1], ck_inner); atomicAdd(&g_bvh_clock[2], ck_leaf);
    atomicAdd(&g_bvh_clock[3], t - ck_start); atomicAdd(&g_bvh_clock[4], 1ULL);
    atomicMin(&g_bvh_clock[5], t); atomicMax(&g_bvh_clock[6], t); atomicMin(&g_bvh_clock[7], ck_start);
    atomicAdd(&g_bvh_clock[8], ck_retire); atomicAdd(&g_bvh_clock[9], ck_claim); atomicAdd(&g_bvh_clock[10], ck_startray);
  }
#endif
#ifdef BVH_STATS
  atomicAdd(&g_bvh_stats[0], (unsigned long long)st_inner); atomicAdd(&g_bvh_stats[1], (unsigned long long)st_leaf);
  if (lane == 0) { atomicAdd(&g_bvh_stats[2], 64ULL * st_wi); atomicAdd(&g_bvh_stats[3], 64ULL * st_wl); }
  atomicMax(&g_bvh_stats[7], (unsigned long long)st_max); atomicAdd(&g_bvh_stats[6], 0ULL);
  atomicAdd(&g_bvh_stats[4], (unsigned long long)st_spine); atomicAdd(&g_bvh_stats[5], (unsigned long long)st_spush); atomicAdd(&g_bvh_stats[6], (unsigned long long)st_tris);
#endif
}

extern "C" int tf_bvh_trace(const uint32_t* pairs, const float* tris12, const float* frame_host, int64_t n_pairs, const float* o, const float* d,
                            int64_t rays_per_origin, const int32_t* slot_order, float origin_offset0, float origin_offset1, const uint8_t* live,
                            int64_t m, float* pos, float* nrm, float* depth, uint8_t* hit, int32_t hit_rows_only,
                            const int32_t* origin_order, int64_t* work_counter, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(m >= 0 && n_pairs > 0, TF_ESHAPE, "tf_bvh_trace: m < 0 or empty BVH");
  TF_REQUIRE(rays_per_origin >= 1, TF_ESHAPE, "tf_bvh_trace: rays_per_origin must be >= 1");
  if (m == 0) return TF_OK;
  TF_REQUIRE(pairs && tris12 && frame_host && o && d && depth, TF_EINVAL, "tf_bvh_trace: null pointer");
  TF_REQUIRE((((uintptr_t)pairs | (uintptr_t)tris12) & 15) == 0, TF_EINVAL, "tf_bvh_trace: pairs / tris12 must be 16-byte aligned");
  TraceArgs A;
  A.pairs = reinterpret_cast<const uint4*>(pairs); A.tris = reinterpret_cast<const float4*>(tris12);
  for (int k = 0; k < 3; ++k) { A.org[k] = frame_host[k]; A.scl[k] = frame_host[3 + k]; }
  A.n_pairs = (int)n_pairs;
  A.o = o; A.d = d; A.live = live; A.m = m; A.rays_per_origin = rays_per_origin; A.order = slot_order; A.origin_order = origin_order;
#ifdef BVH_NO_TOP   // dev-only switch: every record from global memory
  A.n_top = 0;
#else
  A.n_top = (int)std::min<int64_t>(BVH_TOP, n_pairs);
#endif
  TF_REQUIRE(!slot_order || m % rays_per_origin == 0, TF_ESHAPE, "tf_bvh_trace: slot_order needs m to be a multiple of rays_per_origin");
  A.off0 = origin_offset0; A.off1 = origin_offset1; A.counter = (unsigned long long*)work_counter;
  A.pos = pos; A.nrm = nrm; A.depth = depth; A.hit = hit; A.hits_only = hit_rows_only ? 1 : 0;
  if (work_counter) {
    hipError_t e = hipMemsetAsync(work_counter, 0, 8 * sizeof(int64_t), stream);     // work_counter[8]: one pool per XCD
    TF_REQUIRE(e == hipSuccess, TF_EHIP, "tf_bvh_trace: hipMemsetAsync failed: %s", hipGetErrorString(e));
    static int resident = 0;    // blocks per CU the hardware admits (registers / LDS), queried once
    if (!resident) {
      int nb = 0;
      hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)bvh_trace_kernel<true, true, true>, 256, 0);
      resident = (e2 == hipSuccess && nb > 0) ? (nb > 8 ? 8 : nb) : 4;
      // Round 5, measured (tools/exp_bvh_k.py: budgets interleaved in one process, ms per 50 M / 201 M rays of the bench): 5 workgroups
      // per CU 4.51 / 17.3, 6: 4.09 / 15.6, all 7 that fit: 3.86 / 14.4; 1-4: 15.9 / 8.8 / 6.4 / 5.2 per 50 M (tools/exp_cosched.py).  Every
      // wave slot still pays: the default is all that fit.
    }
    long long blocks = (m + 255) / 256;
    // tf_set_launch_budget: fewer resident workgroups per CU than fit, so that another stream's kernel (the inner-light net on the
    // matrix cores, the flow sampler on the vector unit) finds registers / wave slots on every CU beside this latency-bound one
    const int budget = tf_launch_budget().bvh_blocks_per_cu;
    const int per_cu = budget > 0 && budget < resident ? budget : resident;
    if (blocks > 256LL * per_cu) blocks = 256LL * per_cu;   // persistent blocks pull rays until the pool is empty
#ifndef BVH_NO_SPINE
    const bool use_spine = rays_per_origin >= 64 && m % rays_per_origin == 0;
#else
    const bool use_spine = false;
#endif
    A.unit_parts = (int)((rays_per_origin + BVH_CHUNK_MAX - 1) / BVH_CHUNK_MAX);
    A.unit_size = (int)((rays_per_origin + A.unit_parts - 1) / A.unit_parts);
    A.spine_radius = (fabsf(origin_offset0) + fabsf(origin_offset1)) * 1.001f + 1e-6f;
    if (use_spine && BVH_STAGE && !slot_order) bvh_trace_kernel<true, true, true><<<(unsigned)blocks, 256, 0, stream>>>(A);
    else if (use_spine) bvh_trace_kernel<true, true><<<(unsigned)blocks, 256, 0, stream>>>(A);
    else bvh_trace_kernel<true, false><<<(unsigned)blocks, 256, 0, stream>>>(A);
  } else {
    A.unit_parts = 1; A.unit_size = 1; A.spine_radius = 0.f;
    bvh_trace_kernel<false, false><<<tf_blocks(m, 256), 256, 0, stream>>>(A);
  }
  TF_LAUNCH_CHECK("tf_bvh_trace");
  return TF_OK;
}
