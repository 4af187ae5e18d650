/* TEST INFRASTRUCTURE (oracle): first-hit ray / triangle-mesh intersection on the CPU with a plain binary BVH.
 *
 * Same contract as oracle/mesh.py:ray_triangles (the brute-force restatement of the third-party `raytracing` package the
 * reference calls at network/materialRenderer.py:253-263, wrapper raytracing/raytracer.py:19-54): nearest t over all triangles
 * with the Moeller-Trumbore / iq acceptance window u >= 0, u <= 1, v >= 0, u + v <= 1, t >= 0; t = 10.0 and face = -1 on a miss.
 * The per-triangle arithmetic is the brute-force oracle's, operation for operation, in fp32 without fused multiply-adds
 * (build with -ffp-contract=off): a ray's t agrees with the brute-force value to an ulp or two (torch rounds its 3-term
 * reductions differently on ~1 % of rays) and the hit sets are identical; the BVH only decides which triangles are looked at (boxes are tested with a relative + absolute slack, never culling a triangle the exact test accepts).
 * Ties between coincident hits may pick a different face than the brute-force argmin (unpinned in the reference as well).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this library (through oracle/mesh.py).
 *
 *   gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC oracle/bvh_cpu.c -o oracle/_bvh_cpu.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_DIST 10.0f
#define LEAF 4

typedef struct { float lo[3], hi[3]; int32_t left, count; } Node;   /* count > 0: leaf over tris [left, left + count) */
typedef struct {
  Node* nodes; int64_t n_nodes;
  float* tri;      /* [nf][9] reordered a, b, c */
  int32_t* face;   /* reordered -> original face index */
  int64_t nf;
} Bvh;

static void tri_box(const float* t, float* lo, float* hi) {
  for (int k = 0; k < 3; ++k) {
    float a = t[k], b = t[3 + k], c = t[6 + k];
    lo[k] = fminf(a, fminf(b, c)); hi[k] = fmaxf(a, fmaxf(b, c));
  }
}

typedef struct { const float* cen; int axis; } SortCtx;
static SortCtx g_ctx;   /* build is single-threaded */
static int cmp_axis(const void* x, const void* y) {
  float a = g_ctx.cen[3 * (*(const int32_t*)x) + g_ctx.axis], b = g_ctx.cen[3 * (*(const int32_t*)y) + g_ctx.axis];
  return (a > b) - (a < b);
}

static void build(Bvh* B, const float* tri_in, const float* cen, int32_t* order, int64_t node, int64_t begin, int64_t end) {
  Node* nd = &B->nodes[node];
  float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = 0; k < 3; ++k) { nd->lo[k] = INFINITY; nd->hi[k] = -INFINITY; }
  for (int64_t i = begin; i < end; ++i) {
    float lo[3], hi[3];
    tri_box(tri_in + 9 * (int64_t)order[i], lo, hi);
    for (int k = 0; k < 3; ++k) {
      nd->lo[k] = fminf(nd->lo[k], lo[k]); nd->hi[k] = fmaxf(nd->hi[k], hi[k]);
      float c = cen[3 * (int64_t)order[i] + k];
      clo[k] = fminf(clo[k], c); chi[k] = fmaxf(chi[k], c);
    }
  }
  if (end - begin <= LEAF) { nd->left = (int32_t)begin; nd->count = (int32_t)(end - begin); return; }
  int ax = 0;
  if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
  if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
  g_ctx.cen = cen; g_ctx.axis = ax;
  qsort(order + begin, (size_t)(end - begin), sizeof(int32_t), cmp_axis);   /* median split on the widest centroid axis */
  int64_t mid = begin + (end - begin) / 2;
  int64_t left = B->n_nodes;
  B->n_nodes += 2;
  nd->left = (int32_t)left; nd->count = 0;
  build(B, tri_in, cen, order, left, begin, mid);
  build(B, tri_in, cen, order, left + 1, mid, end);
}

/* verts [nv][3], faces [nf][3] -> opaque handle */
void* obvh_build(const float* verts, const int32_t* faces, int64_t nf) {
  Bvh* B = (Bvh*)calloc(1, sizeof(Bvh));
  B->nf = nf;
  float* tri_in = (float*)malloc(sizeof(float) * 9 * nf);
  float* cen = (float*)malloc(sizeof(float) * 3 * nf);
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * nf);
  for (int64_t f = 0; f < nf; ++f) {
    order[f] = (int32_t)f;
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a) tri_in[9 * f + 3 * k + a] = verts[3 * (int64_t)faces[3 * f + k] + a];
    for (int a = 0; a < 3; ++a) cen[3 * f + a] = (tri_in[9 * f + a] + tri_in[9 * f + 3 + a] + tri_in[9 * f + 6 + a]) / 3.f;
  }
  B->nodes = (Node*)malloc(sizeof(Node) * (2 * nf + 2));
  B->n_nodes = 1;
  build(B, tri_in, cen, order, 0, 0, nf);
  B->tri = (float*)malloc(sizeof(float) * 9 * nf);
  B->face = (int32_t*)malloc(sizeof(int32_t) * nf);
  for (int64_t i = 0; i < nf; ++i) {
    memcpy(B->tri + 9 * i, tri_in + 9 * (int64_t)order[i], sizeof(float) * 9);
    B->face[i] = order[i];
  }
  free(tri_in); free(cen); free(order);
  return B;
}

void obvh_free(void* h) {
  Bvh* B = (Bvh*)h;
  if (!B) return;
  free(B->nodes); free(B->tri); free(B->face); free(B);
}

/* conservative slab test: [t0, t1] overlap with [0, tmax], widened by a relative and an absolute slack */
static int box_hit(const Node* n, const float* o, const float* inv, float tmax) {
  float t0 = 0.f, t1 = tmax;
  for (int k = 0; k < 3; ++k) {
    float a = (n->lo[k] - o[k]) * inv[k], b = (n->hi[k] - o[k]) * inv[k];
    if (a != a || b != b) continue;                       /* 0 * inf: the ray runs inside this slab's plane */
    float lo = fminf(a, b), hi = fmaxf(a, b);
    lo = lo - fabsf(lo) * 1e-5f - 1e-5f; hi = hi + fabsf(hi) * 1e-5f + 1e-5f;
    t0 = fmaxf(t0, lo); t1 = fminf(t1, hi);
  }
  return t0 <= t1;
}

/* the brute-force oracle's per-triangle arithmetic (oracle/mesh.py:ray_triangles), fp32, no contraction */
static inline float tri_t(const float* T, const float* o, const float* d) {
  const float ax = T[0], ay = T[1], az = T[2];
  const float v1x = T[3] - ax, v1y = T[4] - ay, v1z = T[5] - az;
  const float v2x = T[6] - ax, v2y = T[7] - ay, v2z = T[8] - az;
  const float nx = v1y * v2z - v1z * v2y, ny = v1z * v2x - v1x * v2z, nz = v1x * v2y - v1y * v2x;
  const float rx = o[0] - ax, ry = o[1] - ay, rz = o[2] - az;
  const float qx = ry * d[2] - rz * d[1], qy = rz * d[0] - rx * d[2], qz = rx * d[1] - ry * d[0];
  const float det = 1.0f / ((d[0] * nx + d[1] * ny) + d[2] * nz);
  const float u = det * -((qx * v2x + qy * v2y) + qz * v2z);
  const float v = det * ((qx * v1x + qy * v1y) + qz * v1z);
  const float t = det * -((nx * rx + ny * ry) + nz * rz);
  if (u < 0.f || u > 1.f || v < 0.f || u + v > 1.f || t < 0.f || t != t) return 1e6f;
  return t;
}

/* o, d [m][3] -> t [m] (10.0 on a miss), face [m] (-1 on a miss; original face index) */
void obvh_trace(const void* h, const float* o, const float* d, int64_t m, float* t_out, int32_t* face_out) {
  const Bvh* B = (const Bvh*)h;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < m; ++i) {
    const float* oo = o + 3 * i; const float* dd = d + 3 * i;
    const float inv[3] = {1.f / dd[0], 1.f / dd[1], 1.f / dd[2]};
    float best = MAX_DIST; int32_t bf = -1;
    int32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
      const Node* n = &B->nodes[stack[--sp]];
      if (!box_hit(n, oo, inv, best)) continue;
      if (n->count > 0) {
        for (int k = 0; k < n->count; ++k) {
          const float t = tri_t(B->tri + 9 * (int64_t)(n->left + k), oo, dd);
          if (t < best) { best = t; bf = B->face[n->left + k]; }
        }
      } else if (sp + 2 <= 128) {
        stack[sp++] = n->left; stack[sp++] = n->left + 1;
      }
    }
    t_out[i] = best; face_out[i] = bf;
  }
}
