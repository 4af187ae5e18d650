// Cube-map bilinear tap set shared by the lookup kernels (light.hip) and the fused shape-shading kernel (shape_shade.hip):
// dr.texture(..., filter_mode='linear', boundary_mode='cube') -- taps that leave the face are re-projected onto the
// neighbouring face, the tap that falls off a cube corner is dropped and the other three renormalised
// (same rule as oracle/texture.py:cube_bilinear).
#pragma once
#include <hip/hip_runtime.h>

struct CubeTaps {
  int idx[4];   // flat texel index (face*R + y)*R + x
  float w[4];
};

__device__ __forceinline__ void cube_face_uv(float dx, float dy, float dz, int& face, float& x, float& y) {
  const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
  if (az > fmaxf(ax, ay)) {
    const float m = 1.f / az;
    face = dz < 0.f ? 5 : 4;
    x = (dz < 0.f ? -dx : dx) * m;
    y = -dy * m;
  } else if (ay > ax) {
    const float m = 1.f / ay;
    face = dy < 0.f ? 3 : 2;
    x = dx * m;
    y = (dy < 0.f ? -dz : dz) * m;
  } else {
    const float m = 1.f / ax;
    face = dx < 0.f ? 1 : 0;
    x = (dx < 0.f ? dz : -dz) * m;
    y = -dy * m;
  }
}

__device__ __forceinline__ void cube_face_dir(int face, float x, float y, float& dx, float& dy, float& dz) {
  switch (face) {
    case 0: dx = 1.f; dy = -y; dz = -x; break;
    case 1: dx = -1.f; dy = -y; dz = x; break;
    case 2: dx = x; dy = 1.f; dz = y; break;
    case 3: dx = x; dy = -1.f; dz = -y; break;
    case 4: dx = x; dy = -y; dz = 1.f; break;
    default: dx = -x; dy = -y; dz = -1.f; break;
  }
}

// Where a tap that leaves face f over edge e (0: u = -1, 1: u = R, 2: v = -1, 3: v = R) lands: the re-projection of the tap centre
// ((i + .5) / R * 2 - 1 on the extended face plane -> direction -> face, floor of its uv) always ends in the FIRST row of texels of
// the neighbouring face behind that edge, at the tap's own along-edge index a or its mirror image -- for every R (the projection
// moves the along-edge coordinate by less than half a texel: |a + .5 - R/2| / (R + 1) < .5).  7 bits per (f, e): face' | ku << 3 |
// kv << 5 with k = 0: a, 1: R - 1 - a, 2: 0, 3: R - 1.  Derived by running the float path below over every border tap of
// R = 4 ... 512 (round 6); tests/test_gpu_parity.py::test_cube_lookup_golden hugs edges and corners against the oracle.
// The float path is a re-projection with an IEEE division per tap, run by the whole wave whenever one of its 64 rays has a tap at an edge
// (nearly always); outputs are bit-identical (tools/exp_cube_taps.py: sha1 of 2 M edge- and corner-hugging lookups at R = 4 ... 512).
// Measured gain: shape_shade_kernel 3.91 -> 3.82 ms per launch, shade_reduce_env_kernel none (it waits on its row and texel round trips).
#ifndef CUBE_TAPS_FLOAT
#define CUBE_TAPS_FLOAT 0      // 1 (dev switch): the re-projection in floating point, as rounds 1-5 ran it
#endif
__device__ __forceinline__ unsigned cube_edge_word(int face) {      // the four 7-bit entries of a face, edge e at bit 7 e
  unsigned w = 0x36e8a9cu;
  w = face == 1 ? 0x6648a1du : w; w = face == 2 ? 0x8936441u : w; w = face == 3 ? 0xdb93069u : w; w = face == 4 ? 0x8788819u : w; w = face == 5 ? 0xd728898u : w;
  return w;
}

__device__ __forceinline__ void cube_taps(float dx, float dy, float dz, int R, CubeTaps& T) {
  int face;
  float x, y;
  cube_face_uv(dx, dy, dz, face, x, y);
  const float u = (x * 0.5f + 0.5f) * (float)R - 0.5f;
  const float v = (y * 0.5f + 0.5f) * (float)R - 0.5f;
  const float fu0 = floorf(u), fv0 = floorf(v);
  const float fu = u - fu0, fv = v - fv0;
  const int iu0 = (int)fu0, iv0 = (int)fv0;
#if !CUBE_TAPS_FLOAT
  const unsigned eword = cube_edge_word(face);
#endif
  float wsum = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int du = t & 1, dv = t >> 1;
    const int iu = iu0 + du, iv = iv0 + dv;
    float w = (du ? fu : 1.f - fu) * (dv ? fv : 1.f - fv);
    const bool ou = iu < 0 || iu > R - 1, ov = iv < 0 || iv > R - 1;
    int f2 = face, ju = iu, jv = iv;
    if (ou || ov) {
#if CUBE_TAPS_FLOAT
      const float tx = ((float)iu + 0.5f) / (float)R * 2.f - 1.f;
      const float ty = ((float)iv + 0.5f) / (float)R * 2.f - 1.f;
      float ex, ey, ez, x2, y2;
      cube_face_dir(face, tx, ty, ex, ey, ez);
      cube_face_uv(ex, ey, ez, f2, x2, y2);
      ju = min(max((int)floorf((x2 * 0.5f + 0.5f) * (float)R), 0), R - 1);
      jv = min(max((int)floorf((y2 * 0.5f + 0.5f) * (float)R), 0), R - 1);
#else
      // (a tap off a CORNER has weight 0: any texel of the map will do for it -- the clamp keeps its index inside)
      const int edge = ou ? (iu < 0 ? 0 : 1) : (iv < 0 ? 2 : 3);
      const int a = min(max(ou ? iv : iu, 0), R - 1);
      const unsigned en = eword >> (7 * edge);
      const int ku = (en >> 3) & 3, kv = (en >> 5) & 3;
      f2 = en & 7;
      ju = ku == 0 ? a : ku == 1 ? R - 1 - a : ku == 2 ? 0 : R - 1;
      jv = kv == 0 ? a : kv == 1 ? R - 1 - a : kv == 2 ? 0 : R - 1;
#endif
      if (ou && ov) w = 0.f;
    }
    T.idx[t] = (f2 * R + jv) * R + ju;
    T.w[t] = w;
    wsum += w;
  }
  const float inv = 1.f / wsum;
#pragma unroll
  for (int t = 0; t < 4; ++t) T.w[t] *= inv;
}

// bilinear cube fetch of an RGB map [6,R,R,3]
__device__ __forceinline__ void cube_fetch_rgb(const float* __restrict__ base, int R, float dx, float dy, float dz, float& r,
                                               float& g, float& b) {
  CubeTaps T;
  cube_taps(dx, dy, dz, R, T);
  r = 0.f; g = 0.f; b = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float* p = base + 3LL * T.idx[t];
    r += T.w[t] * p[0]; g += T.w[t] * p[1]; b += T.w[t] * p[2];
  }
}
