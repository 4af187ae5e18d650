// Env-light prefilter: EnvLight.build_mips (network/light.py:52-64) = 2x2 box mips (network/light_utils.py:66-70),
// cosine ("diffuse") filter and GGX-lobe ("specular") filter of a [6,R,R,3] log-radiance cube map, plus the adjoints the
// shape stage needs every training step (network/shapeRenderer.py:1291).  Semantics follow
// network/renderutils/c_src/cubemap.cu (:17-46 texel area / direction, :112-140 diffuse, :170-176,239-287 GGX) --
// the CUDA code walks a per-face texel bounding box with one thread per output texel and scatters the adjoint with
// atomics; here
//   * one 64-lane wave owns one output texel; its lanes first cull 8x8 source tiles against the lobe cone (ballot), then
//     the 64 lanes take the 64 texels of each surviving tile, so the pair work is lane-parallel and the reduction is a
//     wave butterfly;
//   * the weight of a (V, L) pair depends on L only through dot products and area(L), so the adjoint is the SAME gather
//     with the roles of the fixed and the running texel swapped -- deterministic, no atomics.
// Pair arithmetic is written in the reference's operation order with FMA contraction off: the GGX term at roughness 0.08
// is ill-conditioned at the lobe centre (see oracle/cubemap.py).
#include <cstdint>
#include "tf_common.h"

#pragma clang fp contract(off)

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 texel_dir(int x, int y, int side, int N) {
  const float fx = 2.0f * (((float)x + 0.5f) / (float)N) - 1.0f;
  const float fy = 2.0f * (((float)y + 0.5f) / (float)N) - 1.0f;
  V3 v;
  switch (side) {
    case 0: v = {1.f, -fy, -fx}; break;
    case 1: v = {-1.f, -fy, fx}; break;
    case 2: v = {fx, 1.f, fy}; break;
    case 3: v = {fx, -1.f, -fy}; break;
    case 4: v = {fx, -fy, 1.f}; break;
    default: v = {-fx, -fy, -1.f}; break;
  }
  const float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  return {v.x / l, v.y / l, v.z / l};
}

__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// weight of the pair (V = filtered texel direction, L = source texel direction, area = solid-angle proxy of L's texel)
template <int MODE>
__device__ __forceinline__ float pair_weight(V3 V, V3 L, float area, float a2, float cutoff) {
  const float t = dot3(L, V);
  if (MODE == 0) {
    const float c = fminf(fmaxf(t, 0.f), 0.999f);
    return c * area / 3.141592f;
  }
  if (!(t >= cutoff)) return 0.f;
  V3 S = {L.x + V.x, L.y + V.y, L.z + V.z};
  const float l = sqrtf(S.x * S.x + S.y * S.y + S.z * S.z);
  V3 H = l > 0.f ? V3{S.x / l, S.y / l, S.z / l} : V3{0.f, 0.f, 0.f};
  const float wi = fmaxf(t, 0.f);
  const float c = fminf(fmaxf(dot3(V, H), 0.f), 1.f);
  const float d = (c * a2 - c) * c + 1.0f;
  const float D = (float)((double)a2 / ((double)(d * d) * 3.14159265358979323846));
  return wi * D * area / 4.0f;
}

// ADJ = 0: out[t] = sum_b w(V=t, L=b) src[b]            (MODE 1: / sum_b w, also written to wsum)
// ADJ = 1: out[t] = sum_a w(V=a, L=t) src[a] (/ wsum[a])
// TILES: the cone test of the 8 x 8 source tiles reads a per-workgroup table in LDS -- (centre direction, cos(theta_cut + angular
// radius + margin)) of every tile, built once by the workgroup's 256 threads -- instead of being re-derived by every wave for every
// output texel (four normalised corner directions, two acos: ~150 instructions per tile and texel, nine tenths of the kernel at
// R = 128 where a narrow lobe keeps 1-4 of 1 536 tiles).  The test stays conservative (margin 3e-3 rad), and a tile that passes
// needlessly only contributes exact zeros: results are unchanged bit for bit.
// TAB (round 5): the running texel's direction and area come from a table (tf_cubemap_texel_table: the same two expressions, evaluated
// once per resolution instead of once per pair -- two integer divisions, five float divisions, a square root and the face switch were
// ~100 of a pair's ~250 instructions); values, hence results, are bit-identical to the in-kernel form.
template <int MODE, int ADJ, bool TILES, bool TAB = false>
__global__ void __launch_bounds__(256) cube_filter_kernel(const float* __restrict__ src, const float* __restrict__ wsum_in, int R,
                                                          float a2, float cutoff, float theta_cut, float* __restrict__ out,
                                                          float* __restrict__ wsum_out, const float4* __restrict__ tab = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float s_ax[];   // per-axis angular extent, pixel_area(x,y) = s_ax[x] * s_ax[y]; then the tile table
  float4* s_tile = reinterpret_cast<float4*>(s_ax + ((R + 3) & ~3));
  if (TILES) {
    const int T = R >> 3, ntile = 6 * T * T;
    for (int tile = threadIdx.x; tile < ntile; tile += 256) {
      const int s = tile / (T * T), y0 = ((tile / T) % T) * 8, x0 = (tile % T) * 8;
      V3 c0 = texel_dir(x0, y0, s, R), c1 = texel_dir(x0 + 7, y0, s, R), c2 = texel_dir(x0, y0 + 7, s, R), c3 = texel_dir(x0 + 7, y0 + 7, s, R);
      V3 c = {c0.x + c1.x + c2.x + c3.x, c0.y + c1.y + c2.y + c3.y, c0.z + c1.z + c2.z + c3.z};
      const float il = 1.f / sqrtf(dot3(c, c));
      c = {c.x * il, c.y * il, c.z * il};
      const float mind = fminf(fminf(dot3(c, c0), dot3(c, c1)), fminf(dot3(c, c2), dot3(c, c3)));
      const float rho = acosf(fminf(fmaxf(mind, -1.f), 1.f));
      const float lim = theta_cut + rho + 3e-3f;
      s_tile[tile] = make_float4(c.x, c.y, c.z, lim >= 3.14159f ? -2.f : cosf(lim) - 1e-6f);   // pass iff dot(centre, F) >= w
    }
  }
  for (int i = threadIdx.x; i < R; i += 256) {
    if (R > 1) {
      const int H = R / 2;
      const int a = abs(i - H);
      s_ax[i] = atanf((float)(a + 1) / (float)H) - atanf((float)a / (float)H);
    } else {
      s_ax[i] = 1.f;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int n = 6 * R * R;
  // a wave owns one output texel at a time and walks the map in strides of the grid: the tables above are built once per workgroup,
  // not once per four texels (round 5: at 128^2 they were more than half of the kernel)
  for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < n; t += gridDim.x * 4) {
  const int ts = t / (R * R), ty = (t / R) % R, tx = t % R;
  const V3 F = texel_dir(tx, ty, ts, R);
  const float areaF = s_ax[tx] * s_ax[ty];
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, ws = 0.f;

  auto pair = [&](int b) {
    V3 G;
    float areaG;
    if (TAB) {
      const float4 q = tab[b];
      G = {q.x, q.y, q.z};
      areaG = q.w;
    } else {
      const int bs = b / (R * R), by = (b / R) % R, bx = b % R;
      G = texel_dir(bx, by, bs, R);
      areaG = s_ax[bx] * s_ax[by];
    }
    float w;
    if (ADJ == 0) w = pair_weight<MODE>(F, G, areaG, a2, cutoff);
    else          w = pair_weight<MODE>(G, F, areaF, a2, cutoff);
    if (w != 0.f) {
      float s0 = src[3 * b], s1 = src[3 * b + 1], s2 = src[3 * b + 2];
      if (ADJ == 1 && MODE == 1) { const float q = wsum_in[b]; s0 = s0 / q; s1 = s1 / q; s2 = s2 / q; }
      acc0 += s0 * w; acc1 += s1 * w; acc2 += s2 * w; ws += w;
    }
  };

  if (MODE == 0 || (R & 7)) {
    for (int b = lane; b < n; b += 64) pair(b);
  } else {
    const int T = R >> 3, ntile = 6 * T * T;
    for (int base = 0; base < ntile; base += 64) {
      const int tile = base + lane;
      bool pass = false;
      if (TILES) {
        if (tile < ntile) { const float4 tc = s_tile[tile]; pass = tc.x * F.x + tc.y * F.y + tc.z * F.z >= tc.w; }
      } else if (tile < ntile) {
        const int s = tile / (T * T), y0 = ((tile / T) % T) * 8, x0 = (tile % T) * 8;
        // centre of the tile's texel-centre hull and its angular radius (attained at a corner texel)
        V3 c0 = texel_dir(x0, y0, s, R), c1 = texel_dir(x0 + 7, y0, s, R), c2 = texel_dir(x0, y0 + 7, s, R),
           c3 = texel_dir(x0 + 7, y0 + 7, s, R);
        V3 c = {c0.x + c1.x + c2.x + c3.x, c0.y + c1.y + c2.y + c3.y, c0.z + c1.z + c2.z + c3.z};
        const float il = 1.f / sqrtf(dot3(c, c));
        c = {c.x * il, c.y * il, c.z * il};
        const float mind = fminf(fminf(dot3(c, c0), dot3(c, c1)), fminf(dot3(c, c2), dot3(c, c3)));
        const float rho = acosf(fminf(fmaxf(mind, -1.f), 1.f));
        const float phi = acosf(fminf(fmaxf(dot3(c, F), -1.f), 1.f));
        pass = phi <= theta_cut + rho + 2e-3f;
      }
      unsigned long long mask = __ballot(pass);
      while (mask) {
        const int tl = base + __builtin_ctzll(mask);
        mask &= mask - 1;
        const int s = tl / (T * T), y0 = ((tl / T) % T) * 8, x0 = (tl % T) * 8;
        pair((s * R + y0 + (lane >> 3)) * R + x0 + (lane & 7));
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    acc0 += __shfl_xor(acc0, o); acc1 += __shfl_xor(acc1, o); acc2 += __shfl_xor(acc2, o); ws += __shfl_xor(ws, o);
  }
  if (lane == 0) {
    if (MODE == 1 && ADJ == 0) {
      out[3 * t] = acc0 / ws; out[3 * t + 1] = acc1 / ws; out[3 * t + 2] = acc2 / ws;
      if (wsum_out) wsum_out[t] = ws;
    } else {
      out[3 * t] = acc0; out[3 * t + 1] = acc1; out[3 * t + 2] = acc2;
    }
  }
  }
}

// table of (texel direction, texel area) of a [6,R,R] map, as cube_filter_kernel derives them
__global__ void __launch_bounds__(256) cube_texel_table_kernel(int R, float4* __restrict__ tab) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= 6 * R * R) return;
  const int bs = b / (R * R), by = (b / R) % R, bx = b % R;
  auto ax = [&](int i) {
    if (R <= 1) return 1.f;
    const int H = R / 2;
    const int a = abs(i - H);
    return atanf((float)(a + 1) / (float)H) - atanf((float)a / (float)H);
  };
  const V3 G = texel_dir(bx, by, bs, R);
  tab[b] = make_float4(G.x, G.y, G.z, ax(bx) * ax(by));
}
extern "C" int tf_cubemap_texel_table(int32_t res, float* table, tf_stream_t stream) {
  TF_REQUIRE(res >= 1 && res <= 2048, TF_ESHAPE, "tf_cubemap_texel_table: res=%d out of range", res);
  TF_REQUIRE(table && (reinterpret_cast<uintptr_t>(table) & 15) == 0, TF_EINVAL, "tf_cubemap_texel_table: null / unaligned table");
  cube_texel_table_kernel<<<tf_blocks(6LL * res * res, 256), 256, 0, (hipStream_t)stream>>>(res, reinterpret_cast<float4*>(table));
  TF_LAUNCH_CHECK("tf_cubemap_texel_table");
  return TF_OK;
}

__global__ void __launch_bounds__(256) cube_mip_kernel(const float* __restrict__ src, int R, float* __restrict__ out) {
  const int Ro = R >> 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 6 * Ro * Ro * 3) return;
  const int c = i % 3, x = (i / 3) % Ro, y = (i / (3 * Ro)) % Ro, s = i / (3 * Ro * Ro);
  const float* p = src + ((long long)(s * R + 2 * y) * R + 2 * x) * 3 + c;
  out[i] = (((p[0] + p[3]) + p[3 * R]) + p[3 * R + 3]) * 0.25f;
}

extern "C" int tf_cubemap_mip_fwd(const float* cube, int32_t res, float* out, tf_stream_t stream) {
  TF_REQUIRE(res >= 2 && res % 2 == 0, TF_ESHAPE, "tf_cubemap_mip_fwd: res=%d must be even and >= 2", res);
  TF_REQUIRE(cube && out, TF_EINVAL, "tf_cubemap_mip_fwd: null pointer");
  cube_mip_kernel<<<tf_blocks(6LL * (res / 2) * (res / 2) * 3, 256), 256, 0, (hipStream_t)stream>>>(cube, res, out);
  TF_LAUNCH_CHECK("tf_cubemap_mip_fwd");
  return TF_OK;
}

// Four output texels per workgroup and turn.  A workgroup takes several turns only where its tables are worth amortising: a narrow lobe
// (128^2 at roughness 0.08: 1-4 of 1 536 tiles survive, the tables are twice a texel's own work) -- a wide lobe's texels cost tens of
// times the tables, and one turn per workgroup leaves their uneven windows to the hardware's dispatcher (a fixed stride measured
// 12-25 % slower there).
static unsigned filter_blocks(int res, float theta_cut, bool tiles) {
  const unsigned want = tf_blocks(6LL * res * res, 4);
  if (!tiles) return want;
  const int T = res >> 3, ntile = 6 * T * T, nround = (ntile + 63) / 64;
  const float frac = 0.5f * (1.f - cosf(fminf(theta_cut + 0.1f, 3.14159f)));            // the lobe's share of the sphere (+ a tile's radius)
  const float per_texel = 15.f * nround + 150.f * fmaxf(1.f, frac * ntile);               // ~instructions of a wave per output texel
  const float tables = 150.f * ntile / 256.f;                                             // ~instructions per thread, once per workgroup
  int turns = (int)ceilf(8.f * tables / per_texel);    // (measured at 128^2 / 0.08: 4 turns 0.276 ms, 9 turns 0.235 ms, one turn 0.530 ms)
  turns = turns < 1 ? 1 : (turns > 16 ? 16 : turns);
  return (want + turns - 1) / turns;
}

template <int MODE, int ADJ>
static int launch_filter(const char* name, const float* src, const float* wsum_in, int res, float roughness, float cos_cutoff,
                         float* out, float* wsum_out, tf_stream_t stream, const float* table = nullptr) {
  TF_REQUIRE(res >= 1 && res <= 2048, TF_ESHAPE, "%s: res=%d out of range", name, res);
  TF_REQUIRE(src && out, TF_EINVAL, "%s: null pointer", name);
  const float alpha = roughness * roughness;
  const float a2 = alpha * alpha;
  const float cc = cos_cutoff < -1.f ? -1.f : (cos_cutoff > 1.f ? 1.f : cos_cutoff);
  const int T = res >> 3;
  const size_t tile_bytes = (size_t)6 * T * T * sizeof(float4);
  TF_REQUIRE(!table || (reinterpret_cast<uintptr_t>(table) & 15) == 0, TF_EINVAL, "%s: unaligned texel table", name);
  if (MODE == 1 && (res & 7) == 0 && tile_bytes <= 48 * 1024 && table)
    cube_filter_kernel<MODE, ADJ, true, true><<<filter_blocks(res, acosf(cc), true), 256, ((res + 3) & ~3) * sizeof(float) + tile_bytes, (hipStream_t)stream>>>(
        src, wsum_in, res, a2, cos_cutoff, acosf(cc), out, wsum_out, reinterpret_cast<const float4*>(table));
  else if (MODE == 1 && (res & 7) == 0 && tile_bytes <= 48 * 1024)
    cube_filter_kernel<MODE, ADJ, true><<<filter_blocks(res, acosf(cc), true), 256, ((res + 3) & ~3) * sizeof(float) + tile_bytes, (hipStream_t)stream>>>(
        src, wsum_in, res, a2, cos_cutoff, acosf(cc), out, wsum_out);
  else
    cube_filter_kernel<MODE, ADJ, false><<<filter_blocks(res, acosf(cc), false), 256, ((res + 3) & ~3) * sizeof(float) + 16, (hipStream_t)stream>>>(
        src, wsum_in, res, a2, cos_cutoff, acosf(cc), out, wsum_out);
  TF_LAUNCH_CHECK(name);
  return TF_OK;
}

extern "C" int tf_cubemap_diffuse_fwd(const float* cube, int32_t res, float* out, tf_stream_t stream) {
  return launch_filter<0, 0>("tf_cubemap_diffuse_fwd", cube, nullptr, res, 0.f, 0.f, out, nullptr, stream);
}
extern "C" int tf_cubemap_diffuse_bwd(const float* g_out, int32_t res, float* g_cube, tf_stream_t stream) {
  return launch_filter<0, 1>("tf_cubemap_diffuse_bwd", g_out, nullptr, res, 0.f, 0.f, g_cube, nullptr, stream);
}
extern "C" int tf_cubemap_specular_fwd(const float* cube, int32_t res, float roughness, float cos_cutoff, float* out, float* wsum,
                                       const float* texel_table, tf_stream_t stream) {
  return launch_filter<1, 0>("tf_cubemap_specular_fwd", cube, nullptr, res, roughness, cos_cutoff, out, wsum, stream, texel_table);
}
extern "C" int tf_cubemap_specular_bwd(const float* g_out, const float* wsum, int32_t res, float roughness, float cos_cutoff,
                                       float* g_cube, const float* texel_table, tf_stream_t stream) {
  TF_REQUIRE(wsum, TF_EINVAL, "tf_cubemap_specular_bwd: null wsum");
  return launch_filter<1, 1>("tf_cubemap_specular_bwd", g_out, wsum, res, roughness, cos_cutoff, g_cube, nullptr, stream, texel_table);
}
