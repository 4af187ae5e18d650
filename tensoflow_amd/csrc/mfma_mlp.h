// fp32 MFMA building blocks for the tiny MLPs on the hot path (gfx950, wave64).
//
// All decoders on the path (flow coupling nets 44-64-64-64-21, inner-light net 123-256-256-256-3,
// SDF decoder 111-256-129) are evaluated TRANSPOSED: activations live as H^T tiles of
// [32 units x 32 rows], rows (samples) on the MFMA column/lane index.  With
// v_mfma_f32_32x32x2_f32 (exact fp32, D = A*B + C):
//     lane l supplies  A[i = l&31][k = l>>5]   and   B[k = l>>5][j = l&31]
//     lane l receives  D[i = rho(reg, l>>5)][j = l&31],  rho(reg,h) = (reg&3) + 8*(reg>>2) + 4*h
// so an accumulator register `reg` of unit-tile t IS the B operand of the next layer's k-step
// (t, reg) -- k = 32*t + rho(reg, h) for lane half h -- with no cross-lane movement and no LDS
// round trip between layers.  Weights are pre-packed in that fragment order ("wfrag"):
//     wfrag[((tout * ksteps) + s) * 64 + lane] = W[32*tout + (lane&31)][kmap(s, lane>>5)],
//     s = 16*t + reg,  kmap(s,h) = 32*(s>>4) + rho(s&15, h)
// which makes every A-operand fetch one contiguous 256-byte wave access (global or LDS).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int tf_rho(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
__host__ __device__ __forceinline__ int tf_kmap(int s, int h) {
  int reg = s & 15;
  return 32 * (s >> 4) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
}

__device__ __forceinline__ f32x16 tf_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- sin / cos without the libm call (which inlines its large-argument path at every call site).
// tf_sincos: argument reduction by pi/2 in double precision (exact to 1e-10 for |x| < 1e6), Cephes single-precision
// kernels on [-pi/4, pi/4]: max abs error 9.3e-8 for |x| <= 1e7 (checked against double precision, 4e7 random arguments).
__device__ __forceinline__ void tf_sincos_poly(float r, int q, float& s, float& c) {
  const float z = r * r;
  const float sp = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float cp = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                        fmaf(z, -0.5f, 1.0f));
  const float a = (q & 1) ? cp : sp, b = (q & 1) ? sp : cp;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}
__device__ __forceinline__ void tf_sincos(float x, float& s, float& c) {
  const double xd = (double)x;
  const double kd = rint(xd * 0.63661977236758134);
  const float r = (float)fma(kd, -1.5707963267948966, xd);
  tf_sincos_poly(r, (int)(long long)kd & 3, s, c);
}
// |x| < 200: three-constant Cody-Waite reduction in fp32 (k * A and k * B are exact for |k| < 2^8), max abs error 1.2e-7.
__device__ __forceinline__ void tf_sincos_small(float x, float& s, float& c) {
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(k, -1.57073974609375f, x);
  r = fmaf(k, -5.657970905303955e-05f, r);
  r = fmaf(k, -9.920936294705029e-10f, r);
  tf_sincos_poly(r, (int)k & 3, s, c);
}

// Pack W [nout, ld] (columns col0 .. col0+kin-1 used) into fragment order; zero padded.
static __global__ void __launch_bounds__(256) tf_pack_wfrag_kernel(const float* __restrict__ W, int nout, int ld, int col0,
                                                            int kin, int tout_tiles, int ksteps,
                                                            float* __restrict__ dst, int s_major = 0,
                                                            int transpose = 0) {
  // transpose = 1 packs W^T: logical row r / column k read W[k*ld + col0 + r]  (nout = logical rows, kin = logical cols)
  int e = blockIdx.x * 256 + threadIdx.x;
  int total = tout_tiles * ksteps * 64;
  if (e >= total) return;
  int lane = e & 63;
  int s, tout;
  if (s_major) {  // [s][tout][lane]: one k-group of all unit tiles is contiguous (LDS-staged streaming)
    tout = (e >> 6) % tout_tiles;
    s = (e >> 6) / tout_tiles;
  } else {        // [tout][s][lane]
    s = (e >> 6) % ksteps;
    tout = (e >> 6) / ksteps;
  }
  int row = 32 * tout + (lane & 31);
  int k = tf_kmap(s, lane >> 5);
  if (transpose) dst[e] = (row < nout && k < kin) ? W[(long long)k * ld + col0 + row] : 0.f;
  else dst[e] = (row < nout && k < kin) ? W[(long long)row * ld + col0 + k] : 0.f;
}

// Pack a bias vector [n] into accumulator order: dst[(tout*16 + reg)*2 + h] = b[32*tout + rho(reg,h)].
static __global__ void __launch_bounds__(256) tf_pack_bias_kernel(const float* __restrict__ b, int n, int tout_tiles,
                                                           float* __restrict__ dst) {
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= tout_tiles * 32) return;
  int h = e & 1, reg = (e >> 1) & 15, tout = e >> 5;
  int row = 32 * tout + (reg & 3) + 8 * (reg >> 2) + 4 * h;
  dst[e] = row < n ? b[row] : 0.f;
}

// One dense layer: out[TOUT] (+)= Wfrag * in[...]; KSTEPS k-steps taken from in[s>>4][s&15].
// `wf` points at this layer's fragment block (LDS or global), already offset by `lane`.
template <int KSTEPS, int TOUT, int TIN>
__device__ __forceinline__ void tf_layer(const float* __restrict__ wf, const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const float b = in[s >> 4][s & 15];
#pragma unroll
    for (int t = 0; t < TOUT; ++t) out[t] = tf_mfma(wf[(t * KSTEPS + s) * 64], b, out[t]);
  }
}

// Same, with a scheduling barrier every SB k-steps: bounds how far ahead the compiler hoists the
// (global-memory) A-operand loads, i.e. the registers they pin.
template <int KSTEPS, int TOUT, int TIN, int SB>
__device__ __forceinline__ void tf_layer_sb(const float* __restrict__ wf, const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
#pragma unroll
  for (int s0 = 0; s0 < KSTEPS; s0 += SB) {
#pragma unroll
    for (int s = s0; s < s0 + SB && s < KSTEPS; ++s) {
      const float b = in[s >> 4][s & 15];
#pragma unroll
      for (int t = 0; t < TOUT; ++t) out[t] = tf_mfma(wf[(t * KSTEPS + s) * 64], b, out[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same product for a wave that runs ALONE on its SIMD (nothing else covers a fragment fetch): the A operands of the NEXT group
// of G k-steps are requested before the matrix steps of the current one (two register buffers of G * TOUT values), and the first
// group can be requested by the caller long before the product starts (tf_layer_pf_first: e.g. ahead of a workgroup barrier).
// tf_layer_sb exposes one full L2 (or LDS) round trip per group: 1.01 -> 0.76 ms for tf_flow_logq_bwd's 262 k rows.
// WP: `const float*` in any address space, already offset by the lane.
template <int KSTEPS, int TOUT, int G, typename WP>
__device__ __forceinline__ void tf_layer_pf_first(WP wf, float (&a0)[G * TOUT]) {
#pragma unroll
  for (int q = 0; q < G && q < KSTEPS; ++q)
#pragma unroll
    for (int t = 0; t < TOUT; ++t) a0[q * TOUT + t] = wf[(t * KSTEPS + q) * 64];
}
template <int KSTEPS, int TOUT, int TIN, int G, typename WP>
__device__ __forceinline__ void tf_layer_pf_rest(WP wf, const float (&a0)[G * TOUT], const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
  constexpr int NG = (KSTEPS + G - 1) / G;
  float a[2][G * TOUT];
#pragma unroll
  for (int q = 0; q < G * TOUT; ++q) a[0][q] = a0[q];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int q = 0; q < G && (g + 1) * G + q < KSTEPS; ++q)
#pragma unroll
        for (int t = 0; t < TOUT; ++t) a[(g + 1) & 1][q * TOUT + t] = wf[(t * KSTEPS + (g + 1) * G + q) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < G && g * G + q < KSTEPS; ++q) {
      const int s = g * G + q;
      const float b = in[s >> 4][s & 15];
#pragma unroll
      for (int t = 0; t < TOUT; ++t) out[t] = tf_mfma(a[g & 1][q * TOUT + t], b, out[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int KSTEPS, int TOUT, int TIN, int G, typename WP>
__device__ __forceinline__ void tf_layer_pf(WP wf, const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
  float a0[G * TOUT];
  tf_layer_pf_first<KSTEPS, TOUT, G>(wf, a0);
  tf_layer_pf_rest<KSTEPS, TOUT, TIN, G>(wf, a0, in, out);
}

// Dense layer with the weights STREAMED through LDS by the whole 256-thread workgroup:
// wslab is s-major fragment order [KSTEPS][TOUT][64]; it is consumed in groups of SL k-steps
// (SL*TOUT*64 == 4096 floats = 16 KB), double buffered in `lds` (2 x 4096 floats).  Each thread moves
// 4 float4 per group global->register while the previous group's MFMAs run, then register->LDS.
// All four waves must call this with the same trip counts (it contains workgroup barriers).
typedef const __attribute__((address_space(1))) void* tf_gptr_t;
typedef __attribute__((address_space(3))) void* tf_lptr_t;

// One 16 KB weight slab global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, asynchronous):
// each of the 4 waves moves 4 pieces of 1 KB (lane l supplies the source address of its 16 bytes).
// The DMA is issued from an asm statement, not through __builtin_amdgcn_global_load_lds: hipcc's wait-count pass treats every
// outstanding LDS-DMA as a possible alias of every later LDS read and puts `s_waitcnt vmcnt(0)` in front of each ds_read
// group -- i.e. every slab step drained the whole ring (the slab requested a few hundred cycles earlier included) and the
// hand-counted vmcnt(4) never got to act.  Hidden from the compiler, the only waits on the ring are the counted ones placed
// by this file (tf_wait_vmcnt_barrier / tf_h3s_step / tf_stream_begin / tf_stream_end); its own vmcnt bookkeeping for other
// loads stays safe (unknown extra operations in flight can only make a counted wait longer, never shorter).
// M0 = LDS byte address of the piece (wave-uniform); lane l writes its 16 bytes at M0 + 16 l.
template <int BYTE_OFFSET = 0>
__device__ __forceinline__ void tf_dma16(const float* gsrc, float* ldst) {
#ifndef TF_ABLATE_DMA   // dev-only timing ablation (tools/build_variant.sh): results are garbage when defined
#ifdef TF_DMA_BUILTIN
  __builtin_amdgcn_global_load_lds((tf_gptr_t)(gsrc + BYTE_OFFSET / 4), (tf_lptr_t)(ldst + BYTE_OFFSET / 4), 16, 0, 0);
#else
  // BYTE_OFFSET goes into the instruction's immediate offset field (one 64-bit address per slab instead of one per piece).
  // The hardware adds that offset to the GLOBAL address and to the LDS address (M0 + offset + 16 * lane): `ldst` is therefore
  // the LDS address of the wave's FIRST piece for all four.
  const unsigned laddr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(tf_lptr_t)ldst);
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off offset:%2" ::"s"(laddr), "v"(gsrc), "n"(BYTE_OFFSET) : "memory");
#endif
#endif
}

__device__ __forceinline__ void tf_slab_dma(const float* gthread /* slab + wave*1024 + lane*4 */, float* __restrict__ lbuf,
                                            int wave) {
  tf_dma16<0>(gthread, lbuf + wave * 1024);
  tf_dma16<1024>(gthread, lbuf + wave * 1024);
  tf_dma16<2048>(gthread, lbuf + wave * 1024);
  tf_dma16<3072>(gthread, lbuf + wave * 1024);
}

__device__ __forceinline__ void tf_slab_dma_piece(const float* gthread, float* __restrict__ lbuf, int wave, int i) {
  // `i` is a compile-time constant at every (unrolled) call site
  if (i == 0) tf_dma16<0>(gthread, lbuf + wave * 1024);
  else if (i == 1) tf_dma16<1024>(gthread, lbuf + wave * 1024);
  else if (i == 2) tf_dma16<2048>(gthread, lbuf + wave * 1024);
  else tf_dma16<3072>(gthread, lbuf + wave * 1024);
}

template <int N>
__device__ __forceinline__ void tf_wait_vmcnt_barrier() {
  // counted wait (the N youngest vector-memory ops may stay in flight) + raw barrier: a __syncthreads() would make
  // the compiler drain vmcnt(0), i.e. also wait for the slab that was only just requested.
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// NBUF = 2: slab g+1 streams while slab g computes (32 KB LDS).  NBUF = 3: slabs g+1 and g+2 are in flight
// (48 KB LDS) -- one slab of MFMAs (~2 us) does not always cover an L2 round trip under load.
template <int KSTEPS, int TOUT, int TIN, int SL, int NBUF = 2>
__device__ __forceinline__ void tf_layer_stream(const float* __restrict__ wslab, float* __restrict__ lds, int tid, int lane,
                                                const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
  static_assert(SL * TOUT * 64 == 4096, "one group must be 16 KB");
  static_assert(KSTEPS % SL == 0, "KSTEPS must be a multiple of the group size");
  static_assert(NBUF == 2 || NBUF == 3, "double or triple buffering");
  constexpr int G = KSTEPS / SL;
  constexpr int AHEAD = NBUF - 1;
  const int wave = tid >> 6;
  // ONE running per-thread source pointer, advanced by 16 KB per slab and made opaque each step: otherwise the
  // compiler materialises every slab address as its own loop-invariant 64-bit VGPR pair, hoists them out of the
  // tile loop and spills them (each reload then serialises the DMAs behind an s_waitcnt vmcnt(0)).
  const float* gp = wslab + wave * 1024 + lane * 4;
  asm volatile("" : "+v"(gp));
  tf_slab_dma(gp, lds, wave);
  if (AHEAD == 2 && G > 1) {
    gp += 4096;
    asm volatile("" : "+v"(gp));
    tf_slab_dma(gp, lds + 4096, wave);
    tf_wait_vmcnt_barrier<4>();
  } else {
    tf_wait_vmcnt_barrier<0>();
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g + AHEAD < G) {
      gp += 4096;
      asm volatile("" : "+v"(gp));
      tf_slab_dma(gp, lds + ((g + AHEAD) % NBUF) * 4096, wave);
    }
    const float* buf = lds + (g % NBUF) * 4096 + lane;
    // A operands are read from LDS one k-step AHEAD of the MFMAs that consume them (register double buffer),
    // so the LDS latency hides under the previous step's MFMAs instead of stalling every issue.
    float a_cur[TOUT], a_nxt[TOUT];
#pragma unroll
    for (int t = 0; t < TOUT; ++t) a_cur[t] = buf[t * 64];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int s = g * SL + sl;
      const float b = in[s >> 4][s & 15];
      if (sl + 1 < SL) {
#pragma unroll
        for (int t = 0; t < TOUT; ++t) a_nxt[t] = buf[((sl + 1) * TOUT + t) * 64];
      }
#pragma unroll
      for (int t = 0; t < TOUT; ++t) out[t] = tf_mfma(a_cur[t], b, out[t]);
#pragma unroll
      for (int t = 0; t < TOUT; ++t) a_cur[t] = a_nxt[t];
    }
    // next slab must have landed for every wave; the one after it (NBUF = 3) may still be in flight
    if (AHEAD == 2 && g + 2 < G) tf_wait_vmcnt_barrier<4>();
    else tf_wait_vmcnt_barrier<0>();
  }
}

// =====================================================================================================
// f16x3 path: fp32-accurate products on the f16 matrix cores (16x the fp32 MFMA rate per instruction).
// Every operand is split  x = hi + lo  with hi = f16(x), lo = f16(x - hi)  (22 significant bits, abs floor
// ~3e-8 from f16 subnormals) and the product is accumulated in fp32 as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
// (the dropped a_lo*b_lo term is ~2^-22 relative): 3 x v_mfma_f32_32x32x16_f16 replace 8 x
// v_mfma_f32_32x32x2_f32, i.e. 5.3x fewer matrix-core cycles at fp32-level accuracy.
//
// v_mfma_f32_32x32x16_f16: lane l supplies A[i = l&31][k = 8*(l>>5) + e], B[k = 8*(l>>5) + e][j = l&31], e = 0..7;
// the accumulator layout is the fp32 one, so k-step s16 = 2*t + u of the next layer takes element e of lane half h
// from accumulator register 8u + e of unit tile t, i.e. unit  32t + (e&3) + 8*(2u + (e>>2)) + 4h.
// Weight slabs (global, then LDS): [s16][tout][hi|lo][lane][8 halves]  (16 bytes per lane, conflict-free b128 reads).
typedef _Float16 tf_h8 __attribute__((ext_vector_type(8)));

__host__ __device__ __forceinline__ int tf_kmap16(int s16, int h, int e) {
  return 32 * (s16 >> 1) + (e & 3) + 8 * (2 * (s16 & 1) + (e >> 2)) + 4 * h;
}

__device__ __forceinline__ f32x16 tf_mfma_h(tf_h8 a, tf_h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// x[0..7] -> hi = f16(x) (round to nearest even), lo = f16(x - hi).  Three instructions per PAIR of values:
// v_cvt_pk_f16_f32, then v_fma_mixlo_f16 / v_fma_mixhi_f16 compute (f32)hi * -1 + x in fp32 (exact: the difference has
// <= 13 significant bits) and round it to f16 straight into the packed lo register (bit-identical to the C expression
// on 1 M random values incl. the f16-subnormal range).  Written as (_Float16)(x - (float)h) the compiler spends 3
// instructions per VALUE (convert back, subtract, convert).  ONE statement ending in `s_nop 1`: the outputs feed MFMA
// B operands, and hipcc's hazard recognizer does not see VALU writes made inside an asm string (VALU write -> MFMA
// operand needs 2 wait states; without them a few samples per launch read the stale register).
__device__ __forceinline__ void tf_split8(const float (&x)[8], tf_h8& hi, tf_h8& lo) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 h, l;
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  asm("v_cvt_pk_f16_f32 %0, %8, %9\n\t"
      "v_cvt_pk_f16_f32 %1, %10, %11\n\t"
      "v_cvt_pk_f16_f32 %2, %12, %13\n\t"
      "v_cvt_pk_f16_f32 %3, %14, %15\n\t"
      "v_fma_mixlo_f16 %4, %0, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %5, %1, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %6, %2, -1.0, %12 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %7, %3, -1.0, %14 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %4, %0, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %5, %1, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %6, %2, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %7, %3, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 1"
      : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  h[0] = h0; h[1] = h1; h[2] = h2; h[3] = h3;
  l[0] = l0; l[1] = l1; l[2] = l2; l[3] = l3;
  hi = __builtin_bit_cast(tf_h8, h);
  lo = __builtin_bit_cast(tf_h8, l);
}

// hi part only (TF_PREC_F16: plain f16 operands, one MFMA per product term)
__device__ __forceinline__ void tf_cvt8(const float (&x)[8], tf_h8& hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 h;
  unsigned h0, h1, h2, h3;
  asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
      "v_cvt_pk_f16_f32 %2, %8, %9\n\t"
      "v_cvt_pk_f16_f32 %3, %10, %11\n\t"
      "s_nop 1"
      : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  h[0] = h0; h[1] = h1; h[2] = h2; h[3] = h3;
  hi = __builtin_bit_cast(tf_h8, h);
}

// tf_split8 in three pieces of four instructions (convert | low halves of lo | high halves of lo), for placement between MFMAs:
// issued as one 13-instruction block the split kept the matrix pipe idle for ~50 cycles per slab step (-15 % when ablated); a
// lone wave issues a vector instruction every 4 cycles, so four of them fit the 32-cycle shadow of one MFMA.  The consumer is the
// NEXT slab step (behind a barrier), so no hazard nop is needed here.
__device__ __forceinline__ void tf_split8_p0(const float (&x)[8], unsigned (&h)[4]) {
  asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
      "v_cvt_pk_f16_f32 %2, %8, %9\n\t"
      "v_cvt_pk_f16_f32 %3, %10, %11"
      : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
}
__device__ __forceinline__ void tf_split8_p1(const float (&x)[8], const unsigned (&h)[4], unsigned (&l)[4]) {
  asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %5, -1.0, %9 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %2, %6, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %3, %7, -1.0, %11 op_sel_hi:[1,0,0]"
      : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3])
      : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(x[0]), "v"(x[2]), "v"(x[4]), "v"(x[6]));
}
__device__ __forceinline__ void tf_split8_p2(const float (&x)[8], const unsigned (&h)[4], unsigned (&l)[4]) {
  asm("v_fma_mixhi_f16 %0, %4, -1.0, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %5, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %2, %6, -1.0, %10 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %3, %7, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3])
      : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(x[1]), "v"(x[3]), "v"(x[5]), "v"(x[7]));
}

static __global__ void __launch_bounds__(256) tf_pack_wfrag_h3_kernel(const float* __restrict__ W, int nout, int ld, int col0,
                                                               int kin, int tout_tiles, int ksteps16,
                                                               _Float16* __restrict__ dst) {
  int e_ = blockIdx.x * 256 + threadIdx.x;          // one thread per (s16, tout, lane)
  int total = tout_tiles * ksteps16 * 64;
  if (e_ >= total) return;
  int lane = e_ & 63;
  int tout = (e_ >> 6) % tout_tiles;
  int s16 = (e_ >> 6) / tout_tiles;
  int row = 32 * tout + (lane & 31);
  _Float16* base = dst + ((long long)(s16 * tout_tiles + tout) * 2) * 64 * 8;   // [hi|lo][lane][8]
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int k = tf_kmap16(s16, lane >> 5, e);
    float w = (row < nout && k < kin) ? W[(long long)row * ld + col0 + k] : 0.f;
    _Float16 hi = (_Float16)w;
    _Float16 lo = (_Float16)(w - (float)hi);
    base[lane * 8 + e] = hi;
    base[64 * 8 + lane * 8 + e] = lo;
  }
}

// ---- batched packing.  A training step re-packs every network it touches (the weights have just been updated): the flow nets alone
// were 7 launches per coupling net in the forward and 11 in the backward entry point -- 107 tiny launches per material training step.
// TfPackBatch collects the jobs of an entry point and runs them as ONE launch (blockIdx.y = job, blockIdx.x over its elements).
struct TfPackJob {
  const float* src;
  void* dst;
  int kind;                       // 0: tf_pack_wfrag_kernel, 1: tf_pack_wfrag_h3_kernel, 2: tf_pack_bias_kernel, 3: accumulator order, lane-half major
  int nout, ld, col0, kin, tout_tiles, ksteps, s_major, transpose;
};
#define TF_PACK_MAX_JOBS 24
struct TfPackJobs { TfPackJob j[TF_PACK_MAX_JOBS]; int n; };

static __global__ void __launch_bounds__(256) tf_pack_batch_kernel(TfPackJobs J) {
  const TfPackJob& q = J.j[blockIdx.y];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (q.kind == 2) {
    if (e >= q.tout_tiles * 32) return;
    const int h = e & 1, reg = (e >> 1) & 15, tout = e >> 5;
    const int row = 32 * tout + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    reinterpret_cast<float*>(q.dst)[e] = row < q.nout ? q.src[row] : 0.f;
    return;
  }
  if (q.kind == 3) {              // dst[h * (tout_tiles * 16) + tout * 16 + reg] = src[32 tout + rho(reg, h)]  (sdf.hip: ds_read_b128 per lane)
    if (e >= q.tout_tiles * 32) return;
    const int h = e / (q.tout_tiles * 16), r = e % (q.tout_tiles * 16), tout = r >> 4, reg = r & 15;
    const int row = 32 * tout + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    reinterpret_cast<float*>(q.dst)[e] = row < q.nout ? q.src[row] : 0.f;
    return;
  }
  const int total = q.tout_tiles * q.ksteps * 64;
  if (e >= total) return;
  const int lane = e & 63;
  if (q.kind == 0) {
    int s, tout;
    if (q.s_major) { tout = (e >> 6) % q.tout_tiles; s = (e >> 6) / q.tout_tiles; }
    else { s = (e >> 6) % q.ksteps; tout = (e >> 6) / q.ksteps; }
    const int row = 32 * tout + (lane & 31), k = tf_kmap(s, lane >> 5);
    float v = 0.f;
    if (row < q.nout && k < q.kin) v = q.transpose ? q.src[(long long)k * q.ld + q.col0 + row] : q.src[(long long)row * q.ld + q.col0 + k];
    reinterpret_cast<float*>(q.dst)[e] = v;
  } else {
    const int tout = (e >> 6) % q.tout_tiles, s16 = (e >> 6) / q.tout_tiles;
    const int row = 32 * tout + (lane & 31);
    _Float16* base = reinterpret_cast<_Float16*>(q.dst) + ((long long)(s16 * q.tout_tiles + tout) * 2) * 64 * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int k = tf_kmap16(s16, lane >> 5, c);
      const float w = (row < q.nout && k < q.kin) ? q.src[(long long)row * q.ld + q.col0 + k] : 0.f;
      const _Float16 hi = (_Float16)w;
      base[lane * 8 + c] = hi;
      base[64 * 8 + lane * 8 + c] = (_Float16)(w - (float)hi);
    }
  }
}

struct TfPackBatch {
  TfPackJobs J;
  int max_elems;
  hipStream_t stream;
  explicit TfPackBatch(hipStream_t s) : max_elems(0), stream(s) { J.n = 0; }
  void add(const TfPackJob& q, int elems) {
    if (J.n == TF_PACK_MAX_JOBS) flush();
    J.j[J.n++] = q;
    max_elems = elems > max_elems ? elems : max_elems;
  }
  // the argument lists of the single-job kernels above
  void wfrag(const float* W, int nout, int ld, int col0, int kin, int tout_tiles, int ksteps, float* dst, int s_major = 0, int transpose = 0) {
    add(TfPackJob{W, dst, 0, nout, ld, col0, kin, tout_tiles, ksteps, s_major, transpose}, tout_tiles * ksteps * 64);
  }
  void wfrag_h3(const float* W, int nout, int ld, int col0, int kin, int tout_tiles, int ksteps16, _Float16* dst) {
    add(TfPackJob{W, dst, 1, nout, ld, col0, kin, tout_tiles, ksteps16, 0, 0}, tout_tiles * ksteps16 * 64);
  }
  void bias(const float* b, int n, int tout_tiles, float* dst) { add(TfPackJob{b, dst, 2, n, 0, 0, 0, tout_tiles, 0, 0, 0}, tout_tiles * 32); }
  void acc_major(const float* b, int n, int tout_tiles, float* dst) { add(TfPackJob{b, dst, 3, n, 0, 0, 0, tout_tiles, 0, 0, 0}, tout_tiles * 32); }
  void flush() {
    if (J.n == 0) return;
    tf_pack_batch_kernel<<<dim3((unsigned)((max_elems + 255) / 256), (unsigned)J.n), 256, 0, stream>>>(J);
    J.n = 0;
    max_elems = 0;
  }
};

// Dense layer, f16x3, weights streamed through LDS in 16 KB slabs of SL16 k-steps (SL16*TOUT*2 KB == 16 KB).
template <int K16, int TOUT, int TIN, int SL16, int NBUF = 3>
__device__ __forceinline__ void tf_layer_stream_h3(const _Float16* __restrict__ wslab, float* __restrict__ lds, int tid, int lane,
                                                   const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
  static_assert(SL16 * TOUT * 2048 == 16384, "one slab must be 16 KB");
  static_assert(K16 % SL16 == 0, "K16 must be a multiple of the slab size");
  constexpr int G = K16 / SL16;
  constexpr int AHEAD = NBUF - 1;
  const int wave = tid >> 6;
  const float* gp = reinterpret_cast<const float*>(wslab) + wave * 1024 + lane * 4;
  asm volatile("" : "+v"(gp));
  tf_slab_dma(gp, lds, wave);
  if (AHEAD == 2 && G > 1) {
    gp += 4096;
    asm volatile("" : "+v"(gp));
    tf_slab_dma(gp, lds + 4096, wave);
    tf_wait_vmcnt_barrier<4>();
  } else {
    tf_wait_vmcnt_barrier<0>();
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g + AHEAD < G) {
      gp += 4096;
      asm volatile("" : "+v"(gp));
      tf_slab_dma(gp, lds + ((g + AHEAD) % NBUF) * 4096, wave);
    }
    const tf_h8* buf = reinterpret_cast<const tf_h8*>(lds + (g % NBUF) * 4096) + lane;   // 16-byte units
#pragma unroll
    for (int sl = 0; sl < SL16; ++sl) {
      const int s16 = g * SL16 + sl;
      tf_h8 b_hi, b_lo;
      {
        float x8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x8[e] = in[s16 >> 1][8 * (s16 & 1) + e];
        tf_split8(x8, b_hi, b_lo);
      }
      tf_h8 a_hi = buf[(sl * TOUT) * 128], a_lo = buf[(sl * TOUT) * 128 + 64];
#pragma unroll
      for (int t = 0; t < TOUT; ++t) {
        tf_h8 n_hi = a_hi, n_lo = a_lo;
        if (t + 1 < TOUT) {   // next tile's fragments are requested before this tile's MFMAs
          n_hi = buf[(sl * TOUT + t + 1) * 128];
          n_lo = buf[(sl * TOUT + t + 1) * 128 + 64];
        }
        out[t] = tf_mfma_h(a_hi, b_hi, out[t]);
        out[t] = tf_mfma_h(a_hi, b_lo, out[t]);
        out[t] = tf_mfma_h(a_lo, b_hi, out[t]);
        a_hi = n_hi; a_lo = n_lo;
      }
    }
    if (AHEAD == 2 && g + 2 < G) tf_wait_vmcnt_barrier<4>();
    else tf_wait_vmcnt_barrier<0>();
  }
}

// ---- software-pipelined streaming.  tf_layer_stream_h3 issues a slab's 16 fragment reads and then needs the first one:
// all four waves leave the barrier together, their 64 KB of ds_read_b128 queue on the LDS (>= 256 cycles) while the
// matrix cores idle -- a third of every slab step.  Here the fragments of slab g+1 are read into a SECOND register set
// while the MFMAs of slab g run, and the LDS ring is 4 slabs deep (slab g+3 is requested while g computes).
// Per step:  counted vmcnt -> s_barrier -> MFMAs of slab g with the reads of slab g+1 and the DMA of slab g+3 interleaved.
// The compiler waits with lgkmcnt(0) before the first MFMA that needs a fragment; placing this step's reads BEHIND the
// first MFMA group (scheduling barriers pin the order) makes that wait cover only reads issued most of a step earlier.
//   RAW: a slab is read one phase after the vmcnt + barrier that retire its DMA.
//   WAR: DMA(g+3) overwrites the buffer of slab g-1, whose reads every wave completed before its MFMAs of step g-1,
//        i.e. before this step's barrier.
struct TfFrag { tf_h8 hi[8], lo[8]; };

__device__ __forceinline__ void tf_frag_read(TfFrag& F, const tf_h8* __restrict__ buf /* slab + lane */) {
#pragma unroll
  for (int c = 0; c < 8; ++c) { F.hi[c] = buf[c * 128]; F.lo[c] = buf[c * 128 + 64]; }
}

// ---- continuous weight stream.  A pipeline restarted at every layer pays three slab requests, an L2 round trip and a barrier
// before the layer's first MFMA.  A network's whole f16x3 image is therefore laid out as ONE sequence of 16 KB slabs (layer
// after layer), and the ring simply keeps running: the last steps of a layer already request and read the first slabs of the
// next one, and the last layer of a tile those of the next tile's first layer (the sequence wraps).  Per step, unconditionally:
//   s_waitcnt vmcnt(4) -> s_barrier -> MFMAs of slab s, interleaved with the fragment reads of slab s+1 and the DMA of s+3.
// (vmcnt(4): at least the four DMA pieces of slab s+2 were issued after slab s+1; other vector-memory operations issued in
// between only make the wait longer.)  Ring slots are run-time (slab count per pass need not be a multiple of 4); the two
// fragment register sets alternate per slab, so a layer is instantiated for the parity P0 of its first slab.
struct TfStream {
  const float* gp;    // this thread's source pointer of the next slab to request
  const float* g0;    // ... of slab 0
  int next, total;    // index of the next slab to request / slabs per pass
  unsigned slot_req;  // ring slot the next request goes to
  unsigned slot_rd;   // ring slot of the slab whose fragments are read next
  float* lds;         // 4 x 4096 floats
  int lane, wave;
};

__device__ __forceinline__ void tf_stream_advance(TfStream& S) {
  S.gp += 4096;
  if (++S.next == S.total) { S.next = 0; S.gp = S.g0; }
  S.slot_req = (S.slot_req + 1) & 3;
}

// Once per kernel, by all four waves: request slabs 0..2, read slab 0 into F0.  `total` >= 3.
__device__ __forceinline__ void tf_stream_begin(TfStream& S, const _Float16* wslab, int total, float* lds, int tid, int lane, TfFrag& F0) {
  S.lds = lds; S.lane = lane; S.wave = tid >> 6; S.total = total; S.next = 0; S.slot_req = 0;
  S.g0 = reinterpret_cast<const float*>(wslab) + S.wave * 1024 + lane * 4;
  S.gp = S.g0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    asm volatile("" : "+v"(S.gp));
    tf_slab_dma(S.gp, S.lds + S.slot_req * 4096, S.wave);
    tf_stream_advance(S);
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  tf_frag_read(F0, reinterpret_cast<const tf_h8*>(S.lds) + lane);
  S.slot_rd = 1;
  __builtin_amdgcn_sched_barrier(0);
}

// Before the kernel returns: the requests issued for a tile that does not exist must land before the LDS is released.
__device__ __forceinline__ void tf_stream_end() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// TERMS = 3: f16x3 (fp32-accurate).  TERMS = 1: plain f16 operands (TF_PREC_F16) -- the lo fragments are neither read nor used.
// ReLU of a matrix-core result in ONE instruction: median(x, 0, big) = max(x, 0).  fmaxf(x, 0.f) costs two -- the compiler first
// canonicalises x (v_max x, x), because an MFMA result is not a known-canonical float to it; `big` is made opaque so that the
// median is not folded back into that form.
__device__ __forceinline__ float tf_relu(float x) {
  float big = 3.0e38f;
  asm("" : "+s"(big));
  return __builtin_amdgcn_fmed3f(x, 0.f, big);
}

// B operands (activations) of one slab step: SL16 k-steps, split hi | lo.
template <int SL16>
struct TfBsplit { tf_h8 hi[SL16], lo[SL16]; };

template <int SL16, int TIN, int TERMS>
__device__ __forceinline__ void tf_bsplit(TfBsplit<SL16>& B, int s16base, const f32x16 (&in)[TIN]) {
#pragma unroll
  for (int sl = 0; sl < SL16; ++sl) {
    const int s16 = s16base + sl;
    float x8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x8[e] = in[s16 >> 1][8 * (s16 & 1) + e];
#ifdef TF_ABLATE_SPLIT   // dev-only timing ablation: operands are register reinterpretations, no conversion (results are garbage)
    typedef float tf_f4 __attribute__((ext_vector_type(4)));
    const tf_f4 va = {x8[0], x8[1], x8[2], x8[3]}, vb = {x8[4], x8[5], x8[6], x8[7]};
    B.hi[sl] = __builtin_bit_cast(tf_h8, va);
    B.lo[sl] = __builtin_bit_cast(tf_h8, vb);
#else
    if (TERMS >= 3) tf_split8(x8, B.hi[sl], B.lo[sl]);
    else tf_cvt8(x8, B.hi[sl]);
#endif
  }
}

// TERMS = 3: f16x3 (fp32-accurate).  TERMS = 1: plain f16 operands (TF_PREC_F16) -- the lo fragments are neither read nor used.
// TERMS = 2: weights split (hi + lo), activations rounded to f16 once per layer (w_hi x + w_lo x).
// TERMS = 4: weights rounded to f16, activations split (w_hi x_hi + w_hi x_lo): the lo fragments are not read.
// `bc`: this step's B operands, prepared by the PREVIOUS step (or by the layer prologue); `bn` (NEXT >= 0): the next step's are
// split here, behind the second MFMA group -- at the step boundary the split's 13 vector instructions ran with an idle matrix
// pipe (one wave per SIMD: nothing else to issue), 42 times per tile.
template <int TOUT, int TIN, int SL16, int TERMS>
__device__ __forceinline__ void tf_h3s_step(TfStream& S, const TfFrag& cur, TfFrag& nxt, const TfBsplit<SL16>& bc, TfBsplit<SL16>& bn,
                                            int NEXT /* first k-step of the next slab step, -1: none (constant after inlining) */,
                                            const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#ifndef TF_ABLATE_BARRIER   // dev-only timing ablation: results are garbage when defined
  __builtin_amdgcn_s_barrier();
#endif
  asm volatile("" ::: "memory");
  const tf_h8* nbuf = reinterpret_cast<const tf_h8*>(S.lds + S.slot_rd * 4096) + S.lane;
  float* dbuf = S.lds + S.slot_req * 4096;
  const float* gsrc = S.gp;
  asm volatile("" : "+v"(gsrc));
  unsigned sp_h[SL16 <= 2 ? SL16 : 1][4], sp_l[SL16 <= 2 ? SL16 : 1][4];   // split pieces in flight across the MFMA groups
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int sl = c / TOUT, t = c % TOUT;
    if (c >= 1 && c <= 4) {
#pragma unroll
      for (int q = 2 * (c - 1); q < 2 * c; ++q) {
        nxt.hi[q] = nbuf[q * 128];
        if (TERMS == 3 || TERMS == 2) nxt.lo[q] = nbuf[q * 128 + 64];
      }
    }
#ifdef TF_ABLATE_HALF_DMA   // dev-only timing ablation: half the weight bytes are fetched (results are garbage)
    if (c >= 4 && ((c - 4) & 1) == 0) tf_slab_dma_piece(gsrc, dbuf, S.wave, c - 4);
#else
    if (c >= 4) tf_slab_dma_piece(gsrc, dbuf, S.wave, c - 4);
#endif
#ifndef TF_SPLIT_AT_BOUNDARY
    if (NEXT >= 0) {
      if (TERMS >= 3 && SL16 <= 2) {
        // one four-instruction piece per MFMA group: k-step 0 of the next slab in groups 5, 6, 7 (SL16 = 1) or 2, 3, 4 with
        // k-step 1 in 5, 6, 7 (SL16 = 2)
        const int first = SL16 == 1 ? 5 : 2;
        if (c >= first) {
          const int sl_n = (c - first) / 3, piece = (c - first) % 3, s16n = NEXT + sl_n;
          float x8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = in[s16n >> 1][8 * (s16n & 1) + e];
          if (piece == 0) tf_split8_p0(x8, sp_h[sl_n]);
          else if (piece == 1) tf_split8_p1(x8, sp_h[sl_n], sp_l[sl_n]);
          else {
            tf_split8_p2(x8, sp_h[sl_n], sp_l[sl_n]);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 hv = {sp_h[sl_n][0], sp_h[sl_n][1], sp_h[sl_n][2], sp_h[sl_n][3]};
            const u32x4 lv = {sp_l[sl_n][0], sp_l[sl_n][1], sp_l[sl_n][2], sp_l[sl_n][3]};
            bn.hi[sl_n] = __builtin_bit_cast(tf_h8, hv);
            bn.lo[sl_n] = __builtin_bit_cast(tf_h8, lv);
          }
        }
      } else if (c == 5) {
        tf_bsplit<SL16, TIN, TERMS>(bn, NEXT, in);
      }
    }
#endif
    out[t] = tf_mfma_h(cur.hi[c], bc.hi[sl], out[t]);
    if (TERMS >= 3) out[t] = tf_mfma_h(cur.hi[c], bc.lo[sl], out[t]);
    if (TERMS == 3 || TERMS == 2) out[t] = tf_mfma_h(cur.lo[c], bc.hi[sl], out[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef TF_SPLIT_AT_BOUNDARY   // dev-only: the previous placement
  if (NEXT >= 0) tf_bsplit<SL16, TIN, TERMS>(bn, NEXT, in);
#endif
  tf_stream_advance(S);
  S.slot_rd = (S.slot_rd + 1) & 3;
}

// One dense layer on the stream: K16 k-steps of TOUT unit tiles (G = K16 * TOUT / 8 slabs); P0 = parity of its first slab
// (which of FA / FB already holds that slab's fragments).  Returns nothing; the caller continues with parity (P0 + G) & 1.
// (Issuing the three terms of a tile back to back on its accumulator is the fast order: a term-major order -- eight hi*hi,
// then eight hi*lo, then eight lo*hi -- measured 6 % slower.)
template <int K16, int TOUT, int TIN, int P0, int TERMS = 3>
__device__ __forceinline__ void tf_layer_h3s(TfStream& S, TfFrag& FA, TfFrag& FB, const f32x16 (&in)[TIN], f32x16 (&out)[TOUT]) {
  static_assert(8 % TOUT == 0, "a 16 KB slab holds 8 (k-step, tile) fragment pairs");
  constexpr int SL16 = 8 / TOUT;
  static_assert(K16 % SL16 == 0, "K16 must be a multiple of the slab size");
  constexpr int G = K16 / SL16;
  TfBsplit<SL16> B0, B1;
  tf_bsplit<SL16, TIN, TERMS>(B0, 0, in);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    // static indices after unrolling: parity of the fragment sets and of the operand sets
    if (((P0 + g) & 1) == 0) {
      if ((g & 1) == 0) tf_h3s_step<TOUT, TIN, SL16, TERMS>(S, FA, FB, B0, B1, g + 1 < G ? (g + 1) * SL16 : -1, in, out);
      else tf_h3s_step<TOUT, TIN, SL16, TERMS>(S, FA, FB, B1, B0, g + 1 < G ? (g + 1) * SL16 : -1, in, out);
    } else {
      if ((g & 1) == 0) tf_h3s_step<TOUT, TIN, SL16, TERMS>(S, FB, FA, B0, B1, g + 1 < G ? (g + 1) * SL16 : -1, in, out);
      else tf_h3s_step<TOUT, TIN, SL16, TERMS>(S, FB, FA, B1, B0, g + 1 < G ? (g + 1) * SL16 : -1, in, out);
    }
  }
}

// ---- pre-split variant: the layer's whole input is converted to (hi | lo) f16 operands ONCE, at the layer boundary, instead of
// k-step by k-step inside the slab steps.  Inside a step every operand conversion reads accumulator registers of the previous
// layer while the matrix pipe is writing the current ones, and costs 15 % of the kernel (ablation) wherever its instructions are
// placed; at the boundary nothing is in flight.  The fp32 input dies with the conversion (its registers become the next output).
template <int K16>
struct TfSplitIn { tf_h8 hi[K16], lo[K16]; };

template <int K16, int TIN, int TERMS>
__device__ __forceinline__ void tf_presplit(const f32x16 (&x)[TIN], TfSplitIn<K16>& B) {
#pragma unroll
  for (int s16 = 0; s16 < K16; ++s16) {
    float x8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x8[e] = x[s16 >> 1][8 * (s16 & 1) + e];
    if (TERMS == 3) tf_split8(x8, B.hi[s16], B.lo[s16]);
    else tf_cvt8(x8, B.hi[s16]);
  }
}

template <int TOUT, int K16, int SL16, int TERMS>
__device__ __forceinline__ void tf_h3s_step_ps(TfStream& S, int s16base, const TfFrag& cur, TfFrag& nxt, const TfSplitIn<K16>& B,
                                               f32x16 (&out)[TOUT]) {
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const tf_h8* nbuf = reinterpret_cast<const tf_h8*>(S.lds + S.slot_rd * 4096) + S.lane;
  float* dbuf = S.lds + S.slot_req * 4096;
  const float* gsrc = S.gp;
  asm volatile("" : "+v"(gsrc));
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int sl = c / TOUT, t = c % TOUT;
    if (c >= 1 && c <= 4) {
#pragma unroll
      for (int q = 2 * (c - 1); q < 2 * c; ++q) {
        nxt.hi[q] = nbuf[q * 128];
        if (TERMS == 3) nxt.lo[q] = nbuf[q * 128 + 64];
      }
    }
    if (c >= 4) tf_slab_dma_piece(gsrc, dbuf, S.wave, c - 4);
    out[t] = tf_mfma_h(cur.hi[c], B.hi[s16base + sl], out[t]);
    if (TERMS == 3) {
      out[t] = tf_mfma_h(cur.hi[c], B.lo[s16base + sl], out[t]);
      out[t] = tf_mfma_h(cur.lo[c], B.hi[s16base + sl], out[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  tf_stream_advance(S);
  S.slot_rd = (S.slot_rd + 1) & 3;
}

template <int K16, int TOUT, int P0, int TERMS>
__device__ __forceinline__ void tf_layer_h3s_ps(TfStream& S, TfFrag& FA, TfFrag& FB, const TfSplitIn<K16>& B, f32x16 (&out)[TOUT]) {
  static_assert(8 % TOUT == 0, "a 16 KB slab holds 8 (k-step, tile) fragment pairs");
  constexpr int SL16 = 8 / TOUT;
  static_assert(K16 % SL16 == 0, "K16 must be a multiple of the slab size");
  constexpr int G = K16 / SL16;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (((P0 + g) & 1) == 0) tf_h3s_step_ps<TOUT, K16, SL16, TERMS>(S, g * SL16, FA, FB, B, out);
    else tf_h3s_step_ps<TOUT, K16, SL16, TERMS>(S, g * SL16, FB, FA, B, out);
  }
}

// Dense layer, f16x3, fragment weights resident in LDS ([s16][tout][hi|lo][lane][8 halves]).
template <int K16, int TOUT, int TIN, int TERMS = 3>
__device__ __forceinline__ void tf_layer_h3(const tf_h8* __restrict__ wf /* + lane */, const f32x16 (&in)[TIN],
                                            f32x16 (&out)[TOUT]) {
#pragma unroll
  for (int s16 = 0; s16 < K16; ++s16) {
    tf_h8 b_hi, b_lo;
    {
      float x8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x8[e] = in[s16 >> 1][8 * (s16 & 1) + e];
      if (TERMS == 3) tf_split8(x8, b_hi, b_lo);
      else tf_cvt8(x8, b_hi);
    }
#pragma unroll
    for (int t = 0; t < TOUT; ++t) {
      const tf_h8 a_hi = wf[(s16 * TOUT + t) * 128];
      out[t] = tf_mfma_h(a_hi, b_hi, out[t]);
      if (TERMS == 3) {
        const tf_h8 a_lo = wf[(s16 * TOUT + t) * 128 + 64];
        out[t] = tf_mfma_h(a_hi, b_lo, out[t]);
        out[t] = tf_mfma_h(a_lo, b_hi, out[t]);
      }
    }
  }
}
