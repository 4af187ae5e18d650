/*
 * tensoflow_hip.h -- C ABI of libtensoflow_hip.so (MI355X / gfx950).
 *
 * The reference (fudan-zvg/tensoflow) has no FFI for its hot path: it is PyTorch modules
 * calling third-party CUDA extensions.  Its only native-boundary pattern is
 * network/renderutils/ops.py:391-425 (autograd.Function -> pybind op of a JIT extension).
 * This header is the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md).
 * Each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns all memory (inputs, outputs, workspaces); nothing is retained;
 *   - work is enqueued on `stream` (a hipStream_t) and returns without synchronising;
 *   - return 0 on success, negative TfStatus otherwise; message via tf_last_error()
 *     (thread-local); no C++ exception crosses the boundary;
 *   - all floating point is fp32, indices int64 (torch.long) unless stated.
 */
#ifndef TENSOFLOW_HIP_H
#define TENSOFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tf_stream_t; /* hipStream_t */

typedef enum TfStatus {
  TF_OK = 0,
  TF_EINVAL = -1, /* null pointer / bad flag */
  TF_ESHAPE = -2, /* sizes inconsistent with what the kernel and its grid assume */
  TF_EHIP = -3    /* a HIP runtime call failed */
} TfStatus;

/* Matrix-core arithmetic of the decoders.  TF_PREC_F32: v_mfma_f32_32x32x2_f32, bitwise an fp32 fma chain.
 * TF_PREC_F16X3: every operand split x = hi + lo in f16 and a_hi*b_hi + a_hi*b_lo + a_lo*b_hi accumulated in
 * fp32 on v_mfma_f32_32x32x16_f16 (22 significant bits per operand; 5.3x fewer matrix-core cycles).
 * TF_PREC_F16: plain f16 operands (the hi halves only), fp32 accumulate -- one MFMA per product term; in the flow nets NOT
 * within the 1e-4 parity bar (BASELINE configs[4] "fp16 field + flow": reported with its PSNR against the fp32-accurate path).
 * Accepted by tf_flow_sample_fwd / tf_flow_logq_fwd and tf_inner_light_fwd / tf_inner_light_indexed_fwd; others reject it.
 * TF_PREC_F16X2 (inner-light net only): weights split hi + lo, activations rounded to f16 once per layer (w_hi x + w_lo x): per ray
 * inside the error of fp32 PyTorch against fp64; runs on the 128-ray form of the staggered kernel. */
typedef enum TfPrecision { TF_PREC_F32 = 0, TF_PREC_F16X3 = 1, TF_PREC_F16 = 2, TF_PREC_F16X2 = 3, TF_PREC_BF16X3 = 4 } TfPrecision;
/* OR-ed into a `precision` argument: `workspace` still holds this network's packed weights from an earlier call with
 * the same weights and precision (the caller tracks weight updates), so the fragment re-pack launches are skipped. */
#define TF_WEIGHTS_PACKED 0x100

/* Activation fused into tf_linear_fwd / differentiated by tf_linear_bwd. */
typedef enum TfActivation {
  TF_ACT_NONE = 0,
  TF_ACT_RELU = 1,
  TF_ACT_SOFTPLUS = 2,  /* torch.nn.Softplus(beta = act_param, threshold = 20): TensoSDF decoder (network/fields.py:79) */
  TF_ACT_SIGMOID = 3,   /* material predictors' final activation (network/other_field.py:50-84) */
  TF_ACT_EXP_CLAMP = 4  /* ExpActivation: exp(min(z, act_param)) (network/other_field.py:12-18) */
} TfActivation;

int tf_version(void);
const char* tf_last_error(void);

/* Launch budget of the three stage kernels of the rendering integral (per calling thread; 0 = the kernel's own default), read at
 * enqueue time by tf_bvh_trace, tf_flow_sample_fwd / tf_flow_logq_fwd and tf_inner_light_indexed_fwd.  The reference runs
 * sample -> get_lights -> reduce strictly one after another (network/fields.py:1085-1225); here a caller may run the stages of
 * successive sub-batches on different streams, and whether two kernels are CO-RESIDENT on a CU is decided by registers, LDS and wave
 * slots -- so each kernel can be told to take less than the whole CU:
 *   bvh_blocks_per_cu    1..8: persistent traversal workgroups per CU (one wave per SIMD each, 72 registers; default: all that fit);
 *   flow_waves_per_block 4 / 8 / 12: waves of the flow kernel's one workgroup per CU (one / two / three per SIMD, 168 registers each);
 *   inner_teams          1: the inner-light kernel as ONE four-wave team per workgroup (one wave per SIMD, half a CU's registers, 88 KB of
 *                        LDS); 2: the staggered two-team workgroup that owns a CU.
 * Results do not depend on the budget (bit-identical). */
int tf_set_launch_budget(int32_t bvh_blocks_per_cu, int32_t flow_waves_per_block, int32_t inner_teams);

/* Measurement aid (no reference counterpart: the reference never prices its kernels; SURVEY.md 8(d) asks for achieved / peak of the
 * dominant kernel).  Runs a dense v_mfma_f32_32x32x16_f16 stream shaped like the inner-light decoder's k-step (64 units x 64 rays,
 * three product terms, operands re-read from LDS, pseudo-random f16 data, one wave per SIMD on every CU) for `iters` iterations per
 * wave, SYNCHRONOUSLY (it times itself with HIP events on `stream`), and returns the executed TFLOP/s in *tflops_host: the rate the
 * matrix cores of THIS device hold under that load (the part lowers its clock: ~1.5-1.6 PFLOP/s against the 2.5 PFLOP/s quoted).
 * relu_like != 0: the B operands (the "activations") are ReLU outputs -- about half of them zero, as in the decoder's hidden layers; the
 * clock the part holds depends on what the multipliers toggle, so this is the ceiling for the decoder's own operand statistics.
 * scratch: device buffer of >= 256 floats per CU (written). */
int tf_probe_mfma_f16(int32_t iters, int32_t relu_like, float* scratch, int64_t scratch_floats, double* tflops_host, tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * VM-decomposed tensorial field (3 planes + 3 lines, C components each).
 * Replaces the 6 x dr.texture(...) + permute/contiguous + per-call mip rebuild of
 * network/fields.py:262-291 (TensoSDF), :776-806 (material field), network/flow.py:709-738.
 *
 * Packed pyramid layout in HBM (floats), built once per optimizer step by tf_vm_pack_fwd:
 *   for i in 0..2: for l in 0..n_levels-1:  plane_i level l  as [H>>l][W>>l][C]   (channel-last,
 *   one texel = C*4 contiguous bytes), then for i: for l: line_i level l as [L>>l][C].
 * Plane i is addressed with u = xyz[mat_mode[i][0]] along W and v = xyz[mat_mode[i][1]] along H
 * (mat_mode = {0,1},{0,2},{1,2}); line i with xyz[vec_mode[i]] (vec_mode = 2,1,0).
 * Sizes > 1 must be divisible by 2^(n_levels-1).
 * ------------------------------------------------------------------------------------------ */
typedef struct TfVmDesc {
  int32_t C;        /* components per plane/line (36 sdf/material, 12 flow) */
  int32_t n_levels; /* mip levels incl. level 0 (1..4) */
  int32_t ph[3];    /* plane heights at level 0 (v axis) */
  int32_t pw[3];    /* plane widths  at level 0 (u axis) */
  int32_t ll[3];    /* line lengths  at level 0 */
  int32_t texel_f16; /* 0: the pyramid holds fp32 texels.  1: IEEE half texels (BASELINE configs[4] "fp16 field"): `packed` arguments
                      * of the gather / decoder entry points then point at tf_vm_packed_floats() HALVES (same element offsets, half the
                      * bytes: a C = 36 texel is 72 B instead of 144 B); taps are widened to fp32 on load, blends and products stay fp32.
                      * Inference only (the gather / pack adjoints take fp32 pyramids).  Not parity-grade: opt-in. */
} TfVmDesc;

size_t tf_vm_packed_floats(const TfVmDesc* d);

/* The half pyramid of an fp32 pyramid built by tf_vm_pack_fwd: packed16[e] = round-to-nearest-even(packed[e]) for all
 * tf_vm_packed_floats(d) elements (mips are averaged in fp32 first, rounded once).  Replaces the `.half()` of the field tensors
 * in an fp16 evaluation of network/fields.py:276-288,790-802 and network/flow.py:723-735. */
int tf_vm_pack_to_f16(const TfVmDesc* d, const float* packed, void* packed16, tf_stream_t stream);

/* planes[i]: [C,H,W] (the reference's nn.Parameter [1,C,H,W]); lines[i]: [C,L] ([1,C,L,1]). */
int tf_vm_pack_fwd(const TfVmDesc* d, const float* const planes[3], const float* const lines[3],
                   float* packed, tf_stream_t stream);
/* adjoint of tf_vm_pack_fwd: folds the pyramid gradient back to [C,H,W] / [C,L] (overwrites). */
int tf_vm_pack_bwd(const TfVmDesc* d, const float* gpacked, float* const gplanes[3],
                   float* const glines[3], tf_stream_t stream);

/* feat[n, i*C + c] = plane_i(c) * line_i(c) at xyz[n]; level may be NULL (== 0).
 * aabb_host: 6 floats (min xyz, max xyz), contraction of utils/network_utils.py:90-91. */
int tf_vm_gather_fwd(const TfVmDesc* d, const float* packed, const float* xyz, const float* level,
                     const float* aabb_host, int64_t n, float* feat, tf_stream_t stream);
/* gpacked += d feat / d packed (float atomics; zero it first). */
int tf_vm_gather_bwd(const TfVmDesc* d, const float* packed, const float* xyz, const float* level,
                     const float* aabb_host, int64_t n, const float* gfeat, float* gpacked,
                     tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * TensoSDF decoder: cat[feat(3C), xyz(3)] -> Linear(3C+3, Hd) -> Softplus(beta=100) ->
 * Linear(Hd, 1+A)   (network/fields.py:78-81, :293-299).  Weights in torch layout.
 * ------------------------------------------------------------------------------------------ */
typedef struct TfSdfMlp {
  const float* w1; /* [Hd, 3C+3] */
  const float* b1; /* [Hd] */
  const float* w2; /* [1+A, Hd] */
  const float* b2; /* [1+A] */
  int32_t hidden;  /* Hd (256) */
  int32_t app_dim; /* A (128) */
} TfSdfMlp;

/* Device scratch (floats) the two SDF entry points need for fragment-ordered weights. */
size_t tf_sdf_workspace_floats(void);

/* TensoSDF.forward (fields.py:262-299), returned split the way every caller slices it
 * (fields.py:148-152): sdf [n] = out[:,0], feat [n,A] = out[:,1:] (feat may be NULL: sdf only).
 * This build instantiates C = 36, Hd = 256, A = 128 (configs/shape/syn/compressor.yaml:64-66).  * precision (TfPrecision): arithmetic of the two decoder products.  TF_PREC_F16X3 (3 x f16 MFMA per fp32 product term,
 * ~22 significant bits) makes the fused march kernel gather-bound instead of fp32-MFMA-bound; the second finite difference
 * `nhess` divides the decoder's rounding noise by eps^2, so training passes that need it use TF_PREC_F32.
 */
int tf_sdf_forward(const TfVmDesc* d, const float* packed, const TfSdfMlp* mlp, const float* xyz,
                   const float* level, const float* aabb_host, int64_t n, float* sdf, float* feat,
                   int32_t precision, float* workspace, size_t workspace_floats, tf_stream_t stream);

/* ShapeRenderer.compute_sdf_alpha (shapeRenderer.py:995-1025) = forward + 6-tap central
 * differences (fields.py:227-260) + NeuS alpha.  units_host[3] = aabbSize/(R-1).
 * Outputs: alpha[n], grad[n,3], feat[n,A] (may be NULL), sdf[n], nhess[n] (normal_hessian; may be NULL), taps[n,6] (may be NULL:
 * the six finite-difference sdf values x+, x-, y+, y-, z+, z- of fields.py:236-249, kept for tf_sdf_alpha_bwd). */
int tf_sdf_alpha_fwd(const TfVmDesc* d, const float* packed, const TfSdfMlp* mlp, const float* pts,
                     const float* level, const float* dists, const float* dirs, const float* aabb_host,
                     const float* units_host, float inv_s, float cos_anneal, int64_t n, float* alpha,
                     float* grad, float* feat, float* sdf, float* nhess, float* taps, int32_t precision, float* workspace,
                     size_t workspace_floats, tf_stream_t stream);

/* Backward of compute_sdf_alpha -- what autograd does to shapeRenderer.py:995-1025 over fields.py:227-260 (central differences,
 * normal_hessian), :262-299 (7 x gather + decoder) and other_field.py:193-207 (inv_s) in the reference -- as ONE entry point: the
 * closed-form adjoint of alpha / cos annealing / finite differences / hessian term per sample, one recompute of the hidden layer per
 * tap (same arithmetic as the forward: `precision`), the decoder's three products on the exact-fp32 matrix cores and the 7-tap
 * scatter into the pyramid gradient.  Inputs as tf_sdf_alpha_fwd plus its outputs sdf [n] and taps [n,6] and the upstream gradients
 * g_alpha [n], g_grad [n,3], g_feat [n,A], g_sdf [n], g_nhess [n] (each may be NULL = zero).
 * Outputs: gpacked (pyramid-shaped, += with fp32 atomics: zero it first; tf_vm_pack_bwd folds it back to planes / lines),
 * g_w1 [Hd, 3C+3], g_b1 [Hd], g_w2 [1+A, Hd], g_b2 [1+A] (overwritten), g_inv_s (one device float, overwritten; may be NULL).
 * workspace: tf_sdf_alpha_bwd_workspace_floats(n) floats (bounded: samples are processed in chunks of 2^18). */
size_t tf_sdf_alpha_bwd_workspace_floats(int64_t n);
int tf_sdf_alpha_bwd(const TfVmDesc* d, const float* packed, const TfSdfMlp* mlp, const float* pts, const float* level,
                     const float* dists, const float* dirs, const float* aabb_host, const float* units_host, float inv_s,
                     float cos_anneal, int64_t n, const float* sdf, const float* taps, const float* g_alpha, const float* g_grad,
                     const float* g_feat, const float* g_sdf, const float* g_nhess, float* gpacked, float* g_w1, float* g_b1,
                     float* g_w2, float* g_b2, float* g_inv_s, int32_t precision, float* workspace, size_t workspace_floats,
                     tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Packed-ray compositing: nerfacc.render_weight_from_alpha + accumulate_along_rays
 * (call sites network/shapeRenderer.py:1166-1206).  ray_indices sorted ascending.
 *   weights[i] = alpha[i] * prod_{j<i, same ray}(1-alpha[j]);  acc[r] = sum w;  out[r,:] = sum w*values
 * ------------------------------------------------------------------------------------------ */
int tf_composite_fwd(const float* alpha, const int64_t* ray_indices, const float* values, int64_t n,
                     int64_t n_rays, int32_t k, float* weights, float* acc, float* out,
                     tf_stream_t stream);
/* gradients wrt alpha and values given g_acc[n_rays], g_out[n_rays,k] (either may be NULL). */
int tf_composite_bwd(const float* alpha, const int64_t* ray_indices, const float* values,
                     const float* weights, const float* g_acc, const float* g_out, int64_t n,
                     int64_t n_rays, int32_t k, float* g_alpha, float* g_values, tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * TensoFlow coupling flow ('pwquad', 2 blocks, n_bins = 10): network/flow.py:314-525, :549-641,
 * :766-855.  Net k (flows.k.nn.{1,3,5,7}): Linear 44->64->64->64->21, LeakyReLU(0.01), input
 * Reshift(2,-1).  cond [pn,37] = [nis feature 16 | embed3(view_angles) 14 | 0 x 7].
 * ------------------------------------------------------------------------------------------ */
typedef struct TfCouplingNet {
  const float* w[4]; /* [64,44] [64,64] [64,64] [21,64] */
  const float* b[4];
} TfCouplingNet;

/* Device scratch (floats) both flow entry points need: fragment-ordered weights + the per-point
 * hoisted layer-1 vectors [2,pn,64]. */
size_t tf_flow_workspace_floats(int64_t pn);

/* TensoFlow.sample(..., return_jacobian=True) given the condition vectors.
 * latent [sn,2] = SphereSampler set (flow.py:62-76); jitter [pn,sn] or NULL (training azimuth noise).
 * Outputs angles [pn,sn,2], logj [pn,sn]; bins [pn,sn,2] int32 or NULL (spline bin per block). */
int tf_flow_sample_fwd(const TfCouplingNet nets[2], const float* cond, const float* latent,
                       const float* jitter, int64_t pn, int32_t sn, float* angles, float* logj,
                       int32_t* bins, int32_t precision /* TfPrecision */, float* workspace,
                       size_t workspace_floats, tf_stream_t stream);

/* TensoFlow.forward(..., return_jacobian=True): x [m,2]; row r uses cond[rays_id[r]] or, when
 * rays_id is NULL, cond[r / sn].  Outputs z [m,2], logq [m]; bins [m,2] int32 or NULL. */
int tf_flow_logq_fwd(const TfCouplingNet nets[2], const float* cond, const float* x,
                     const int64_t* rays_id, int64_t m, int32_t sn, int64_t pn, float* z, float* logq,
                     int32_t* bins, int32_t precision /* TfPrecision */, float* workspace,
                     size_t workspace_floats, tf_stream_t stream);

/* ElementWisePWQuadraticTransform alone (network/flow.py:332-413 forward / density direction, :415-525 inverse / sampling
 * direction): wv [m,21] = (v_tilde[11], w_tilde[10]) rows, y [m] in (0,1) -> out [m], logj [m], bins [m] int32 or NULL.
 * Runs the same device functions the fused flow kernels call; exists so that the reference's spline vectors (edge rows:
 * y -> 0 / 1, equal knots, w_tilde = -12 / +6, all-zero rows) are exercised on the device. */
int tf_pwquad_eval(const float* wv, const float* y, int64_t m, int32_t inverse, float* out, float* logj,
                   int32_t* bins, tf_stream_t stream);

/* Backward of tf_flow_logq_fwd (the NIS-loss training direction, fields.py:1257-1333): gradient of
 * sum_r g_logq[r]*logq[r] wrt the 16 net tensors (accumulated with float atomics: zero them first) and wrt the
 * hoisted per-point layer-1 pre-activation, g_point [2,pn,64] (zero first).  The caller folds g_point into
 * d W1[:, 7:44] = g_point^T (2 cond - 1), d b1 = sum_pt g_point, d cond = 2 g_point W1[:, 7:44]  (three small GEMMs).
 * d W1[:, 0:7] (the sample-embedding columns) is written by the kernel into gnets[k].w[0] ([64,44], cols 0..6).
 * g_x [m,2] or NULL: gradient wrt the sample coordinates x themselves (asked for between nis_loss_iter and nis_start_iter, where the
 * NIS loss is fitted on the fixed GGX half angles and those depend on the predicted roughness: fields.py:1296-1318 with
 * sample_specular_directions :858-903) -- closed form through both splines and the kept coordinate's embedding.
 * z [m,2] or NULL: the z output of tf_flow_logq_fwd on the same inputs (round 5).  The reverse pass needs block 1's output z[:,0]
 * before it can re-evaluate block 0; handed the forward's own value it skips one of its three net evaluations per row, NULL
 * recomputes it (split f16 operands, as the forward's TF_PREC_F16X3).
 * g_cond [pn,37] or NULL: given, the call folds g_point itself (round 5) -- g_cond = 2 sum_k g_point[k] W1_k[:, 7:44] (overwritten),
 * gnets[k].w[0][:, 7:44] += g_point[k]^T (2 cond - 1), gnets[k].b[0] += sum_pt g_point[k] -- and the caller has nothing left to do. */
typedef struct TfCouplingNetGrad {
  float* w[4];
  float* b[4];
} TfCouplingNetGrad;
size_t tf_flow_bwd_workspace_floats(int64_t pn);
int tf_flow_logq_bwd(const TfCouplingNet nets[2], const float* cond, const float* x, const float* z, const int64_t* rays_id,
                     int64_t m, int32_t sn, int64_t pn, const float* g_logq, const TfCouplingNetGrad gnets[2],
                     float* g_point, float* g_cond, float* g_x, float* workspace, size_t workspace_floats, tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Environment light: EnvLight.direct_light (network/light.py:125-162) = exp(bilinear cube lookup
 * of the log-radiance cubemap base [6,R,R,3]) with seam-crossing taps.
 * ------------------------------------------------------------------------------------------ */
/* depth [m] or NULL: out is zeroed where !(depth > near_eps) (the near mask of get_lights, fields.py:973-974). */
int tf_cube_lookup_fwd(const float* base, int32_t res, const float* dirs, int64_t m, int32_t apply_exp,
                       const float* depth, float near_eps, float* out, tf_stream_t stream);
/* g_base += d out/d base (atomics; zero first); out = forward result (needed when apply_exp). */
int tf_cube_lookup_bwd(const float* base, int32_t res, const float* dirs, int64_t m, int32_t apply_exp,
                       const float* g_out, float* g_base, tf_stream_t stream);
/* Same, plus the gradient wrt the lookup direction (dr.texture is differentiable in its coordinates; the shape stage
 * reaches the SDF through envlight(normal) / envlight(reflective, roughness), network/fields.py:436-446, :419-439).
 * g_base [6,R,R,3] (+=, zero first) or NULL; g_dirs [m,3] (overwritten) or NULL; at least one of them. */
int tf_cube_lookup_bwd_dirs(const float* base, int32_t res, const float* dirs, int64_t m, int32_t apply_exp,
                            const float* g_out, float* g_base, float* g_dirs, tf_stream_t stream);
/* EnvLight.__call__ with a roughness (network/light.py:95-122: dr.texture(..., mip=..., mip_level_bias=..., 'linear-mipmap-linear',
 * boundary_mode='cube')): out [m,3] = (exp of) the bilinear cube fetch of levels floor(mip) and floor(mip) + 1 of the stack
 * texs[0..n_levels-1] ([6,res[l],res[l],3] each, n_levels <= 8), blended by the fraction of mip [m] (0 <= mip <= n_levels - 1).
 * _bwd: g_texs[l] += d out / d level l (float atomics; NULL: none), g_dirs [m,3] and g_mip [m] (either may be NULL).  Round 5: one
 * launch each way for what the shape stage composed from ~130 element-wise launches per training step. */
int tf_cube_lookup_mips_fwd(const float* const* texs, const int32_t* res, int32_t n_levels, const float* dirs, const float* mip,
                            int64_t m, int32_t apply_exp, float* out, tf_stream_t stream);
int tf_cube_lookup_mips_bwd(const float* const* texs, const int32_t* res, int32_t n_levels, const float* dirs, const float* mip,
                            int64_t m, int32_t apply_exp, const float* g_out, float* const* g_texs, float* g_dirs, float* g_mip,
                            tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sample generation for the march.
 * tf_alpha_mask_sample: AlphaGridMask.sample_alpha(pts) > 0 (network/shapeRenderer.py:78-97, :1120-1121) on a binary
 *   u8 volume [D,H,W] (W <- x) spanning aabb_host[6]; alive[i] in {0,1}, bit-exact with the reference mask.
 * tf_march_uniform: fixed-step sampler + occupancy culling + (ray, t)-ordered packing; stands in for
 *   nerfacc.OccGridEstimator.sampling (shapeRenderer.py:950-959; third-party, unpinned) and is BASELINE configs[1]'s
 *   "n uniform steps in the aabb slab" when step_size <= 0.  Ray/box slab test and near/far clamp as sample_ray
 *   (shapeRenderer.py:878-884).  Call twice: (1) offsets = NULL -> counts[rn] = live samples per ray;
 *   (2) offsets[rn] = exclusive prefix sum of counts -> t_starts/t_ends/ray_indices [sum counts] written densely.
 *   volume may be NULL (no occupancy culling).  occupancy_mode 0: `volume` is an AlphaGridMask volume [D,H,W] (W <- x), a sample
 *   lives when the trilinear fetch at its mid-point is > 0; 1: `volume` is an occupancy grid [rx = D, ry = H, rz = W]
 *   (OccGridEstimator.binaries), a sample lives when the CELL holding its mid-point is set.  t_jitter [rn] or NULL: added to each
 *   ray's start after the near/far clamp (stratified sampling: U[0,1) * step drawn by the caller).
 * ------------------------------------------------------------------------------------------ */
int tf_alpha_mask_sample(const uint8_t* volume, int32_t D, int32_t H, int32_t W, const float* aabb_host, const float* pts,
                         int64_t n, uint8_t* alive, tf_stream_t stream);
int tf_march_uniform(const float* rays_o, const float* rays_d, const float* near, const float* far, int64_t rn,
                     int32_t n_steps, float step_size, const float* aabb_host, const uint8_t* volume, int32_t D, int32_t H,
                     int32_t W, const float* mask_aabb_host, int32_t occupancy_mode, const float* t_jitter,
                     const int64_t* offsets, int64_t* counts, float* t_starts, float* t_ends, int64_t* ray_indices,
                     tf_stream_t stream);

/* ShapeRenderer.sample_ray (network/shapeRenderer.py:871-932) with upsample / cat_z_vals (:820-869) and sample_pdf
 * (utils/network_utils.py:117-147, det=True), three kernels around the field evaluations (tf_sdf_forward) of each round:
 * tf_sample_ray_init: aabb slab test clamped to [near, far] (:878-884), z [rn, n_samples] = tmin + (tmax - tmin) * lin[k]
 *   (+ t_rand[r] * 2 / n_samples, the per-ray stratification offset of perturb > 0, or NULL), the sample points [rn n_samples, 3]
 *   and their mip levels log2(ball_radius / base_radii) (:966-970).  lin [n_samples] = torch.linspace(0, 1, n_samples) on the device.
 * tf_sample_ray_upsample: one NeuS up-sampling round on the sorted rows z / sdf [rn, n_cur] (n_cur <= 128) with sharpness inv_s:
 *   new_t [rn, n_imp] (ascending), their points / levels (new_pts may be NULL: last round).  u_lin [n_imp] =
 *   torch.linspace(0.5 / n_imp, 1 - 0.5 / n_imp, n_imp).
 * tf_sample_ray_merge: z_out [rn, n_cur + n_imp] = the stable sort of cat(z, new_t) (old samples first on ties), sdf_out the
 *   gather of cat(sdf, new_sdf) in the same order (sdf_out may be NULL: last round). */
int tf_sample_ray_init(const float* rays_o, const float* rays_d, const float* near, const float* far, const float* radiis,
                       const float* rays_cos, const float* aabb_host, const float* lin, const float* t_rand, int64_t rn,
                       int32_t n_samples, float base_radii, float* z, float* pts, float* level, tf_stream_t stream);
int tf_sample_ray_upsample(const float* rays_o, const float* rays_d, const float* radiis, const float* rays_cos, const float* z,
                           const float* sdf, int64_t rn, int32_t n_cur, int32_t n_imp, float inv_s, const float* u_lin,
                           float base_radii, float* new_t, float* new_pts, float* new_level, tf_stream_t stream);
int tf_sample_ray_merge(const float* z, const float* sdf, const float* new_t, const float* new_sdf, int64_t rn, int32_t n_cur,
                        int32_t n_imp, float* z_out, float* sdf_out, tf_stream_t stream);
/* The element-wise algebra between the sampler and compute_sdf_alpha, one launch each (round 5: ~30 element-wise launches per shape
 * training step).  tf_sample_ray_intervals: sample_ray's tail (shapeRenderer.py:921-932) on the merged grid t [rn, n_samples] --
 * t0 = t, t1 = t + dist (dist to the next sample; the last one repeats its predecessor's), inner [rn n_samples] bytes = the interval's
 * midpoint o + d (t + dist / 2) lies inside aabb_host[6]; the caller compacts the rows with inner != 0.
 * tf_sample_points: render_core's prelude (:1118-1131) on the packed samples -- mid = (t0 + t1) / 2, dists = t1 - t0,
 * viewdir [n,3] = dirs[ray_indices], points [n,3] = rays_o[ray_indices] + viewdir mid, level [n] = log2(ball radius / base_radii)
 * (compute_ball_radii, :1038-1044). */
int tf_sample_ray_intervals(const float* rays_o, const float* dirs, const float* t, int64_t rn, int32_t n_samples,
                            const float* aabb_host, float* t0, float* t1, uint8_t* inner, tf_stream_t stream);
int tf_sample_points(const float* rays_o, const float* dirs, const float* radiis, const float* rays_cos, const int64_t* ray_indices,
                     const float* t0, const float* t1, int64_t n, float base_radii, float* mid, float* dists, float* viewdir,
                     float* points, float* level, tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Env-light prefilter: EnvLight.build_mips (network/light.py:52-64), rebuilt every shape-stage training step
 * (network/shapeRenderer.py:1291).  Replaces light_utils.cubemap_mip (network/light_utils.py:66-70) and the renderutils
 * plugin entry points diffuse_cubemap_fwd/bwd, specular_cubemap_fwd/bwd (network/renderutils/ops.py:391-458,
 * c_src/cubemap.cu:112-168,239-349; the specular_bounds table of cubemap.cu:178-237 is not needed -- the lobe cone is
 * culled in-kernel).  All maps are [6,res,res,3] float32.  The *_bwd entry points WRITE the full gradient (no atomics,
 * no need to zero first).
 * ------------------------------------------------------------------------------------------ */
int tf_cubemap_mip_fwd(const float* cube, int32_t res, float* out /*[6,res/2,res/2,3]*/, tf_stream_t stream);
int tf_cubemap_diffuse_fwd(const float* cube, int32_t res, float* out, tf_stream_t stream);
int tf_cubemap_diffuse_bwd(const float* g_out, int32_t res, float* g_cube, tf_stream_t stream);
/* cos_cutoff = cosine of the lobe half-angle keeping `cutoff` of the GGX NDF mass (ops.py:428-441, computed by the host).
 * out = sum w*cube / sum w; wsum [6,res,res] (may be NULL) receives sum w for the backward.
 * texel_table (round 5; NULL: derived per pair in the kernel, same values): [6,res,res,4] floats from tf_cubemap_texel_table --
 * (unit direction, area) of every texel as c_src/cubemap.cu:17-46 defines them; a function of res alone, built once. */
int tf_cubemap_texel_table(int32_t res, float* table /* 16-byte aligned */, tf_stream_t stream);
int tf_cubemap_specular_fwd(const float* cube, int32_t res, float roughness, float cos_cutoff, float* out, float* wsum,
                            const float* texel_table, tf_stream_t stream);
int tf_cubemap_specular_bwd(const float* g_out, const float* wsum, int32_t res, float roughness, float cos_cutoff,
                            float* g_cube, const float* texel_table, tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * First-hit ray/mesh intersection: raytracing.RayTracer.trace (raytracing/raytracer.py:19-54) +
 * MaterialRenderer.trace post-processing (network/materialRenderer.py:253-263).
 * The BVH is built on the host (tf_bvh_build_host) and uploaded by the caller.
 * ------------------------------------------------------------------------------------------ */
typedef struct TfBvhNode { /* 32 bytes */
  float lo[3];
  int32_t left;  /* inner: index of left child (right = left+1); leaf: first triangle */
  float hi[3];
  int32_t count; /* 0 = inner node, >0 = leaf with `count` triangles */
} TfBvhNode;

/* verts_host [nv,3], faces_host [nf,3] -> nodes_host (capacity 2*nf), tris_host [nf,9] reordered
 * triangle soup (a,b,c); returns number of nodes (>0) or a negative TfStatus. */
int64_t tf_bvh_build_host(const float* verts_host, int64_t nv, const int32_t* faces_host, int64_t nf,
                          TfBvhNode* nodes_host, float* tris_host);

/* Traversal layout for the device (tf_bvh_trace does not walk TfBvhNode): `pairs_host` receives one 32-byte record per
 * inner node -- both child boxes quantised to 16 bits per coordinate on one global grid (rounded outward) + both child
 * references, depth-first order -- capacity (n_nodes/2 + 1) x 8 uint32; `frame_host` [6] receives the grid (origin xyz,
 * scale xyz: coordinate = origin + q * scale); `tris12_host` [nf,12] receives (a, e1 = b - a, e2 = c - a, 0 0 0) per
 * triangle of the reordered soup (full precision: hits and depths are exact).  Child reference: >= 0 pair index,
 * < -1 leaf = ~((first_triangle << 3) | count), -1 none.  Returns the number of pairs (> 0) or a negative TfStatus. */
/* dwords per record of the traversal layout tf_bvh_pack_host writes (a build-time property of the library: 8 = child pairs,
 * 16 = 4-wide nodes); the caller sizes `pairs_host` as [n_nodes / 2 + 1, tf_bvh_record_dwords()]. */
int32_t tf_bvh_record_dwords(void);
int64_t tf_bvh_pack_host(const TfBvhNode* nodes_host, int64_t n_nodes, const float* tris_host, int64_t nf,
                         uint32_t* pairs_host, float* tris12_host, float* frame_host);

/* Replaces raytracing.RayTracer.trace + MaterialRenderer.trace (network/materialRenderer.py:221-223,253-263).
 * pairs [n_pairs,8] uint32, tris12 [nf,12]: device copies of tf_bvh_pack_host's output (16-byte aligned); frame_host [6]:
 * its quantisation grid (host memory).
 * o [m / rays_per_origin, 3], d [m,3] -> pos [m,3] (= origin + depth*d), nrm [m,3] (= normalize(-face_normal), 0 on a
 * miss), depth [m] (10.0 on a miss), hit [m] uint8 (depth < 10); pos/nrm/hit may be NULL.
 * rays_per_origin: ray i starts at row i / rays_per_origin of o (1 = one origin row per ray; T = the T secondary rays of
 * one surface point share its row, so the caller need not materialise pts[:,None].expand(pn,T,3), fields.py:1188).
 * slot_order [rays_per_origin] int32 (device) or NULL: traversal ORDER of the rays of one point -- the j-th ray traced is
 * slot slot_order[j]; results are still written at the ray's own index.  The integral's direction sets are Fibonacci
 * spirals (consecutive slots point ~222 degrees apart); sorting the slots along a space-filling curve makes the 64 rays
 * of a wavefront walk the same part of the tree.
 * origin = (o + d*origin_offset0) + origin_offset1*d, the two roundings of the reference's
 * `p + 1e-5 d` (fields.py:955) followed by `o + 2*unit_size*d` (materialRenderer.py:223); pass 0,0 for
 * a plain trace.
 * live [m] uint8 or NULL: rays with live == 0 are not traversed and reported as misses (rays whose weight in the
 * integral is exactly zero -- below-horizon samples, fields.py:1156,1209 -- per-wavefront live-sample culling).
 * hit_rows_only != 0: pos / nrm rows are written for rays that hit only (rows of missing rays are left untouched): the
 * integral reads them through the compacted hit list, and 85 % of its rays miss.
 * origin_order [m / rays_per_origin] int32 (device) or NULL (used when rays_per_origin >= 64): the ORDER in which origins are handed
 * to the persistent waves -- the u-th origin traced is origin_order[u]; results stay at each ray's own index.  Passing a
 * spatially sorted order (Morton code of the origin) lets every XCD work on one contiguous eighth of the scene (its L2 then
 * holds that part of the tree and triangles).
 * work_counter: 64 bytes (8 x int64) of device scratch (zeroed by the call) that switches on the persistent kernel with
 * dynamic ray fetch (lanes pull new rays as their wave-mates finish); NULL = one statically assigned ray per lane. */
int tf_bvh_trace(const uint32_t* pairs, const float* tris12, const float* frame_host, int64_t n_pairs, const float* o, const float* d,
                 int64_t rays_per_origin, const int32_t* slot_order, float origin_offset0, float origin_offset1,
                 const uint8_t* live, int64_t m,
                 float* pos, float* nrm, float* depth, uint8_t* hit, int32_t hit_rows_only, const int32_t* origin_order,
                 int64_t* work_counter, tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Generic small MLP on rows (weight-norm already folded by the caller): used for the inner-light
 * net of MCShadingNetwork.get_inner_lights (network/fields.py:905-911: pos_enc8 + IDE5 -> 123 ->
 * 256 -> 256 -> 256 -> 3, exp(min(x, exp_max))).
 * ------------------------------------------------------------------------------------------ */
typedef struct TfMlp4 {
  const float* w[4]; /* [256,123] [256,256] [256,256] [3,256] */
  const float* b[4];
} TfMlp4;
size_t tf_inner_light_workspace_floats(void);
/* pts/view/nrm [m,3] (hit position, direction back along the ray = -d, surface normal) -> out [m,3]. */
int tf_inner_light_fwd(const TfMlp4* net, const float* pts, const float* view, const float* nrm,
                       int64_t m, float exp_max, int32_t precision, float* out, float* workspace,
                       size_t workspace_floats, tf_stream_t stream);

/* Device-side form used inside get_lights (fields.py:970-974): rows are the rays listed in idx[0 .. *count_dev)
 * (built by tf_compact_mask from the hit flags; at most `capacity` rows), view = -dirs[i], and the result is
 * scattered: lights[i] = light * (depth[i] > near_eps).  No host synchronisation between trace and shading. */
int tf_inner_light_indexed_fwd(const TfMlp4* net, const float* pos, const float* dirs, const float* nrm,
                               const int64_t* idx, const int64_t* count_dev, int64_t capacity, const float* depth,
                               float near_eps, float exp_max, int32_t precision, float* lights, float* workspace,
                               size_t workspace_floats, tf_stream_t stream);
/* The same in a TRAINING step (what autograd keeps of get_inner_lights, fields.py:905-911, for the backward pass of its four Linear
 * layers): besides `lights`, the post-ReLU activations of the three hidden layers, acts[3][capacity][256] fp32, row r = the ray
 * idx[r] (rows >= *count_dev are not written).  tf_linear_bwd reads them as the saved inputs / outputs of the layers instead of
 * recomputing the net with the dense-layer kernels.  precision: TF_PREC_F16X3 (| TF_WEIGHTS_PACKED) only. */
int tf_inner_light_indexed_train_fwd(const TfMlp4* net, const float* pos, const float* dirs, const float* nrm,
                                     const int64_t* idx, const int64_t* count_dev, int64_t capacity, const float* depth,
                                     float near_eps, float exp_max, int32_t precision, float* lights, float* acts,
                                     float* workspace, size_t workspace_floats, tf_stream_t stream);
/* MCShadingNetwork.predict_outer_lights with outer_light_version = 'direction' (network/fields.py:913-916, the net built at
 * :716-718: make_predictor_4layer(72, 3, 'exp', light_exp_max); configs/mat/syn/{lego,armadillo,horse}.yaml) for the rays that MISSED
 * the mesh (get_lights :962-968): lights[i] = exp(min(net(IDE5(dirs[i], roughness 0)), exp_max)) for i = idx[r], r < *count_dev.
 * net->w[0] is the [256,72] first layer; the other layers as TfMlp4.  The direction rows are encoded as they are (the reference
 * does not normalise them).  precision: TF_PREC_F16X3 or TF_PREC_F16X2 (| TF_WEIGHTS_PACKED) -- the net runs on the staggered kernel of
 * the inner light (its 64-ray / 128-ray form); `workspace` as tf_inner_light_workspace_floats(), one workspace PER NET.  A missing ray's depth is TF_MISS_DEPTH, so
 * get_lights' near mask (:973) is 1 on every row written here. */
int tf_outer_light_indexed_fwd(const TfMlp4* net, const float* dirs, const int64_t* idx, const int64_t* count_dev,
                               int64_t capacity, float exp_max, int32_t precision, float* lights, float* workspace,
                               size_t workspace_floats, tf_stream_t stream);
/* Input encoding of the inner-light net alone: X [capacity,123] = cat[pos_enc8(pos[i]), IDE5(reflect(-dirs[i], nrm[i]))] for
 * i = idx[r] (r < *count_dev), or i = r when idx is NULL (then view = dirs).  Used by the training backward, whose
 * weight-gradient products are plain library GEMMs on X. */
int tf_inner_light_encode(const float* pos, const float* dirs, const float* nrm, const int64_t* idx,
                          const int64_t* count_dev, int64_t capacity, float* X, int32_t ld /* row stride of X in floats, >= 123:
                          columns [123, ld) are written as zeros (ld = 128: rows aligned for the dense-layer kernels' DMA path) */,
                          float* workspace, size_t workspace_floats, tf_stream_t stream);
/* idx[0 .. *count) = indices i with mask[i] != 0 (unordered); *count is zeroed by the call (replaces the boolean-mask
 * indexing of fields.py:962-971). */
int tf_compact_mask(const uint8_t* mask, int64_t m, int64_t* idx, int64_t* count, tf_stream_t stream);
/* The same with the mask v[i] < thr of a float array: the hit list straight from tf_bvh_trace's depth (a miss holds TF_MISS_DEPTH),
 * so that the traversal need not store a flag byte per ray (hit == NULL there). */
#define TF_MISS_DEPTH 10.0f
int tf_compact_below(const float* v, float thr, int64_t m, int64_t* idx, int64_t* count, tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Split-sum shading of the shape stage in ONE launch: ShapeShadingNetwork.forward (network/fields.py:448-567,
 * predict_specular_lights :419-439) + EnvLight.__call__ (network/light.py:72-80, :95-122) over the pre-filtered stack of
 * EnvLight.build_mips (tf_cubemap_*).  Per live march sample: mat_mlp 128-128-128-5 -> albedo / roughness / metallic,
 * diffuse + two-mip specular cube lookups, inner_light [pos_enc8 51, IDE5 72] 123-128-128-3, inner_weight
 * [pos_enc8 51, pos_enc6(refl) 39] 90-128-128-1, FG LUT, sRGB.  Decoders run in f16x3 (TF_PREC_F16X3 arithmetic).
 * tf_shape_shade_pack: weights in torch layout (weight-norm folded by the caller) -> fragment order in `workspace`
 *   (tf_shape_shade_workspace_floats() floats, caller-owned); once per weight update.
 * tf_shape_shade_fwd: spec_mips [n_spec] device pointers to [6,R_i,R_i,3] log-radiance maps (host array of pointers),
 *   spec_res [n_spec] (host), diffuse_map [6,Rd,Rd,3], fg_lut [H,W,2]; pts / normals / view [n,3] (normals, view need not
 *   be normalised), feat [n,128] -> color [n,3] (sRGB, clamped), occ [n] (unclamped occlusion probability), roughness [n],
 *   refl [n,3]; occ / roughness / refl may be NULL.
 * ------------------------------------------------------------------------------------------ */
typedef struct TfMlp3 {
  const float* w[3];
  const float* b[3];
} TfMlp3;
typedef struct TfShapeNets {
  TfMlp3 mat_mlp;      /* [128,128] [128,128] [5,128] */
  TfMlp3 inner_light;  /* [128,123] [128,128] [3,128] */
  TfMlp3 inner_weight; /* [128,90]  [128,128] [1,128] */
} TfShapeNets;
size_t tf_shape_shade_workspace_floats(void);
int tf_shape_shade_pack(const TfShapeNets* nets, float* workspace, size_t workspace_floats, tf_stream_t stream);
int tf_shape_shade_fwd(const float* workspace, const float* const* spec_mips, const int32_t* spec_res, int32_t n_spec,
                       const float* diffuse_map, int32_t diffuse_res, const float* fg_lut, int32_t fg_h, int32_t fg_w,
                       float min_roughness, float max_roughness, float light_exp_max, const float* pts, const float* normals,
                       const float* view, const float* feat, int64_t n, float* color, float* occ, float* roughness,
                       float* refl, tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-surface-point preparation of the rendering integral in ONE launch (the reference runs ~30):
 *   MCShadingNetwork.tenso_feature + predict_materials (network/fields.py:776-810, :1010-1017):
 *     VM gather (C = 36, level 0) -> weight-norm MLPs 108-128-{1,1,3} (ReLU, sigmoid),
 *     roughness = r * (1 - rough_min^2) + rough_min^2;
 *   TensoFlow.tenso_feature of the diffuse and specular flows (network/flow.py:709-744) and the condition row
 *     cat[feat16, embed3(view_angles) 14, 0 * embed3(rough) 7] of TensoFlow.sample / .forward (:836-848, :803-815).
 * tf_point_pack: weights in torch layout (weight-norm folded by the caller) -> MFMA fragment order in `workspace`
 *   (tf_point_workspace_floats() floats, caller-owned); call once per weight update.
 * tf_point_fwd: pts [pn,3], view_angles [pn,2] (tf_view_angles) -> metallic [pn], roughness [pn], albedo [pn,3],
 *   cond_d / cond_s [pn,37].  All three fields share `aabb_host` [2,3].
 * ------------------------------------------------------------------------------------------ */
typedef struct TfPointNets {
  const float* mat_w1[3]; /* metallic, roughness, albedo: [128,108] */
  const float* mat_b1[3]; /* [128] */
  const float* mat_w2[3]; /* [1,128] [1,128] [3,128] */
  const float* mat_b2[3];
  const float* nis_w1[2]; /* diffuse, specular flow: nis_mat.0 [64,57] */
  const float* nis_b1[2];
  const float* nis_w2[2]; /* nis_mat.2 [16,64] */
  const float* nis_b2[2];
} TfPointNets;
size_t tf_point_workspace_floats(void);
int tf_point_pack(const TfPointNets* nets, float* workspace, size_t workspace_floats, tf_stream_t stream);
int tf_point_fwd(const float* workspace, const TfVmDesc* mat_desc, const float* mat_packed, const TfVmDesc* flow_d_desc,
                 const float* flow_d_packed, const TfVmDesc* flow_s_desc, const float* flow_s_packed,
                 const float* aabb_host, const float* pts, const float* view_angles, int64_t pn, float rough_min,
                 float* metallic, float* roughness, float* albedo, float* cond_d, float* cond_s, tf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Monte-Carlo shading integral, MCShadingNetwork.shade_mixed (network/fields.py:1075-1235),
 * eval path with the flow samplers active (use_half_* = True, human lights off).
 *
 * tf_shade_dirs: per point builds the tangent frame (fields.py:812-822), turns the flow samples
 *   (half-vector angles in [0,1]^2 + logq) and the fixed cosine set (fields.py:824-856) into
 *   outgoing directions, pdfs and per-direction BRDF weights.  Slot layout per point, T = sd+nf+ss:
 *     [0,sd) flow diffuse | [sd,sd+nf) fixed diffuse | [sd+nf,T) flow specular.
 *   dirs [pn,T,3]; wgt [pn,T,3] = (BRDF weight / max(pdf,1e-6)) / count  (0 for masked specular);
 *   spec_mask [pn,ss] uint8 = dot(dir, n) > 0 (fields.py:1209).
 * tf_shade_reduce: colors [pn,3] = linear_to_srgb(sum_t wgt*light) (fields.py:1230-1231) plus the
 *   linear diffuse / specular sums.
 * ------------------------------------------------------------------------------------------ */
int tf_view_angles(const float* normals, const float* view, int64_t pn, float* view_angles,
                   tf_stream_t stream);
/* fixed_d [nf,2] = (azimuth/2pi, 1-2*elevation/pi) of the Fibonacci set (fields.py:734-737);
 * az_jitter [pn] in [0,1) = the training-mode random azimuth (fields.py:837-838) or NULL. */
int tf_shade_dirs(const float* normals, const float* view, const float* metallic, const float* roughness,
                  const float* albedo, const float* ang_d, const float* logq_d, int32_t sd,
                  const float* fixed_d, const float* az_jitter, int32_t nf, const float* ang_s,
                  const float* logq_s, int32_t ss, int64_t pn, float* dirs, float* wgt, uint8_t* spec_mask,
                  uint8_t* live /* [pn,T] = any(wgt != 0), may be NULL */,
                  float* flow_logjac /* [pn, sd+ss] = log max(4 pi^2 HoV sin(theta), 1e-6) of the flow slots (NIS loss,
                                        fields.py:1275,1312), may be NULL */,
                  const int32_t* slot_of_pos /* [T] device permutation or NULL: row j of a point's dirs / wgt / live holds slot
                                                slot_of_pos[j] -- the rays of a point stored in TRAVERSAL order, so that tf_bvh_trace
                                                (called without slot_order) reads and writes consecutive rows from consecutive
                                                lanes; spec_mask / flow_logjac stay indexed by sample.  NULL: row = slot */,
                  int32_t row_begin, int32_t row_count /* only rows [row_begin, row_begin + row_count) of every point are built
                                                          (row_count < 0: all T): the rows of a direction set can be built as soon
                                                          as ITS samples exist, while the next set is still being sampled; the
                                                          sample arrays of sets outside the range are not read */,
                  tf_stream_t stream);
/* tf_shade_dirs with the flows sampling the OUTGOING direction instead of the half vector (cfg use_half_diffuse / use_half_specular = False,
 * network/fields.py:1117-1134, :1190-1203): whole_mask bit 0 = the diffuse lobe's flow samples, bit 1 = the specular lobe's are directions
 * (phi, theta) in the normal's frame; their density is exp(-clamp(logq)) / max(pi^2 sin(theta), 1e-6) and flow_logjac the log of that
 * denominator (:1279, :1317).  Bit 2 (4, also accepted by the two *_mode entry points below): cfg geometry_type = 'ggx_smith' -- the
 * specular weight's geometry term is geometry_ggx_smith_correlated, 1 / (1 + L(NoV) + L(NoL)) with L(c) = (sqrt(1 + a^2 tan^2) - 1) / 2
 * (fields.py:1000-1008, :1029), instead of the Schlick-GGX product (:987-998).  whole_mask = 0 is tf_shade_dirs. */
int tf_shade_dirs_whole(const float* normals, const float* view, const float* metallic, const float* roughness, const float* albedo,
                        const float* ang_d, const float* logq_d, int32_t sd, const float* fixed_d, const float* az_jitter, int32_t nf,
                        const float* ang_s, const float* logq_s, int32_t ss, int64_t pn, float* dirs, float* wgt, uint8_t* spec_mask,
                        uint8_t* live, float* flow_logjac, const int32_t* slot_of_pos, int32_t row_begin, int32_t row_count,
                        int32_t whole_mask, tf_stream_t stream);
/* The sampler of the NON-NIS pass of shade_mixed (nis_sample False / flows not yet active; the pass whose colours
 * MCShadingNetwork.forward returns in eval, fields.py:1467-1473): nf fixed cosine directions (sample_diffuse_directions,
 * fields.py:824-856) followed by ss fixed specular directions -- the GGX half-vector warp of the Fibonacci samples fixed_s [ss,2]
 * by the point's (squared) roughness with pdf D NoH / (4 VoH) x the elevation Jacobian (sample_specular_directions,
 * fields.py:858-903).  Slot layout [0,nf) diffuse | [nf,nf+ss) specular; outputs as tf_shade_dirs.  az_jitter / az_jitter_s [pn]
 * = the training-mode random azimuths (random_azimuth) or NULL. */
int tf_shade_dirs_fixed(const float* normals, const float* view, const float* metallic, const float* roughness,
                        const float* albedo, const float* fixed_d, const float* az_jitter, int32_t nf,
                        const float* fixed_s, const float* az_jitter_s, int32_t ss, int64_t pn, float* dirs, float* wgt,
                        uint8_t* spec_mask, uint8_t* live, tf_stream_t stream);
/* tf_shade_dirs_fixed under cfg geometry_type: mode 0 = 'schlick' (tf_shade_dirs_fixed), 4 = 'ggx_smith' (see tf_shade_dirs_whole). */
int tf_shade_dirs_fixed_mode(const float* normals, const float* view, const float* metallic, const float* roughness,
                             const float* albedo, const float* fixed_d, const float* az_jitter, int32_t nf,
                             const float* fixed_s, const float* az_jitter_s, int32_t ss, int64_t pn, float* dirs, float* wgt,
                             uint8_t* spec_mask, uint8_t* live, int32_t mode, tf_stream_t stream);
/* Backward of the BRDF weights wrt the per-point materials (training): g_wgt [pn,T,3] ->
 * g_albedo [pn,3], g_metallic [pn], g_roughness [pn] (overwritten).  dirs / wgt are tf_shade_dirs' outputs. */
int tf_shade_dirs_bwd(const float* normals, const float* view, const float* metallic, const float* roughness,
                      const float* albedo, const float* dirs, const float* wgt, const float* g_wgt, int32_t sd,
                      int32_t nf, int32_t ss, int64_t pn, float* g_albedo, float* g_metallic, float* g_roughness,
                      tf_stream_t stream);
/* ... under cfg geometry_type: mode 0 = 'schlick' (tf_shade_dirs_bwd), 4 = 'ggx_smith' (d log G / d roughness of the Smith term). */
int tf_shade_dirs_bwd_mode(const float* normals, const float* view, const float* metallic, const float* roughness,
                           const float* albedo, const float* dirs, const float* wgt, const float* g_wgt, int32_t sd,
                           int32_t nf, int32_t ss, int64_t pn, float* g_albedo, float* g_metallic, float* g_roughness,
                           int32_t mode, tf_stream_t stream);
/* Same reduction with get_lights' miss branch folded in (fields.py:951-975): a slot whose ray hit the mesh (hit[r] != 0)
 * takes hit_lights[r] (tf_inner_light_indexed_fwd's scatter target; other rows are never read), a slot whose ray missed
 * takes exp(cube(env_base, dirs[r])) * (depth[r] > near_eps) evaluated on the fly -- the [pn,T,3] light array of the
 * miss branch is never written or read.  Zero-weight slots are skipped.  hit may be NULL: a ray then hit iff depth[r] < TF_MISS_DEPTH.
 * env_base may be NULL (outer_light_version = 'direction'): then every slot with a non-zero weight takes hit_lights[r] -- the rows of
 * the rays that missed were written by tf_outer_light_indexed_fwd. */
int tf_shade_reduce_env(const float* wgt, const float* dirs, const float* depth, const uint8_t* hit, const float* hit_lights,
                        const float* env_base, int32_t env_res, float near_eps, int64_t pn, int32_t n_diffuse, int32_t ss,
                        float* colors, float* diffuse_lin, float* specular_lin,
                        const int32_t* slot_of_pos /* as tf_shade_dirs: the lobe of row j is that of slot slot_of_pos[j]; or NULL */,
                        tf_stream_t stream);
/* The same reduction plus the per-point statistics behind the REST of shade_mixed's output dict (fields.py:1232-1256, :1288-1291):
 * aux [pn,16] = { sum of the diffuse rays' lights [3] (-> diffuse_light :1242, approximate_light :1248), sum of the unmasked specular
 * rays' lights [3] (-> specular_light :1230/:1243), the same over rays that hit the mesh [3] and their count (-> indirect_light :1229,
 * visibility :1228), then count / mean / sum of squared deviations of g = mean_c(fx_c) / max(p, 1e-6) over the unmasked specular rays
 * (-> variance :1289, variance_specular_vis :1291) and mean / sum of squared deviations of g over the n_diffuse diffuse rays
 * (-> variance_diffuse_vis :1256), one pad }.  EVERY ray's light enters (zero weight or not): call it on rays traced without the
 * zero-weight culling.  `lights` [pn,T,3] non-NULL: the light of every ray is given (training composition) and hit / depth only
 * supply the hit flags; NULL: as tf_shade_reduce_env.  spec_mask [pn,ss] = tf_shade_dirs' mask.  colors / diffuse_lin /
 * specular_lin may be NULL. */
int tf_shade_reduce_aux(const float* wgt, const float* lights, const float* dirs, const float* depth, const uint8_t* hit,
                        const float* hit_lights, const float* env_base, int32_t env_res, float near_eps, const uint8_t* spec_mask,
                        int64_t pn, int32_t n_diffuse, int32_t ss, float* colors, float* diffuse_lin, float* specular_lin,
                        float* aux, const int32_t* slot_of_pos, tf_stream_t stream);
/* n_diffuse = sd + nf.  diffuse_lin / specular_lin [pn,3] may be NULL. */
int tf_shade_reduce(const float* wgt, const float* lights, int64_t pn, int32_t n_diffuse, int32_t ss,
                    float* colors, float* diffuse_lin, float* specular_lin, tf_stream_t stream);

/* -----------------------------------------------------------------------------------
 * The element-wise algebra of ShapeShadingNetwork.forward in the TRAINING direction (network/fields.py:448-567) as two differentiable
 * stages around the nets / cube lookups / encodings (which are entry points of their own), forward and adjoint: 4 launches instead of
 * the ~150 element-wise launches of the torch composition and its autograd mirror image.
 *   pre  (:455-463): normals [n,3], view [n,3] (any length), mat [n,5] (mat_mlp's sigmoid outputs) -> normals_u (F.normalize + the
 *        reference's patch of rows with n.x + n.y == 0), view_u, nov [n] = n . v, reflective [n,3] = 2 nov n - v, roughness [n] = 0.9 m3 + 0.09
 *        and, with mip non-NULL, mip [n] = clamp(EnvLight.get_mip(roughness), 0, n_levels - 1) (network/light.py:72-80, :101: the
 *        coordinate in the n_levels-deep specular stack, two linear pieces meeting at max_roughness).
 *   post (:460-561): mat, nov, diffuse_light / direct_light / indirect_light [n,3], occ_raw [n] (inner_weight's output), FG LUT
 *        [fg_h,fg_w,2] -> color [n,3] = clamp(sRGB((1 - m) a L_d + ((0.04 (1 - m) + m a) FG.x + FG.y) (L_i occ + L_s (1 - occ))), 0, 1) with
 *        a = 0.77 m012 + 0.03, occ = clamp(occ_prob, 0, 1), occ_prob [n] = 0.5 occ_raw + 0.5, FG = F.grid_sample(bilinear, border,
 *        align_corners=False) at (clamp(nov, 0, 1), clamp(roughness, 0, 1)).
 * The adjoints follow torch autograd's conventions for that composition (clamp masks inclusive, zero LUT gradient on clipped
 * coordinates, the sRGB branch taken).  pre_bwd: g_normals_u / g_nov / g_reflective / g_roughness / g_mip may be NULL (no gradient arrived; g_mip needs mat);
 * writes g_normals [n,3] and g_mat [n,5] (column 3, the others zero).  post_bwd: g_occ_prob may be NULL; writes every g_* given. */
int tf_shape_glue_pre_fwd(const float* normals, const float* view, const float* mat, int64_t n, float* normals_u, float* view_u,
                          float* nov, float* reflective, float* roughness, float* mip, float min_roughness, float max_roughness,
                          int32_t n_levels, tf_stream_t stream);
int tf_shape_glue_pre_bwd(const float* normals, const float* view, const float* g_normals_u, const float* g_nov, const float* g_reflective,
                          const float* g_roughness, const float* g_mip, const float* mat, float min_roughness, float max_roughness,
                          int32_t n_levels, int64_t n, float* g_normals, float* g_mat, tf_stream_t stream);
int tf_shape_glue_post_fwd(const float* mat, const float* nov, const float* diffuse_light, const float* direct_light,
                           const float* indirect_light, const float* occ_raw, const float* fg_lut, int32_t fg_h, int32_t fg_w, int64_t n,
                           float* color, float* occ_prob, tf_stream_t stream);
int tf_shape_glue_post_bwd(const float* mat, const float* nov, const float* diffuse_light, const float* direct_light,
                           const float* indirect_light, const float* occ_raw, const float* fg_lut, int32_t fg_h, int32_t fg_w,
                           const float* g_color, const float* g_occ_prob, int64_t n, float* g_mat, float* g_nov, float* g_diffuse_light,
                           float* g_direct_light, float* g_indirect_light, float* g_occ_raw, tf_stream_t stream);

/* F.normalize(x, dim=-1) of [n,3] rows in the training direction of render_core (network/shapeRenderer.py:1137 normals of the samples,
 * :1145 the eikonal residual err [n] = (|x| - 1)^2 of the same rows, :1207-1208 the composited ray normal F.normalize(v acc + (1 - acc) c)):
 * one launch each way.  acc [n] non-NULL (with blend_c, 3 HOST floats): the rows are x acc + (1 - acc) c.  err may be NULL.  The adjoint takes
 * g_y [n,3] and / or g_err [n] (either may be NULL) and writes g_x [n,3] (and g_acc [n] with acc); conventions as torch autograd
 * (x / eps below eps = 1e-12, zero gradient of the norm at x = 0). */
int tf_normalize3_fwd(const float* x, const float* acc, const float* blend_c, int64_t n, float* y, float* err, tf_stream_t stream);
int tf_normalize3_bwd(const float* x, const float* acc, const float* blend_c, const float* g_y, const float* g_err, int64_t n, float* g_x,
                      float* g_acc, tf_stream_t stream);

/* -----------------------------------------------------------------------------------
 * Dense layers of the training direction.  Replace torch.nn.Linear + activation -- the library GEMMs under the reference's
 * TensoSDF decoder (network/fields.py:78-81), make_predictor_3layer / _4layer (network/other_field.py:50-119: material
 * predictors fields.py:1010-1017, inner-light net :905-911, ShapeShadingNetwork's nets :448-567) -- in forward AND backward.
 * precision: TF_PREC_F32 = the exact-fp32 instruction v_mfma_f32_32x32x2_f32, bitwise an fp32 fma chain (the yardstick of the parity
 * tests); TF_PREC_BF16X3 (what the training ops pass) = fp32-grade products with fp32's operand range: on the 16-byte-aligned shapes a
 * bf16 TRIPLE split -- x = hi + mid + lo, six v_mfma_f32_32x32x16_bf16 per 16-deep product, every term down to 2^-24 |a||b|, fp32
 * accumulate (3-7e-7 of the largest element against fp64, as the exact instruction; operands must be FINITE and below 3.3895e38 in magnitude, the largest
 * bf16: the outputs an Inf / NaN / larger operand reaches come out NaN, not Inf -- IEEE Inf propagation is TF_PREC_F32's) -- and the exact
 * instruction on the other shapes; TF_PREC_F16X3 = f16 operand
 * split (three v_mfma_f32_32x32x16_f16 per product term, fp32 accumulate) for operands inside the f16 range only -- unscaled
 * gradients of a mean-reduced loss are not.
 * X [n,K], W [N,K] (torch layout), b [N] or NULL, Y [n,N] row-major.
 * tf_linear_bwd: Y = the forward OUTPUT (post-activation), gY [n,N]; gZ [n,N] scratch that receives gY * act'(Y);
 * gX [n,K] or NULL; gW [N,K] / gb [N] or NULL are overwritten (zeroed, then accumulated with fp32 atomics over row slabs).
 * n_dev (device pointer, or NULL): only the first min(n, *n_dev) rows are valid -- the row count of a compacted list (hit rays)
 * stays on the device, the launch is sized for the capacity n and surplus workgroups exit (no host sync in a training step). */
int tf_linear_fwd(const float* X, const float* W, const float* b, int64_t n, int32_t K, int32_t N, int32_t act /* TfActivation */,
                  float act_param, int32_t precision /* TfPrecision */, float* Y, const int64_t* n_dev, tf_stream_t stream);
int tf_linear_bwd(const float* X, const float* W, const float* Y, const float* gY, int64_t n, int32_t K, int32_t N, int32_t act,
                  float act_param, int32_t precision, float* gZ, float* gX, float* gW, float* gb, const int64_t* n_dev,
                  tf_stream_t stream);
/* One layer of a backward CHAIN through stacked layers (what autograd does to make_predictor_4layer, network/other_field.py:86-119, one
 * Linear + activation node after the other): tf_linear_bwd plus, in the same launches, the activation backward of the layer BELOW.
 * x_act / x_act_param: the activation that produced X (TF_ACT_NONE: X is no layer's output -- plain gX).  Then gX [n,K] receives
 * (gZ . W) * x_act'(X), the gradient wrt the PRE-activation of the layer below, and gbx [K] (or NULL) its column sums = that layer's
 * bias gradient.  gy_is_gz bit 0: gY already is this layer's pre-activation gradient (the previous call's gX): no activation pass, Y and
 * gZ are not read, gb must be NULL (it was the previous call's gbx).  gy_is_gz bit 1 (TF_BWD_GRADS_ZEROED = 2): the caller has zeroed gW /
 * gb / gbx (they are accumulated into with atomics) -- a chain zeroes ONE flat buffer for all its layers instead of three fills per layer. */
#define TF_BWD_GRADS_ZEROED 2
int tf_linear_bwd_fused(const float* X, const float* W, const float* Y, const float* gY, int64_t n, int32_t K, int32_t N, int32_t act,
                        float act_param, int32_t gy_is_gz, int32_t x_act, float x_act_param, int32_t precision, float* gZ, float* gX,
                        float* gW, float* gb, float* gbx, const int64_t* n_dev, tf_stream_t stream);

/* -----------------------------------------------------------------------------------
 * Encodings of the TRAINING direction (the inference kernels evaluate the same functions in registers).
 * tf_ide5_fwd: generate_ide_fn(5) (utils/ref_utils.py:53-117) -> out [n,72] = [Re(36) | Im(36)] of (x + i y)^m P_{l,m}(z)
 *   exp(-l (l + 1) / 2 kappa_inv) over (l, m) = (1, 0..1), (2, 0..2), (4, 0..4), (8, 0..8), (16, 0..16); xyz [n,3], kappa_inv [n] or
 *   NULL (= 0); coef [17,36] = the reference's fp32 coefficient table (coefficient of z^k in column (l, m)), supplied by the caller.
 *   Polynomials in fp64 Horner form on that table, results rounded to fp32.
 * tf_ide5_bwd: g_xyz [n,3] and g_kappa [n] (may be NULL) from g_out [n,72], closed form.
 * tf_posenc_fwd: get_embedder (utils/network_utils.py:38-50): out [n, d (1 + 2 n_freq)] = [x, sin(2^k x), cos(2^k x)]_k.
 * ----------------------------------------------------------------------------------- */
int tf_ide5_fwd(const float* xyz, const float* kappa_inv, const float* coef, int64_t n, float* out, tf_stream_t stream);
int tf_ide5_bwd(const float* xyz, const float* kappa_inv, const float* coef, const float* g_out, int64_t n, float* g_xyz,
                float* g_kappa, tf_stream_t stream);
int tf_posenc_fwd(const float* x, int64_t n, int32_t d, int32_t n_freq, float* out, tf_stream_t stream);
/* linear_to_srgb (utils/raw_utils.py:4-17) on n floats, clamp01 != 0: followed by clamp(., 0, 1) as fields.py:1230-1256 wraps it;
 * _bwd: g_lin = g_out * d srgb / d lin of the branch taken (zero where the clamp is active).  Round 5: the training steps spent
 * nine element-wise launches per call on it, ~80 per material step. */
int tf_linear_to_srgb_fwd(const float* lin, int64_t n, int32_t clamp01, float* out, tf_stream_t stream);
int tf_linear_to_srgb_bwd(const float* lin, const float* g_out, int64_t n, int32_t clamp01, float* g_lin, tf_stream_t stream);

/* TVLoss.forward (network/other_field.py:170-191) on one [C,H,W] grid (B = 1: TensoSDF.TV_loss_sdf, fields.py:133-138).
 * tf_tv_fwd: partial [tf_tv_partials()] = per-block (sum of squared differences along H, along W), interleaved; the loss is
 *   weight * 2 * (sum_h / (C (H-1) W) + sum_w / (C H (W-1))) (a term whose count is 0 is left out), summed by the caller in a fixed order.
 * tf_tv_bwd: g_x [C,H,W] = g_dev[0] * (coef_h * d sum_h / dx + coef_w * d sum_w / dx) (overwritten); g_dev: the upstream gradient
 *   of the loss, one float on the device; coef_h = weight * 2 / count_h (0 if count_h == 0), coef_w likewise. */
int32_t tf_tv_partials(void);
int tf_tv_fwd(const float* x, int32_t C, int32_t H, int32_t W, float* partial, tf_stream_t stream);
/* loss[0] += coef_h * (sum of the H partials) + coef_w * (sum of the W partials), fixed order: several grids accumulate into one
 * device scalar (TensoSDF.TV_loss_sdf, fields.py:133-138: six grids) without element-wise launches between them (round 5). */
int tf_tv_finish(const float* partial, float coef_h, float coef_w, float* loss, tf_stream_t stream);
int tf_tv_bwd(const float* x, int32_t C, int32_t H, int32_t W, const float* g_dev, float coef_h, float coef_w, float* g_x,
              tf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TENSOFLOW_HIP_H */
