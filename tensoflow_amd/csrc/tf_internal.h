// Helpers shared BETWEEN translation units of libtensoflow_hip.so (not part of the C ABI: include/tensoflow_hip.h is).
#pragma once
#include <hip/hip_runtime.h>

#include "tf_common.h"

// linear.hip: gX [n,K] = gZ [n,N] . W [N,K] (overwritten; may be NULL) and gW [N,K] += gZ^T . X (fp32 atomics into the buffer as it
// is; may be NULL).  precision: TF_PREC_F32 (exact fp32 matrix cores) or TF_PREC_F16X3.  n_dev: optional device-side row count.
int tf_linear_products(const float* X, const float* W, const float* gZ, long long n, int K, int N, int precision, float* gX, float* gW,
                       const long long* n_dev, hipStream_t stream);

// vm_field.hip: the adjoint scatter of the VM gather for the 7 finite-difference taps of n_pts points (row = tap * n_pts + point;
// tap 0 = the point, 1 + 2 ax / 2 + 2 ax = +- units[ax] along axis ax); gfeat [7 n_pts, ld], gpacked += (fp32 atomics).
int tf_vm_scatter_taps(const VmGeom& g, const float* packed, const float* pts, const float* level, long long n_pts, const float* units,
                       const float* gfeat, int ld, float* gpacked, hipStream_t stream);

// capi.cpp: the per-thread launch budget set through tf_set_launch_budget (0 = the kernel's own default).  Read by the launchers of
// the three stage kernels of the rendering integral at enqueue time, so that a caller that runs them on different streams can leave
// room on every CU for the others (registers / LDS / wave slots are what decides whether two kernels are co-resident).
struct TfLaunchBudget {
  int bvh_blocks_per_cu = 0;       // persistent 256-thread traversal workgroups per CU (one wave per SIMD each; default: all that fit, 7)
  int flow_waves_per_block = 0;    // 4 / 8 / 12 waves in the flow kernel's one workgroup per CU (default 12: three per SIMD)
  int inner_teams = 0;             // 1: the staggered inner-light kernel as ONE four-wave team per workgroup (256 threads, half the LDS and
                                   //    half the register file of a CU); default 2 (a 512-thread workgroup owns the CU)
};
const TfLaunchBudget& tf_launch_budget();
