// Packed-weight workspace of the inner-light net (floats), shared by inner_light.hip (the staggered kernel, the launchers, the
// encode kernel) and inner_light_modes.hip (the exact-fp32 and the plain-f16 operand modes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// fragment workspace layout (floats)
static constexpr int kI1 = 0;                       // [8][64][64]   123 -> 256 (K padded to 128)
static constexpr int kI2 = kI1 + 8 * 64 * 64;       // [8][128][64]
static constexpr int kI3 = kI2 + 8 * 128 * 64;      // [8][128][64]
static constexpr int kI4 = kI3 + 8 * 128 * 64;      // [1][128][64]  256 -> 3
static constexpr int kIB1 = kI4 + 128 * 64;         // biases, accumulator order
static constexpr int kIB2 = kIB1 + 256;
static constexpr int kIB3 = kIB2 + 256;
static constexpr int kIB4 = kIB3 + 256;
static constexpr int kIdeMat = kIB4 + 32;           // [17][36] IDE polynomial coefficients
// f16x3 fragments (hi|lo halves), offsets in FLOAT units (each k-step16 of 8 unit tiles = 4096 floats)
static constexpr int kH1 = ((kIdeMat + 17 * 36 + 1023) / 1024) * 1024;   // 123 -> 256: 8 k-steps16
static constexpr int kH2 = kH1 + 8 * 4096;          // 256 -> 256: 16 k-steps16
static constexpr int kH3 = kH2 + 16 * 4096;
static constexpr int kH4 = kH3 + 16 * 4096;         // 256 -> 3: 16 k-steps16 x 1 tile = 2 slabs
static constexpr int kP1 = kH4 + 2 * 4096;          // 123 -> 256 with the IDE features first (inner_light_cols_kernel): 8 k-steps16
static constexpr int kWp = kP1 + 8 * 4096;          // [256][123] scratch of the column permutation
static constexpr int kW4a = kWp + 256 * 128 + 32;   // [3][256]: rows of the 256 -> 3 layer in accumulator order (tf_pack_bias_kernel), staggered kernel
static constexpr int kQ1 = ((kW4a + 3 * 256 + 1023) / 1024) * 1024;   // 123 -> 256 in the staggered kernel's input order (il3_orig_col): 8 k-steps16
static constexpr int kInnerWsFloats = kQ1 + 8 * 4096;

// inner_light_modes.hip: the two operand modes that do not run on the staggered kernel.  Launch only (arguments validated and
// weights packed by inner_light_launch); 0 / TF_EHIP through the usual launch check of the caller.
void tf_inner_light_launch_f32(const float* workspace, const float* pts, const float* view, const float* nrm, long long m, const long long* idx,
                               const long long* count_dev, const float* depth, float near_eps, float exp_max, float* out, hipStream_t stream);
void tf_inner_light_launch_f16(const float* workspace, const float* pts, const float* view, const float* nrm, long long m, const long long* idx,
                               const long long* count_dev, const float* depth, float near_eps, float exp_max, float* out, hipStream_t stream);
