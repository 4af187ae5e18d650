// Measurement aid of the roofline figures (bench.py): what the matrix cores of THIS device sustain on the instruction the decoders
// run on.  The reference has no counterpart (it never prices its kernels); SURVEY.md 8(d) asks for achieved / peak of the dominant
// kernel, and on this part the spec peak (2.5 PFLOP/s dense f16) is not a rate a kernel can hold: under a dense stream of
// v_mfma_f32_32x32x16_f16 on non-trivial operands the chip lowers its clock (MI355X_MICROARCH.md, "DVFS give-back": 1.6-1.7 GHz held
// against 2.4 GHz quoted), so the line carries this measured ceiling beside the spec peak.
//   one iteration of a wave = the inner-light kernel's k-step pair: 64 units x 64 rays x 32 k as 2 x 2 tiles x 2 k-steps x 3 product
//   terms (24 MFMAs), operands re-read from LDS by ds_read_b128, four waves per CU (one per SIMD), pseudo-random f16 fragments.
#include "tf_common.h"

typedef _Float16 pr_h8 __attribute__((ext_vector_type(8)));
typedef float pr_f16v __attribute__((ext_vector_type(16)));

static __global__ void __launch_bounds__(256) probe_mfma_kernel(int iters, int relu_like, float* __restrict__ out) {
  __shared__ pr_h8 frag[4096];      // 64 KB
  for (int i = threadIdx.x; i < 4096; i += 256) {
    pr_h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned s = (unsigned)(i * 8 + e) * 2654435761u + 12345u;       // integer hash -> values in (-1, 1)
      s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
      float x = ((float)(s & 0xffffu) - 32768.f) * (1.f / 32768.f);
      // relu_like: the B operands (fragments 256.. of every 512-fragment slab: the "activations") are ReLU outputs -- the negative half
      // is zero, as in the decoder's hidden layers (the part's clock under an MFMA stream depends on what the multipliers toggle)
      if (relu_like && ((i & 511) >= 256) && x < 0.f) x = 0.f;
      v[e] = (_Float16)x;
    }
    frag[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  pr_f16v acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const pr_h8* f = frag + ((it & 7) * 512) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      pr_h8 a[2][2], b[2][2];        // [tile][hi | lo]
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) { a[t][p] = f[((s * 2 + t) * 2 + p) * 64]; b[t][p] = f[(((s + 2) * 2 + t) * 2 + p) * 64 % 512]; }
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t * 2 + r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[r][0], acc[t * 2 + r], 0, 0, 0);
          acc[t * 2 + r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[r][1], acc[t * 2 + r], 0, 0, 0);
          acc[t * 2 + r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][1], b[r][0], acc[t * 2 + r], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[n][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int tf_probe_mfma_f16(int32_t iters, int32_t relu_like, float* scratch, int64_t scratch_floats, double* tflops_host, tf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TF_REQUIRE(iters > 0 && scratch && tflops_host, TF_EINVAL, "tf_probe_mfma_f16: iters <= 0 or null pointer");
  int dev = 0;
  hipDeviceProp_t prop;
  TF_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess, TF_EHIP, "tf_probe_mfma_f16: no device");
  const int cus = prop.multiProcessorCount;
  TF_REQUIRE(scratch_floats >= 256LL * cus, TF_ESHAPE, "tf_probe_mfma_f16: scratch needs 256 floats per CU (%d CUs)", cus);
  hipEvent_t e0, e1;
  TF_REQUIRE(hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess, TF_EHIP, "tf_probe_mfma_f16: hipEventCreate failed");
  probe_mfma_kernel<<<cus, 256, 0, stream>>>(iters / 8 + 1, relu_like, scratch);        // settle the clock under this load
  hipEventRecord(e0, stream);
  probe_mfma_kernel<<<cus, 256, 0, stream>>>(iters, relu_like, scratch);
  hipEventRecord(e1, stream);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  TF_REQUIRE(e == hipSuccess && ms > 0.f, TF_EHIP, "tf_probe_mfma_f16: timing failed: %s", hipGetErrorString(e));
  // 4 waves per CU x iters x 24 MFMAs x 2 * 32 * 32 * 16 flop
  *tflops_host = (double)cus * 4.0 * iters * 24.0 * 32768.0 / (ms * 1e-3) * 1e-12;
  return TF_OK;
}
