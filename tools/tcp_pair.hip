// Dev microbenchmark: does the vector L1 (TCP) serve two neighbouring lanes that read the two 16-byte halves of ONE 32-byte record in one
// access?  Every lane needs a random 32-byte record of an L2-resident array per step.
//   mode 0: each lane reads its own record with two dwordx4 loads (today's pair fetch in bvh.hip)
//   mode 1: lanes 2k / 2k+1 read the halves of record(2k) together, then the halves of record(2k+1), and swap through DPP
//   mode 2: like 0 but the next index depends on the loaded data (a traversal's dependent chain)
//   mode 3: like 1 with the dependent chain
// hipcc --offload-arch=gfx950 -O3 tools/tcp_pair.hip -o tools/tcp_pair.bin && tools/tcp_pair.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__device__ __forceinline__ unsigned swap1(unsigned v) {   // value of the lane with the lowest id bit flipped
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
}

template <int MODE>
__global__ void __launch_bounds__(256, 8) probe(const uint4* rec, unsigned nrec_mask, int steps, unsigned* out) {
  const unsigned lane = threadIdx.x & 63;
  unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  unsigned acc = 0;
  for (int s = 0; s < steps; ++s) {
    const unsigned r = idx & nrec_mask;
    uint4 q0, q1;
    if (MODE == 0 || MODE == 2) {
      q0 = rec[2 * r]; q1 = rec[2 * r + 1];
    } else if (MODE == 4) {      // the four lanes of a quad read the four 16-byte pieces of ONE 64-byte record (the quad leader's)
      const unsigned rq = (unsigned)__builtin_amdgcn_mov_dpp((int)r, 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, true) & (nrec_mask >> 1);
      q0 = rec[4 * rq + (lane & 3u)]; q1 = make_uint4(0, 0, 0, 0);
    } else if (MODE == 5) {      // 8 bytes per lane at a random place
      const uint2 t = reinterpret_cast<const uint2*>(rec)[4 * r]; q0 = make_uint4(t.x, t.y, 0, 0); q1 = make_uint4(0, 0, 0, 0);
    } else if (MODE == 6) {      // 4 bytes per lane at a random place
      const unsigned t = reinterpret_cast<const unsigned*>(rec)[8 * r]; q0 = make_uint4(t, 0, 0, 0); q1 = make_uint4(0, 0, 0, 0);
    } else if (MODE == 7) {      // ONE 16-byte load per lane at a random place
      q0 = rec[2 * r]; q1 = make_uint4(0, 0, 0, 0);
    } else {
      const unsigned half = lane & 1u;
      const unsigned r_other = swap1(r);
      const unsigned ra = half ? r_other : r;          // the even lane's record
      const unsigned rb = half ? r : r_other;          // the odd lane's record
      const uint4 X = rec[2 * ra + half];              // even: a0, odd: a1
      const uint4 Y = rec[2 * rb + half];              // even: b0, odd: b1
      uint4 G;                                         // what this lane hands over: even gives Y (b0), odd gives X (a1)
      G.x = half ? X.x : Y.x; G.y = half ? X.y : Y.y; G.z = half ? X.z : Y.z; G.w = half ? X.w : Y.w;
      uint4 R; R.x = swap1(G.x); R.y = swap1(G.y); R.z = swap1(G.z); R.w = swap1(G.w);
      q0.x = half ? R.x : X.x; q0.y = half ? R.y : X.y; q0.z = half ? R.z : X.z; q0.w = half ? R.w : X.w;
      q1.x = half ? Y.x : R.x; q1.y = half ? Y.y : R.y; q1.z = half ? Y.z : R.z; q1.w = half ? Y.w : R.w;
    }
    const unsigned mix = q0.x ^ q0.y ^ q0.z ^ q0.w ^ q1.x ^ q1.y ^ q1.z ^ q1.w;
    acc += mix;
    if (MODE >= 2) idx = idx * 1664525u + 1013904223u + (mix & 0xffu);
    else idx = idx * 1664525u + 1013904223u;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const unsigned nrec = 1u << 16;                      // 65536 records x 32 B = 2 MB (fits one XCD's L2)
  std::vector<uint32_t> h(8u * nrec);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2246822519u + 12345u);
  uint4* d; unsigned* out;
  hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int blocks = 256 * 8, steps = 2000;
  hipMalloc(&out, blocks * 256 * 4);
  std::vector<unsigned> ref(blocks * 256), got(blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pass = 0; pass < 2; ++pass)
  for (int mode = 0; mode < 8; ++mode) {
    const unsigned nrec_used = pass ? 512u : nrec;   // pass 1: 16 KB footprint (L1-resident)
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) probe<0><<<blocks, 256>>>(d, nrec_used - 1, steps, out);
      if (mode == 1) probe<1><<<blocks, 256>>>(d, nrec_used - 1, steps, out);
      if (mode == 2) probe<2><<<blocks, 256>>>(d, nrec_used - 1, steps, out);
      if (mode == 3) probe<3><<<blocks, 256>>>(d, nrec_used - 1, steps, out);
      if (mode == 4) probe<4><<<blocks, 256>>>(d, nrec_used - 1, steps, out);
      if (mode == 5) probe<5><<<blocks, 256>>>(d, nrec_used - 1, steps, out);
      if (mode == 6) probe<6><<<blocks, 256>>>(d, nrec_used - 1, steps, out);
      if (mode == 7) probe<7><<<blocks, 256>>>(d, nrec_used - 1, steps, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost);
    if (mode == 0 || mode == 2) ref = got;
    bool same = ref == got;
    const double recs = (double)blocks * 256 * steps;
    printf("footprint %u KB mode %d: %.3f ms  %.2f G lane-steps/s  %.2f lane-steps/clk/CU at 2.1 GHz  results %s\n", nrec_used * 32 / 1024, mode, best, recs / best * 1e-6,
           recs / best * 1e-6 / 256 / 2.1, mode > 3 ? "-" : same ? "identical to the plain fetch" : "DIFFER");
  }
  return 0;
}
