// Fragment images of one coupling net (44-64-64-64-21, layer-1 point part hoisted) as flow.hip and flow_bwd.hip hold them in LDS.
#pragma once
// exact-fp32 image (floats): L1s [2][4][64] | L2 [2][32][64] | L3 [2][32][64] | L4 [1][32][64] | b2 [64] | b3 [64] | b4 [32]
// (biases in accumulator order)
static constexpr int kL1 = 0, kL2 = kL1 + 2 * 4 * 64, kL3 = kL2 + 2 * 32 * 64, kL4 = kL3 + 2 * 32 * 64,
                     kB2 = kL4 + 32 * 64, kB3 = kB2 + 64, kB4 = kB3 + 64, kNetFloats = kB4 + 32;
// f16x3 image (offsets in floats; one (s16, tout) fragment pair = 512 floats = 2 KB):
//   L1s [1][2] | L2 [4][2] | L3 [4][2] | L4 [4][1] | b2 | b3 | b4
static constexpr int hL1 = 0, hL2 = hL1 + 2 * 512, hL3 = hL2 + 8 * 512, hL4 = hL3 + 8 * 512, hB2 = hL4 + 4 * 512,
                     hB3 = hB2 + 64, hB4 = hB3 + 64, hNetFloats = hB4 + 32;
static_assert(hNetFloats <= kNetFloats + 512, "workspace sizing assumes the f16x3 image is not much larger");
