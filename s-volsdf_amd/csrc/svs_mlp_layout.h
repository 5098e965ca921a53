// Packed weight-stream layout shared by the pack kernels and the MLP kernels (see svs_mlp.hip).
#pragma once
#include "svs_common.h"

namespace svs {
namespace mlp {

constexpr int kWaves = 4;                 // waves per workgroup (one per SIMD: the kernels need > 256 VGPRs)
constexpr int kThreads = kWaves * 64;
constexpr int kTilePts = 32;              // points per wave
constexpr int kWgPts = kWaves * kTilePts; // points per workgroup
constexpr int kHdrF4 = 256;               // chunk header: 16 accumulator regs x 64 lanes (bias block), in float4
constexpr int kBodyF4 = 2048;             // 128 k-steps x 64 lanes floats
constexpr int kChunkF4 = kHdrF4 + kBodyF4;  // 36 KiB
constexpr int kPeDim = 39;                // 3 * (1 + 2 * 6), embedder.py:38-50 with multires = 6
constexpr int kChunk0F4 = kHdrF4 + 320;   // SDF layer-0 chunk: header + 20 k-steps (K = 39 -> 40)
constexpr int kLdsBytes = 2 * kChunkF4 * 16;  // double buffer = 72 KiB

// chunk kinds of the SDF-MLP stream, in stream order
//   forward : 8 x FWD0 | L1,L2: 8 x FWD | L3: 7 x FWD | L4..L7: 8 x FWD | VEC
//   full    : forward | 8 x FEAT | L7..L1 reverse: 8 x REV each | 2 x REV0
constexpr int kSdfFwdChunks = 8 + 7 * 8 - 1 + 1;                    // 64
constexpr size_t kSdfFwdF4 = 8 * (size_t)kChunk0F4 + (size_t)(kSdfFwdChunks - 8) * kChunkF4;
constexpr int kSdfFullChunks = kSdfFwdChunks + 8 + 7 * 8 + 2;       // 130
constexpr size_t kSdfFullF4 = kSdfFwdF4 + (size_t)(kSdfFullChunks - kSdfFwdChunks) * kChunkF4;

// radiance stream: L0: 8 chunks of (hdr + 136 k-steps: 256 feature rows + 16 extra rows) | L1..L3: 8 x FWD | L4: 1 x FWD
constexpr int kRgbL0BodyF4 = 136 * 16;    // 2176
constexpr int kRgbChunk0F4 = kHdrF4 + kRgbL0BodyF4;
constexpr int kRgbChunks = 8 + 3 * 8 + 1;
constexpr size_t kRgbF4 = 8 * (size_t)kRgbChunk0F4 + 25 * (size_t)kChunkF4;   // last chunk: lin4 (3 rows) as one tile

}  // namespace mlp
}  // namespace svs
