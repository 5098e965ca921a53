// Packed weight-stream layout shared by the pack kernel and the MLP kernels (see svs_mlp.hip).
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
constexpr int kChunk0F4 = kHdrF4 + 384;   // SDF layer-0 chunk: header + 20 float32 k-steps (K = 39 -> 40; 320 float4
                                          // used) or 3 fp16x2 k-steps (K -> 48; 384 float4)
constexpr int kLdsBytes = 2 * kChunkF4 * 16;  // double buffer = 72 KiB
constexpr int kRgbL0BodyF4 = 136 * 16;    // radiance layer 0: 128 feature k-steps + 8 k-steps of the 16 extra rows
constexpr int kRgbChunk0F4 = kHdrF4 + kRgbL0BodyF4;
constexpr int kW4TF4 = 8 * 64;            // radiance lin4 transposed: 8 tiles x 64 lanes x 4 k-steps
constexpr int kBlockF = 128 * 64;         // floats of one wave-tile activation block (256 rows x 32 points)
constexpr int kBgPeDim = 84;              // 4 * (1 + 2 * 10): PE-10 of the 4-D inverted-sphere points
constexpr int kBgChunk0F4 = kHdrF4 + 6 * 128;   // bg implicit layer 0: 6 fp16x2 k-steps
constexpr int kW1TF4 = 4 * 64;            // bg radiance lin1 transposed: 4 tiles x 64 lanes x 4 k-steps
constexpr int kBgRgbChunk0F4 = kHdrF4 + 18 * 128;   // bg radiance layer 0: 16 feature k-steps + 2 k-steps of view-PE rows
constexpr int kChunk0WF4 = kHdrF4 + 2 * 2 * 128;    // kSdfFwd0W: header + [2 sub-tiles][2 k-steps][hi, mid][64 lanes]

// chunk kinds (svs_pack.hip packs them, the kernels consume them in stream order)
enum ChunkKind {
  kSdfFwd0 = 0,   // SDF layer 0, output tile t (K = PE rows)
  kSdfFwd,        // SDF layer l, output tile t (K = C-layout rows of the layer input; lin4: skip splice, 1/sqrt2)
  kSdfVec,        // lin8 row 0 (sdf head) as a vector + b8[0]
  kSdfFeat,       // lin8 rows 1..256 (feature head), tile t
  kSdfRev,        // W_l^T: output = layer-input rows, K = layer-output rows (gradient pass / backward)
  kSdfRev0,       // W_0^T onto the PE rows (2 tiles, splice arrangement)
  kSdfFeatT,      // lin8[1:,:]^T: output = h_8 rows, K = feature-vector index
  kRgbFwd0,       // radiance layer 0 (K = 256 feature rows + 16 extra rows)
  kRgbFwd,        // radiance layers 1..4
  kRgbRev,        // radiance W_l^T, l = 1..3
  kRgbRev0,       // radiance W_0^T (9 tiles: 8 feature tiles + the extras tile)
  kRgbW4T,        // radiance lin4^T (3 output rows), short chunk
  // inverted-sphere background networks (VolSDFNetworkBG, network_bg.py): the implicit network reuses kSdfFwd /
  // kSdfVec / kSdfFeat / kSdfFeatT / kSdfRev with the bg geometry (84 PE rows, lin3 emits 172 rows)
  kBgFwd0,        // bg implicit layer 0 (K = 84 -> 96 = 6 fp16x2 k-steps)
  kBgRgbFwd0,     // bg radiance layer 0: 128 rows, K = 256 feature rows + 32 extra rows (27 view-PE entries)
  kBgRgbFwd1,     // bg radiance layer 1: 3 rows, K = 128
  kBgRgbW1T,      // bg radiance lin1^T (3 output rows), short chunk (4 tiles)
  kBgRgbRev0,     // bg radiance lin0^T onto the 256 feature rows, K = 128
  kSdfFwd0W,      // SDF layer 0 for the 16-point-wave kernels (kFmtF16x2W): K = 39 PE rows -> 64 = 2 k-steps of 32
};
constexpr int kNoBias = 0x100;            // flag: zero header

// Body encodings of the MFMA chunks (every kind except kSdfVec and kRgbW4T, which are always float32):
//   kFmtF32   k-step = 2 input rows, one float per lane, [k-step/4][lane][4]     (v_mfma_f32_32x32x2_f32)
//   kFmtF16x2 k-step = 16 input rows, per k-step 64 lanes x 8 fp16 of the hi piece, then of the mid piece
//             (v_mfma_f32_32x32x16_f16, svs_mlp_h2_dev.h).  Same bytes per chunk as kFmtF32.
//   kFmtF16x2W (internal: the streams of the 16-point-wave kernels, svs_mlp_w16.hip): v_mfma_f32_16x16x32_f16.  A chunk is
//             still 32 output rows = two 16-row sub-tiles u; k-step = 32 input rows; fragment index
//             ((u * KS + s) * 2 + piece) * 64 + lane, lane = 16 g + row: the lane's 8 fp16 are input rows
//             16 (2 s + (j >> 2)) + 4 g + (j & 3) -- registers 0..3 of lane group g of the producing layer's 16-row
//             output tiles 2 s and 2 s + 1, so a layer's accumulators are the next layer's B fragments as they stand.
//             Header: float4 index u * 64 + lane = the bias of rows 4 g .. 4 g + 3 of sub-tile u.
enum BodyFormat { kFmtF32 = 0, kFmtF16x2 = 1 };
constexpr int kFmtF16x2W = 3;
// `precision` of the C-ABI: the two above, and kFmtF16x2Half = fp16x2 kernels whose gradient-only activation blocks are
// stored as ONE fp16 piece (svs_blocks_h2.h, GP = false): half the backward's block bytes, parameter gradients 3e-4 ... 8e-4
// of a tensor's largest entry off instead of < 1e-5.  Weight streams are packed identically for both fp16x2 values.
constexpr int kFmtF16x2Half = 2;
__host__ __device__ constexpr bool is_h2(int precision) { return precision == kFmtF16x2 || precision == kFmtF16x2Half; }

__host__ __device__ constexpr int chunk_f4(int kind) {
  return kind == kSdfFwd0 ? kChunk0F4
       : kind == kRgbFwd0 ? kRgbChunk0F4
       : kind == kRgbW4T ? kW4TF4
       : kind == kBgFwd0 ? kBgChunk0F4
       : kind == kBgRgbFwd0 ? kBgRgbChunk0F4
       : kind == kBgRgbW1T ? kW1TF4
       : kind == kSdfFwd0W ? kChunk0WF4
       : kChunkF4;
}

// streams
enum StreamKind {
  kStreamSdfFwd = 0, kStreamSdfFull, kStreamSdfTrain, kStreamRgbFwd, kStreamRgbBwd,
  kStreamBgFwd,      // bg implicit network: trunk + head vector + feature head
  kStreamBgTrain,    // bg implicit network backward: transposed feature head + transposed trunk
  kStreamBgRgbFwd, kStreamBgRgbBwd,
  kStreamSdfFwdW,    // SDF forward (trunk + head vector) in the kFmtF16x2W encoding: svs_mlp_w16.hip
  kNumStreams
};
constexpr bool stream_is_bg(int which) { return which >= kStreamBgFwd && which <= kStreamBgRgbBwd; }
// offsets inside the SDF training stream (float4): pass A starts at 0, pass B after the 63 forward chunks
constexpr size_t kSdfTrainPassBF4 = 8 * (size_t)kChunk0F4 + 55 * (size_t)kChunkF4;

}  // namespace mlp
}  // namespace svs
