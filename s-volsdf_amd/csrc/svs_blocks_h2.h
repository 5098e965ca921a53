// Activation blocks of the fp16x2 path in HBM (the float32-MFMA kernels keep plain float32 blocks, svs_mlp_dev.h).
//
// A block slot is unchanged: kBlockF floats = 32 KiB per wave tile (256 feature rows x 32 points), slots laid out
// [block][wave tile].  What the fp16x2 kernels put INTO a slot is the operand form the matrix cores consume, so no
// consumer converts or re-splits anything:
//
//   pair block   every value as its two fp16 pieces (22 significand bits, svs_mlp_h2_dev.h): the hi plane, 16 KiB =
//                [16 k-steps][64 fragments][16 B], float4 index s * 64 + piece_slot(s, lane), then the mid plane at 1024 + ...
//                The 16 bytes of lane L in k-step s are the MFMA B fragment of that k-step: rows 16 s + 8 (j >> 2) +
//                4 (L >> 5) + (j & 3), j = 0..7, of point L & 31 -- registers 8 (s & 1) .. 8 (s & 1) + 7 of accumulator
//                tile s >> 1.  Forward activations (h_l, the feature vector, the radiance network's r_l, PE(x)).
//   scaled block  value * 2^k with one power of two per point and block, + a RECORD of 64 floats [scale of point 0..31]
//                [max |value| of point 0..31].  Everything that exists only to form parameter gradients: ghat_l (stored
//                unscaled, no record), u_l, a2_l, abar_l, zbar_l, fbar.  Two formats, chosen per launch (template
//                parameter GP of the sweeps, `precision` of the C-ABI):
//                  GP = true  (SVS_MMA_F16X2, the default): both pieces, hi plane then mid plane, like a pair block --
//                             22 significand bits; parameter gradients within 1e-5 of float64 autograd, the float32 class
//                  GP = false (SVS_MMA_F16X2_HALF): the hi plane only, 11 bits, half the bytes; parameter gradients
//                             3e-4 ... 8e-4 of a tensor's largest entry off (tools/study/fp16_blocks_error.py); sweeps
//                             that only need softplus' of h then read the hi plane of the pair block alone.
//
// Slot offsets and strides are those of the float32 blocks, so host code is format-agnostic; a half block leaves the second
// half of its slot untouched.  The records of a buffer of nb blocks x T wave tiles live BEHIND its slots:
// [nb][T][kBlockF] floats, then [nb][T][64] floats (record_ptr(); the size functions of the C-ABI include them).
#pragma once
#include "svs_mlp_h2_dev.h"

namespace svs {
namespace mlp {

constexpr int kPlaneF4 = 1024;          // float4 per fp16 plane of a wave tile (16 KiB)
constexpr int kRecF = 64;               // floats per record of a scaled block

__device__ __forceinline__ f32x4 as_f4(const f16x8& v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ f16x8 as_h8(const f32x4& v) { return __builtin_bit_cast(f16x8, v); }

// Where lane L's fragment of k-step s sits inside the k-step's 1 KiB (in 16-byte slots).  A permutation of the 64 slots,
// chosen for the one consumer that does not read whole fragments: the weight-gradient GEMM copies a plane into LDS as it
// stands (LDS-DMA) and reads it TRANSPOSED (ds_read_b64_tr_b16: 4 points x 16 features per 16 lanes); with L = 32 b + 4 a
// + q (b = feature half, a = group of four points, q = point in the group) the slot 16 (a >> 1) + 8 ((a & 1) ^ (s & 1)) +
// 4 b + q makes every such read conflict-free (svs_wgrad.hip).  A wave still moves the whole 1 KiB per instruction.
__device__ __forceinline__ int piece_slot(int s, int lane) {
  const int b = lane >> 5, a = (lane >> 2) & 7, q = lane & 3;
  return 16 * (a >> 1) + 8 * ((a & 1) ^ (s & 1)) + 4 * b + q;
}

// one fragment (k-step s) of a plane; plane 0 = hi, 1 = mid
__device__ __forceinline__ void store_piece(float* __restrict__ blk, int s, int lane, const f16x8& v, int plane = 0) {
  SVS_STREAM_STORE(as_f4(v), reinterpret_cast<f32x4*>(blk) + plane * kPlaneF4 + s * 64 + piece_slot(s, lane));
}
__device__ __forceinline__ f16x8 load_piece(const float* __restrict__ blk, int s, int lane, int plane = 0) {
  return as_h8(SVS_STREAM_LOAD(reinterpret_cast<const f32x4*>(blk) + plane * kPlaneF4 + s * 64 + piece_slot(s, lane)));
}

// accumulator-layout tile t (16 registers) back from stored fragments
struct TilePieces {
  f16x8 h[2], m[2];     // k-steps 2t, 2t+1
};
__device__ __forceinline__ void load_tile_hi(const float* __restrict__ blk, int t, int lane, TilePieces& p) {
  p.h[0] = load_piece(blk, 2 * t, lane); p.h[1] = load_piece(blk, 2 * t + 1, lane);
}
__device__ __forceinline__ void load_tile_pair(const float* __restrict__ blk, int t, int lane, TilePieces& p) {
  load_tile_hi(blk, t, lane, p);
  p.m[0] = load_piece(blk, 2 * t, lane, 1); p.m[1] = load_piece(blk, 2 * t + 1, lane, 1);
}
// element r (0..15) of the tile
__device__ __forceinline__ float hi_at(const TilePieces& p, int r) { return (float)p.h[r >> 3][r & 7]; }
__device__ __forceinline__ float pair_at(const TilePieces& p, int r) { return (float)p.h[r >> 3][r & 7] + (float)p.m[r >> 3][r & 7]; }

// The same sums and products in one instruction each (v_fma_mix_f32 converts its fp16 operands on the way in; hipcc 7.2 does
// not form it from the C expressions above but emits cvt + cvt + add).  j: the element's index in its fragment (0..7), a
// constant once the epilogue loops are unrolled: the branch on its parity folds away.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned word_of(const f16x8& v, int j) { return __builtin_bit_cast(u32x4, v)[j >> 1]; }
__device__ __forceinline__ float mix_sum(unsigned h, unsigned m, int odd) {      // (float)h.half[odd] + (float)m.half[odd]
  float r;
  if (odd) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(m));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(m));
  return r;
}
__device__ __forceinline__ float mix_fma(unsigned h, int odd, float c, float a) {   // (float)h.half[odd] * c + a
  float r;
  if (odd) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c), "v"(a));
  else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c), "v"(a));
  return r;
}
__device__ __forceinline__ float mix_mul(unsigned h, int odd, float c) {            // (float)h.half[odd] * c
  float r;
  if (odd) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
  else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
  return r;
}
// element r of a tile: value * c (both pieces with GP: hi * c + mid * c, two instructions)
template <bool GP>
__device__ __forceinline__ float grad_times(const TilePieces& p, int r, float c) {
  const int q = r >> 3, j = r & 7;
  if (GP) return mix_fma(word_of(p.h[q], j), j & 1, c, mix_mul(word_of(p.m[q], j), j & 1, c));
  return mix_mul(word_of(p.h[q], j), j & 1, c);
}
template <bool GP>
__device__ __forceinline__ float grad_mix(const TilePieces& p, int r) {
  const int q = r >> 3, j = r & 7;
  if (GP) return mix_sum(word_of(p.h[q], j), word_of(p.m[q], j), j & 1);
  return mix_mul(word_of(p.h[q], j), j & 1, 1.0f);
}

// The record of block l, wave tile `tile` of a buffer of nb blocks x T wave tiles laid out [block][tile] (T = the padded
// tile count of the launch that wrote it: gridDim.x * kWaves).
template <typename F>
__device__ __forceinline__ F* record_ptr(F* buf, int nb, size_t T, int l, size_t tile) {
  return buf + (size_t)nb * T * kBlockF + ((size_t)l * T + tile) * kRecF;
}
// lane L < 32 writes the scale of its point, lane L >= 32 the maximum (both lane halves hold both values)
__device__ __forceinline__ void store_record(float* __restrict__ rec, int lane, float scale, float mx) {
  rec[lane] = lane < 32 ? scale : mx;
}
__device__ __forceinline__ float load_scale(const float* __restrict__ rec, int lane) { return rec[lane & 31]; }
__device__ __forceinline__ float load_max(const float* __restrict__ rec, int lane) { return rec[32 + (lane & 31)]; }

// ---- scaled (gradient-only) blocks in the launch's format GP
// the fragment of k-step s: hi always, mid when GP
template <bool GP>
__device__ __forceinline__ void store_grad(float* __restrict__ blk, int s, int lane, const f16x8& h, const f16x8& m) {
  store_piece(blk, s, lane, h, 0);
  if (GP) store_piece(blk, s, lane, m, 1);
}
// 8 float32 values (already scaled) -> the stored fragment(s) of k-step s
template <bool GP>
__device__ __forceinline__ void store_grad8(float* __restrict__ blk, int s, int lane, const float* v) {
  if (GP) {
    f16x8 h, m;
    split8(v, h, m);
    store_piece(blk, s, lane, h, 0);
    store_piece(blk, s, lane, m, 1);
  } else {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
    store_piece(blk, s, lane, h, 0);
  }
}
template <bool GP>
__device__ __forceinline__ void load_tile_grad(const float* __restrict__ blk, int t, int lane, TilePieces& p) {
  if (GP) load_tile_pair(blk, t, lane, p); else load_tile_hi(blk, t, lane, p);
}
template <bool GP>
__device__ __forceinline__ float grad_at(const TilePieces& p, int r) { return GP ? pair_at(p, r) : hi_at(p, r); }

// Fragment of k-step s of a vector given in natural row order (vec[q], q < n, zero beyond), in the BLOCK convention
// (row 16 s + 8 (j >> 2) + 4 half + (j & 3)) -- NOT the order split_pe() uses for the layer-0 operand (16 s + 8 half + j,
// which the packed layer-0 weights follow): what a weight-gradient GEMM reads as its B operand must be in block order.
template <int N>
__device__ __forceinline__ void block_fragment(const float* vec, int s, int half, float scale, f16x8& h, f16x8& m) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int q0 = 16 * s + 8 * (j >> 2) + (j & 3), q1 = q0 + 4;
    const float a0 = q0 < N ? vec[q0 < N ? q0 : 0] : 0.0f;
    const float a1 = q1 < N ? vec[q1 < N ? q1 : 0] : 0.0f;
    v[j] = (half ? a1 : a0) * scale;
  }
  split8(v, h, m);
}

// 8 float32 values -> the hi fragment of value * s (the stored form of a half block)
__device__ __forceinline__ f16x8 hi8(const float* v, float s) {
  f16x8 h;
#pragma unroll
  for (int j = 0; j < 8; ++j) h[j] = (_Float16)(v[j] * s);
  return h;
}

}  // namespace mlp
}  // namespace svs
