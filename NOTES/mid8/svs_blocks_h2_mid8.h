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

// ---- three-byte scaled blocks (round 5): the hi plane as above + an 8-BIT mid plane.  The residual value - hi lies within half
// a unit in the last place of hi, so its exponent is known from hi: it is stored as a signed byte k in units of
//     unit(hi) = ulp(max(|hi|, 2^-4)) / 256 = 2^(max(e_b, 11) - 33)        (e_b: the biased exponent field of hi)
// -- 19 significand bits instead of 22 (values below 2^-4 -- 2^-8 of the point's largest, scaled blocks put that at 2^4 --
// keep the absolute resolution 2^-22; the clamp keeps 256 unit and its inverse normal fp16 numbers, so that both directions
// are a few PACKED 16-bit instructions), three bytes per value instead of four, and a zero-initialised block still reads
// as zeros.  (A one-piece block, 11 bits, costs the parameter gradients 2e-4 ... 8e-4 of a tensor's largest entry; 8 more bits
// bring that to ~1e-6: the float32 class holds, tests/test_gpu_train.py::test_step_gradient_at_bench_geometry.)  The step is
// HBM-bound by exactly these blocks: every one of them is written once and read one to four times.
// Layout of the mid8 plane: at the fp16 mid plane's place (float4 index kPlaneF4), 8 KiB: [tile t = 0..7][lane][16 bytes] =
// the 8 residual bytes of k-step 2t (elements j = 0..7 of lane's fragment) then those of k-step 2t + 1 -- one 16-byte store /
// load per lane and tile.  Consumers: the sweeps decode in registers (mid8_value); the weight-gradient GEMM copies the plane
// into LDS as it stands and expands it there into the fp16 mid plane its MFMA loop reads (svs_wgrad.hip).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));

// unit(hi) as a float32, from hi's bits
__device__ __forceinline__ float mid8_unit(unsigned hi_bits) {
  unsigned e1 = (hi_bits >> 10) & 31u;
  e1 = e1 < 11u ? 11u : e1;
  return __uint_as_float((e1 + 94u) << 23);
}
// The residual bytes of a fragment from its two fp16 pieces (h, m = fp16(value - h), what split8 produces): per fp16 pair
//   1 / (256 unit) = 2^(25 - max(e_b, 11)) as an fp16 (fields 40 - e_c = 10 .. 29),  k = rint(m * that * 256),  |k| <= 128
// evaluated by ONE packed fma onto 1536 (ulp 1 there: the low byte of the result's bit pattern IS k in two's complement);
// +128 -- only an exact tie of the fp16 rounding gives it -- is clamped to 127.  Packed 16-bit arithmetic on the pieces the
// sweep has anyway: 26 instructions per fragment, no float32 temporaries.
__device__ __forceinline__ u32x2 mid8_bytes(const f16x8& h, const f16x8& m) {
  // (written on whole 8-element vectors: with per-dword extracts of the bit-cast pieces hipcc of ROCm 7.2 multiplied every
  // pair by the FIRST pair's residuals -- tools/micro/mid8_dbg.hip)
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  const u16x8 e = __builtin_bit_cast(u16x8, h) & (unsigned short)0x7c00;
  const u16x8 ec = __builtin_elementwise_max(e, (u16x8)((unsigned short)0x2c00));
  const f16x8 inv = __builtin_bit_cast(f16x8, (u16x8)((u16x8)((unsigned short)0xA000) - ec));      // 2^(25 - e_c)
  f16x8 t = m * inv;                                                                                // residual / ulp: [-0.5, 0.5]
  t = __builtin_elementwise_min(t, (f16x8)((_Float16)(127.0f / 256.0f)));
  const f16x8 kf = t * (f16x8)((_Float16)256.0f) + (f16x8)((_Float16)1536.0f);
  const u32x4 kb = __builtin_bit_cast(u32x4, kf);
  u32x2 out;
  out[0] = __builtin_amdgcn_perm(kb[1], kb[0], 0x06040200u);       // the low bytes of the four halves of (kb[0], kb[1])
  out[1] = __builtin_amdgcn_perm(kb[3], kb[2], 0x06040200u);
  return out;
}
// the fp16 mid fragment (what an MFMA multiplies) of a fragment's hi piece and its 8 residual bytes: k / 256 by one packed fma
// on the halves 0x6600 | (byte ^ 0x80) = 1664 + k, times 256 unit = 2^(e_c - 25) (field e_c - 10): 14 packed instructions
__device__ __forceinline__ f16x8 mid8_fragment(const f16x8& h, const u32x2& m8) {
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  const u16x8 e = __builtin_bit_cast(u16x8, h) & (unsigned short)0x7c00;
  const u16x8 ec = __builtin_elementwise_max(e, (u16x8)((unsigned short)0x2c00));
  const f16x8 pw = __builtin_bit_cast(f16x8, (u16x8)(ec - (u16x8)((unsigned short)0x2800)));
  u32x4 kb;
  kb[0] = __builtin_amdgcn_perm(0x66666666u, m8[0], 0x04010400u) ^ 0x00800080u;
  kb[1] = __builtin_amdgcn_perm(0x66666666u, m8[0], 0x04030402u) ^ 0x00800080u;
  kb[2] = __builtin_amdgcn_perm(0x66666666u, m8[1], 0x04010400u) ^ 0x00800080u;
  kb[3] = __builtin_amdgcn_perm(0x66666666u, m8[1], 0x04030402u) ^ 0x00800080u;
  const f16x8 kf = __builtin_bit_cast(f16x8, kb) * (f16x8)((_Float16)(1.0f / 256.0f)) - (f16x8)((_Float16)6.5f);
  return kf * pw;
}
// 8 float32 values (already scaled) -> their hi fragment and the 8 residual bytes
__device__ __forceinline__ void split8_mid8(const float* v, f16x8& h, u32x2& m8) {
  f16x8 m;
  split8(v, h, m);
  m8 = mid8_bytes(h, m);
}
// the residual of element j (0..7) of a fragment whose 8 residual bytes are m8
__device__ __forceinline__ float mid8_value(const f16x8& h, const u32x2& m8, int j) {
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  const unsigned hb = (unsigned)__builtin_bit_cast(u16x8, h)[j];     // (bit_cast of the single element h[j] read as 0: hipcc 7.2)
  const int k = (int)(m8[j >> 2] << (24 - 8 * (j & 3))) >> 24;      // the signed byte
  return (float)k * mid8_unit(hb);
}
__device__ __forceinline__ void store_mid8(float* __restrict__ blk, int t, int lane, const u32x2& k0, const u32x2& k1) {
  u32x4 v; v[0] = k0[0]; v[1] = k0[1]; v[2] = k1[0]; v[3] = k1[1];
  SVS_STREAM_STORE(__builtin_bit_cast(f32x4, v), reinterpret_cast<f32x4*>(blk) + kPlaneF4 + t * 64 + lane);
}
__device__ __forceinline__ u32x4 load_mid8(const float* __restrict__ blk, int t, int lane) {
  return __builtin_bit_cast(u32x4, SVS_STREAM_LOAD(reinterpret_cast<const f32x4*>(blk) + kPlaneF4 + t * 64 + lane));
}
// accumulator-layout tile t of a three-byte block: hi fragments of k-steps 2t, 2t+1 + the tile's 16 residual bytes
struct TilePieces3 {
  f16x8 h[2];
  u32x4 m8;
};
template <bool GP>       // GP: hi + mid8; else the hi plane only
__device__ __forceinline__ void load_tile3(const float* __restrict__ blk, int t, int lane, TilePieces3& p) {
  p.h[0] = load_piece(blk, 2 * t, lane); p.h[1] = load_piece(blk, 2 * t + 1, lane);
  if (GP) p.m8 = load_mid8(blk, t, lane);
}
template <bool GP>
__device__ __forceinline__ float grad3_at(const TilePieces3& p, int r) {
  const float hi = (float)p.h[r >> 3][r & 7];
  if (!GP) return hi;
  u32x2 m; m[0] = p.m8[2 * (r >> 3)]; m[1] = p.m8[2 * (r >> 3) + 1];
  return hi + mid8_value(p.h[r >> 3], m, r & 7);
}

// Fragment of k-step s of a vector given in natural row order (vec[q], q < n, zero beyond), in the BLOCK convention
// (row 16 s + 8 (j >> 2) + 4 half + (j & 3)) -- NOT the order split_pe() uses for the layer-0 operand (16 s + 8 half + j,
// which the packed layer-0 weights follow): what a weight-gradient GEMM reads as its B operand must be in block order.
template <int N>
__device__ __forceinline__ void block_values(const float* vec, int s, int half, float scale, float* v) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int q0 = 16 * s + 8 * (j >> 2) + (j & 3), q1 = q0 + 4;
    const float a0 = q0 < N ? vec[q0 < N ? q0 : 0] : 0.0f;
    const float a1 = q1 < N ? vec[q1 < N ? q1 : 0] : 0.0f;
    v[j] = (half ? a1 : a0) * scale;
  }
}
template <int N>
__device__ __forceinline__ void block_fragment(const float* vec, int s, int half, float scale, f16x8& h, f16x8& m) {
  float v[8];
  block_values<N>(vec, s, half, scale, v);
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
