// The trunk (layers 0..7) of the implicit networks on fp16x2 MFMAs, shared by the foreground SDF network
// (svs_mlp_h2.hip) and the inverted-sphere background network (svs_bg_h2.hip).
#pragma once
#include "svs_mlp_h2_dev.h"
#include "svs_blocks_h2.h"

namespace svs {
namespace mlp {

// Positional encoding of the background network's 4-D points (unit direction, 1/r): PE-10, 84 entries in the
// reference's order [x(4), sin(2^0 x)(4), cos(2^0 x)(4), ...] (embedder.py:10-36), padded with zeros to 96.
struct PosEncBg {
  float v[96];
  __device__ __forceinline__ void compute(float x0, float x1, float x2, float x3) {
    const float xs[4] = {x0, x1, x2, x3};
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = xs[c];
#pragma unroll
    for (int f = 0; f < 10; ++f) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s, co;
        sincosf(xs[c] * (float)(1 << f), &s, &co);
        v[4 + 8 * f + c] = s;
        v[8 + 8 * f + c] = co;
      }
    }
#pragma unroll
    for (int q = 84; q < 96; ++q) v[q] = 0.0f;
  }
};

// geometry of the two implicit networks (network.py:31-62 with dtu.yaml / bmvs.yaml)
struct NetFg {   // d_in 3, PE-6: 39 inputs; lin3 emits 217 rows, rows 217..255 of lin4's input are the PE splice
  typedef PosEnc Pe;
  static constexpr int kSteps0 = 3, kPePad = 40, kChunk0 = kChunk0F4, kSpliceTile = 6, kSpliceLocal = 25;
};
struct NetBg {   // d_in 4, PE-10: 84 inputs; lin3 emits 172 rows, rows 172..255 of lin4's input are the PE splice
  typedef PosEncBg Pe;
  static constexpr int kSteps0 = 6, kPePad = 96, kChunk0 = kBgChunk0F4, kSpliceTile = 5, kSpliceLocal = 12;
};

// --------------------------------------------------------------------------------------------------------------
// SDF trunk, layers 0..7
// --------------------------------------------------------------------------------------------------------------
// Epilogue of one trunk tile, in slices: softplus of accumulator register r of `prev`, the skip splice (layer 3),
// and either the split into the next layer's operand (xn) -- whose pieces are also what hbuf stores (HBUF) -- or the
// float32 copy y8 (last layer: the caller splits and stores h_8).
template <bool HBUF, bool LAST, typename Net = NetFg>
struct TrunkEpi {
  // the last layer keeps h_8 in float32 (y8: the sdf head and ghat_7 need it); with HBUF it ALSO splits it into xn -- the
  // feature head's operand -- since the pieces are what hbuf stores
  static constexpr bool kSplit = !LAST || HBUF;
  f32x16 prev;
  SoftplusA sa;
  float v8[8];
  Pieces2* xn;
  f32x16* y8;
  const typename Net::Pe* pe;
  float* hb;        // this layer's block of the wave's hbuf tile
  int lane, half;
  bool splice;      // layer 3: rows >= 217 of the output are the PE splice (network.py:80-81)

  // softplus of element r in three slices, one per MFMA gap of a k-step (a gap hides about 24 cycles of vector issue,
  // an exp2 or log2 costs 8): a = exp2, a2 = max + log2, b = the final fma and the element's share of emit()
  __device__ __forceinline__ void a(int r) {
#if SVS_ABL & 1
    sa.lg = 0.0f;
#else
    sa.lg = __builtin_amdgcn_exp2f(__builtin_fabsf(prev[r]) * (-100.0f * 1.44269504088896341f));
#endif
    pin(sa.lg);
  }
  __device__ __forceinline__ void a2(int r) {
#if SVS_ABL & 1
    sa.mx = prev[r];
#else
    // max(a, 0) without the canonicalising v_max hipcc puts in front of fmaxf in IEEE mode (a is an MFMA result)
    // (volatile + first: hipcc pads an inline-asm instruction that directly precedes an MFMA with an s_nop)
    asm volatile("v_max_f32 %0, 0, %1" : "=v"(sa.mx) : "v"(prev[r]));
    sa.lg = __builtin_amdgcn_logf(1.0f + sa.lg);
#endif
    pin(sa.lg);
  }
  template <bool DEFER = false>
  __device__ __forceinline__ void b(int tp, int r) {
    float v = softplus100_b(sa);
    if (tp == Net::kSpliceTile && splice) {
      // the partial tile: local rows >= kSpliceLocal carry PE[32 * (7 - tile) + local - kSpliceLocal]
      constexpr int base = 32 * (7 - Net::kSpliceTile);
      const int l0 = rho(r) - Net::kSpliceLocal, l1 = rho(r) + 4 - Net::kSpliceLocal;
      if (l0 >= 0 || l1 >= 0) {
        const float v0 = l0 >= 0 ? pe->v[base + (l0 >= 0 ? l0 : 0)] : v;
        const float v1 = l1 >= 0 ? pe->v[base + (l1 >= 0 ? l1 : 0)] : v;
        v = half ? v1 : v0;
      }
    }
    pin(v);
    emit<DEFER>(tp, r, v);
  }
  // DEFER (sliced epilogues): the split of elements 8..15 -- 24 instructions that would sit in front of
  // the tile's barrier -- is left to finish(tp), which the NEXT tile issues while it waits for its first LDS reads.
  // HBUF: h is stored as a PAIR block (svs_blocks_h2.h) -- the very pieces the split produces for the next layer.
  // Sliced epilogues issue those stores from late_store() (behind the tile's last LDS-DMA piece), the others at once.
  template <bool DEFER = false>
  __device__ __forceinline__ void emit(int tp, int r, float v) {
    if (LAST) y8[tp][r] = v;
    if (kSplit && !(SVS_ABL & 2)) {
      v8[r & 7] = v;
      if ((r & 7) == 7 && !(DEFER && r == 15)) {
        const int k = 2 * tp + (r >> 3);
        split8(v8, xn->h[k], xn->m[k]);
        pin(xn->h[k], xn->m[k]);
        if (HBUF && !DEFER) { store_piece(hb, k, lane, xn->h[k], 0); store_piece(hb, k, lane, xn->m[k], 1); }
      }
    }
  }
  template <bool STORE = false>
  __device__ __forceinline__ void finish(int tp) {
    if (kSplit && !(SVS_ABL & 2)) {
      const int k = 2 * tp + 1;
      split8(v8, xn->h[k], xn->m[k]);
      pin(xn->h[k], xn->m[k]);
      if (HBUF && STORE) { store_piece(hb, k, lane, xn->h[k], 0); store_piece(hb, k, lane, xn->m[k], 1); }
    }
  }
  // k-step s of tile t (whose MFMAs cover the epilogue of tile t-1): the pair stores of the pieces that are complete by
  // now -- k-step 2(t-1) (split in this tile's k-step 7) and k-step 2(t-2)+1 (split by finish(t-2) in front of this
  // tile) -- behind the last LDS-DMA piece (k-step 8), so that Stream::advance_keep leaves them in flight
  __device__ __forceinline__ void late_store(int t, int s) {
    if (!HBUF) return;
    const int k1 = 2 * (t - 1), k2 = 2 * (t - 2) + 1;
    if (s == 9) store_piece(hb, k1, lane, xn->h[k1], 0);
    if (s == 11) store_piece(hb, k1, lane, xn->m[k1], 1);
    if (t >= 2 && s == 13) store_piece(hb, k2, lane, xn->h[k2], 0);
    if (t >= 2 && s == 15) store_piece(hb, k2, lane, xn->m[k2], 1);
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { a(r); a2(r); b(tp, r); }
  }
  // layer 3, the tiles behind the partial one are pure PE (no MFMA): tile tp = PE[32 * (7 - tp) + local]
  __device__ __forceinline__ void splice_full_tiles() {
#pragma unroll
    for (int tp = Net::kSpliceTile + 1; tp < 8; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) emit(tp, r, half ? pe->v[32 * (7 - tp) + rho(r) + 4] : pe->v[32 * (7 - tp) + rho(r)]);
  }
};

// one 256 -> 256 trunk layer (l >= 1).  On entry the layer's first chunk is current; on return the next layer's is.
template <bool HBUF, bool LAST, typename Net = NetFg>
__device__ __forceinline__ void trunk_layer_h2(Stream& st, const Pieces2& x, TrunkEpi<HBUF, LAST, Net>& ep, int lane) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t == Net::kSpliceTile + 1 && ep.splice) break;   // lin3 has 217 (bg: 172) outputs; the tiles behind are the PE splice
    f32x16 acc;
    // the next chunk's 9 LDS-DMA pieces go behind k-steps 0..8 (Stream::prefetch_step); the hbuf stores (TrunkEpi::
    // late_store) behind them, in k-steps 9, 11, 13, 15: younger than every piece, they may stay in flight
    if (t == 0) acc = tile_mma_h2_pf<16, kChunkF4>(st, x, lane, NoEpi(), NoEpi());
    // (the split of tile t-2's last eight elements is issued while this tile waits for its first LDS reads)
    else acc = tile_mma_h2_pf<16, kChunkF4>(st, x, lane, [&](int s) { ep.a(s); }, [&](int s) { ep.a2(s); },
                                            [&](int s) { ep.template b<true>(t - 1, s); ep.late_store(t, s); },
                                            [&]() { if (t >= 2) ep.finish(t - 2); });
    ep.prev = acc;
    if (HBUF && t > 0) { if (t >= 2) st.advance_keep<4>(); else st.advance_keep<2>(); }
    else st.advance();
  }
  // the tail: the pieces the loop has not stored yet go out at once
  if (ep.splice) {
    ep.template finish<true>(Net::kSpliceTile - 1); ep.all(Net::kSpliceTile); ep.splice_full_tiles();
  } else { ep.template finish<true>(6); ep.all(7); }
}

// Forward through layers 0..7.  x: scratch operand; on return y8 holds h_8 in float32 (the input of lin8) and the
// current chunk is the one that follows the trunk in the stream.  HBUF: h_1..h_8 are also stored to hbuf (pair blocks)
// and x holds the pieces of h_8.
template <bool HBUF, typename Net = NetFg>
__device__ __forceinline__ void forward_trunk_h2(Stream& st, Pieces2& x, Pieces2& xn, f32x16* y8, const typename Net::Pe& pe,
                                                 int lane, int half, float* __restrict__ hbuf) {
  split_pe<Net::kSteps0, Net::kPePad>(pe.v, half, x);
  st.advance();        // chunk 0 (prefetched by the caller before the positional encoding)
  {
    // ---- layer 0 : 39(48) -> 256, three k-steps per tile: the epilogue of tile t-1 follows tile t's MFMAs
    TrunkEpi<HBUF, false, Net> ep;
    ep.xn = &xn; ep.y8 = nullptr; ep.pe = &pe; ep.hb = hbuf; ep.lane = lane; ep.half = half; ep.splice = false;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t < 7) st.prefetch<Net::kChunk0>(); else st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma_h2<Net::kSteps0>(st.cur_buf(), x, lane);
      if (t > 0) ep.all(t - 1);
      ep.prev = acc;
      st.advance();
    }
    ep.all(7);
  }
  // ---- layers 1..6 (layer 3 emits 217 rows + the skip splice)
  // operands ping-pong between xn and x, two layers per loop iteration (copying xn back to x instead, for one layer body,
  // cost 128 v_mov per layer: 3 % of the trunk)
  for (int l = 1; l < 7; l += 2) {
    TrunkEpi<HBUF, false, Net> ep;
    ep.y8 = nullptr; ep.pe = &pe; ep.hb = hbuf + (size_t)l * block_stride(); ep.lane = lane; ep.half = half; ep.splice = l == 3;
    ep.xn = &x; trunk_layer_h2<HBUF, false, Net>(st, xn, ep, lane);
    TrunkEpi<HBUF, false, Net> ep2;
    ep2.y8 = nullptr; ep2.pe = &pe; ep2.hb = hbuf + (size_t)(l + 1) * block_stride(); ep2.lane = lane; ep2.half = half; ep2.splice = false;
    ep2.xn = &xn; trunk_layer_h2<HBUF, false, Net>(st, x, ep2, lane);
  }
  // ---- layer 7: input in xn (layer 6 wrote it), output kept in float32 (y8) and, with HBUF, as pieces in x
  TrunkEpi<HBUF, true, Net> ep;
  ep.xn = HBUF ? &x : nullptr; ep.y8 = y8; ep.pe = &pe; ep.hb = hbuf + (size_t)7 * block_stride(); ep.lane = lane; ep.half = half; ep.splice = false;
  trunk_layer_h2<HBUF, true, Net>(st, xn, ep, lane);
}


}  // namespace mlp
}  // namespace svs
