// fp16x2 variants of the training-backward sweeps of svs_mlp_bwd.hip (same algebra, same buffers): every layer
// product runs on v_mfma_f32_32x32x16_f16 with two-piece fp16 operands (svs_mlp_h2_dev.h), the previous tile's
// epilogue interleaved with the current tile's MFMAs.
//
// Gradients span many orders of magnitude and are typically far below fp16's range, so every point (= MFMA column
// = lane) carries its own power-of-two scale: an operand is split as true * s with s chosen so that the point's
// largest element is ~2^4, and the accumulator is multiplied by 1/s in the epilogue (both exact).  The scale used
// to split the operand a layer produces is derived from the maximum of the operand the layer consumed (the new
// maximum is only known once all eight tiles are done), which leaves 2^11 of headroom for growth across one layer;
// additive external inputs (a2 in pass B) enter through a per-point floor of the maximum.  What the sweeps write to
// memory are these scaled operands with their per-point scale (scaled blocks, svs_blocks_h2.h), in the launch's format GP:
// both pieces (the default: gradients in the float32 accuracy class) or the hi piece only (half the bytes); forward
// activations are read back from pair blocks -- with GP = false by their hi plane where only softplus' is needed.  The
// kernels also publish the global maxima of the gradient-like
// operands of the weight-gradient GEMMs (svs_wgrad.hip, fp16x2 path) with one atomic max per wave.
#include "svs_mlp_h2_dev.h"
#include "svs_mlp_host.h"
#include "svs_mlp_bwd_args.h"
#include "svs_mlp_h2_trunk.h"
#include "svs_mlp_bwd_h2_dev.h"
#include "svs_ticket.h"
#include <cstdlib>

namespace svs {
namespace mlp {

// ==============================================================================================================
// radiance MLP backward
// ==============================================================================================================
// zbar_{l-1} = rbar_l * [r_l > 0] for one tile: track, split; the pieces of the split (value * s_out) are what
// zbuf stores (a scaled block, svs_blocks_h2.h)
template <bool GP>
struct RgbBwdEpi {
  f32x16 prev;
  TilePieces r;        // hi pieces of the stored r_l tile: only its sign is needed
  float v8[8];
  Pieces2* out;
  float* zblk;
  PointScale* ps;
  int lane;
  __device__ __forceinline__ void b(int tp, int rr) {
    float v = prev[rr] * ps->inv_in;
    v = hi_at(r, rr) > 0.0f ? v : 0.0f;
    pin(v);
    ps->track(v);
    v8[rr & 7] = v * ps->s_out;
    if ((rr & 7) == 7) {
      split8(v8, out->h[2 * tp + (rr >> 3)], out->m[2 * tp + (rr >> 3)]);
      pin(out->h[2 * tp + (rr >> 3)], out->m[2 * tp + (rr >> 3)]);
    }
    // stores behind the LDS-DMA pieces: k-step 9 (the piece split at element 7) and 15
    if (rr == 9) store_piece(zblk, 2 * tp, lane, out->h[2 * tp], 0);
    if (GP && rr == 11) store_piece(zblk, 2 * tp, lane, out->m[2 * tp], 1);
    if (rr == 15) store_grad<GP>(zblk, 2 * tp + 1, lane, out->h[2 * tp + 1], out->m[2 * tp + 1]);
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) b(tp, rr);
  }
};

// rbar_l = W_l^T zbar_l fused with zbar_{l-1} (l = 3..1)
template <bool GP>
__device__ __forceinline__ void rgb_bwd_layer_h2(Stream& st, const Pieces2& in, Pieces2& out, const float* rblk, float* zblk,
                                                 float* zrec, PointScale& ps, int lane) {
  RgbBwdEpi<GP> ep;
  ep.out = &out; ep.zblk = zblk; ep.ps = &ps; ep.lane = lane;
  store_record(zrec, lane, ps.s_out, 0.0f);
  // Per tile: the next chunk's 9 LDS-DMA pieces behind k-steps 0..8 (Stream::prefetch_step), then -- younger than every
  // piece, so that the barrier leaves them in flight -- the 2 zbuf stores of tile t-1's epilogue (k-steps 9, 15) and the
  // 2 loads of r tile t+1 (k-steps 10, 12), which the epilogue of tile t+1 consumes during tile t+2.
  TilePieces rnext;
  load_tile_hi(rblk, 0, lane, rnext);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const TilePieces rcur = rnext;
    auto rload = [&](int s) {
      if (t == 7) return;
      if (s == 10) rnext.h[0] = load_piece(rblk, 2 * (t + 1), lane);
      if (s == 12) rnext.h[1] = load_piece(rblk, 2 * (t + 1) + 1, lane);
    };
    f32x16 acc;
    if (t == 0) acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, NoEpi(), NoEpi(), rload);
    else acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, NoEpi(), [&](int s) { ep.b(t - 1, s); }, rload);
    ep.prev = acc; ep.r = rcur;
    // in flight across the barrier: the 2 r loads (t < 7) and the zbuf stores of tile t-1's epilogue (2, with GP 4)
    constexpr int kSt = GP ? 4 : 2;
    if (t == 0) st.advance_keep<2>();
    else if (t < 7) st.advance_keep<2 + kSt>();
    else st.advance_keep<kSt>();
  }
  ep.all(7);
  ps.next();
}

template <bool GP>
__global__ __launch_bounds__(kThreads, 1) void rgb_bwd_h2_kernel(RgbBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const bool livep = p < a.P;
  const int pc = livep ? p : a.P - 1;
  const size_t LS = block_stride();                    // rbuf, zbuf: [block][wave tile]
  const float* rb = a.rbuf + (size_t)wtile * kBlockF;
  float* zb = a.zbuf + (size_t)wtile * kBlockF;
  const size_t T = (size_t)gridDim.x * kWaves;
  auto zrec = [&](int l) { return record_ptr(a.zbuf, 5, T, l, wtile); };

  st.prefetch<kW4TF4>();
  float dz[3];
  float m0 = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float o = a.rgb[3 * pc + c];
    dz[c] = (livep && half == 0) ? a.d_rgb[3 * pc + c] * o * (1.0f - o) : 0.0f;   // through the sigmoid (network.py:189)
    m0 = __builtin_fmaxf(m0, __builtin_fabsf(dz[c]));
  }
  m0 = __builtin_fmaxf(m0, __shfl_xor(m0, 32));
  PointScale ps;
  ps.start(m0, 0.0f);
  {  // zbar_4: rows 0..2 = elements 0..2 of k-step 0 of half 0; the rest of the block stays zero (caller zeroes once)
    float v8[8] = {dz[0] * ps.s_in, dz[1] * ps.s_in, dz[2] * ps.s_in, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    store_grad8<GP>(zb + 4 * LS, 0, lane, v8);
    store_record(zrec(4), lane, ps.s_in, m0);
  }
  st.advance();
  Pieces2 pa, pb;
  // rbar_4 = W_4^T zbar_4 (K = 3: float32 MFMA from the short W4T chunk), masked by r_4 > 0 -> zbar_3
  st.prefetch<kChunkF4>();
  {
    const f32x4* c = st.cur_buf();
    const float* r4 = rb + 3 * LS;
    float* z3 = zb + 3 * LS;
    store_record(zrec(3), lane, ps.s_out, 0.0f);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      TilePieces r;
      load_tile_hi(r4, t, lane, r);
      const f32x4 w = c[t * 64 + lane];
      f32x16 acc = (f32x16)(0.0f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], dz[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], dz[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], dz[2], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[i] = hi_at(r, i) > 0.0f ? acc[i] : 0.0f; ps.track(acc[i]); }
      split_tile_scaled(acc, t, pa, ps.s_out);
      store_grad<GP>(z3, 2 * t, lane, pa.h[2 * t], pa.m[2 * t]);
      store_grad<GP>(z3, 2 * t + 1, lane, pa.h[2 * t + 1], pa.m[2 * t + 1]);
    }
  }
  ps.next();
  st.advance();
  // layers 3..1
  rgb_bwd_layer_h2<GP>(st, pa, pb, rb + 2 * LS, zb + 2 * LS, zrec(2), ps, lane);
  rgb_bwd_layer_h2<GP>(st, pb, pa, rb + 1 * LS, zb + 1 * LS, zrec(1), ps, lane);
  rgb_bwd_layer_h2<GP>(st, pa, pb, rb, zb, zrec(0), ps, lane);
  // layer 0: the input gradients from zbar_0 (feature rows: tiles 0..7, extras: tile 8).  fbar is stored as a scaled block
  // under the scale predicted from zbar_0's maximum (ps.s_out, as between any two layers); its record -- pass B takes
  // the scale and the maximum from it -- is written once the maximum is known
  float* fb = a.feat_bar + (size_t)wtile * kBlockF;
  float fmax = 0.0f;
  const float s_f = ps.s_out;
  {
    f32x16 prev;
    float v8[8];
    auto slice = [&](int tp, int r) {
      const float v = prev[r] * ps.inv_in;
      fmax = __builtin_fmaxf(fmax, __builtin_fabsf(v));
      v8[r & 7] = v * s_f;
      if ((r & 7) == 7) store_grad8<GP>(fb, 2 * tp + (r >> 3), lane, v8);
    };
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t < 8) st.prefetch<kChunkF4>();
      f32x16 acc;
      if (t == 0) acc = tile_mma_h2<16>(st.cur_buf(), pb, lane);
      else acc = tile_mma_h2<16>(st.cur_buf(), pb, lane, NoEpi(), [&](int s) { slice(t - 1, s); });
      if (t < 8) { prev = acc; st.advance(); }
      else if (half == 1 && livep) {
        // extra rows 12,13,14 (normals, network.py:175) = local rows rho(4..6)+4 of the extras tile
        a.d_normals[3 * p] = acc[4] * ps.inv_in; a.d_normals[3 * p + 1] = acc[5] * ps.inv_in;
        a.d_normals[3 * p + 2] = acc[6] * ps.inv_in;
      }
    }
  }
  fmax = __builtin_fmaxf(fmax, __shfl_xor(fmax, 32));
  store_record(record_ptr(a.feat_bar, 1, T, 0, wtile), lane, s_f, fmax);
  publish_max(a.absmax + 1, ps.gmax);
  publish_max(a.absmax + 2, fmax);
}

// ==============================================================================================================
// SDF MLP backward, pass A: u_0 = J_PE nbar;  v_l = W_l u_l;  u_{l+1} = v_l s'(a_l);  a2_l = v_l g(h_{l+1}) s''(a_l)
// ==============================================================================================================
constexpr int kSpliceF = 20;     // floats per lane kept in LDS for the skip splice of pass A (tile 7 + 4 registers of tile 6)

// Since round 5 pass A no longer forms the second-order blocks a2_l = v_l ghat_l 100 (1 - s'(a_l)): with u_{l+1} = v_l s'(a_l)
// they are u_{l+1} ghat_l 100 (1 - s') / s', and pass B -- which reads h_{l+1} for s' anyway -- re-forms them from the stored
// u_{l+1} and ghat_l blocks.  Pass A then neither reads gbuf (8 KB per point) nor writes a2buf (8 KB); pass B reads ubuf and
// gbuf instead of a2buf (+8 KB): 8 KB per point less, and this kernel's GP instance no longer spills.  What pass B still
// needs from here is a bound of max_r |a2_l| per point for its scale floor: the factor max_r |v_l| 100 (1 - s'(a_l)) goes into
// the second half of u_{l+1}'s record, the other factor, max_r |ghat_l|, is in gbuf's records (sdf_full_h2_kernel).
template <bool SPLIT, bool GP>
struct PassAEpi {
  f32x16 prev;
  TilePieces h;       // the stored h_{l+1} tile (softplus' only): both pieces with GP, else the hi pieces
  float v, s1;
  float v8[8];
  f16x8 up[2], uq[2]; // the pieces of u being stored (uq: mid, GP)
  Pieces2* out;       // u_{l+1} as the next operand (SPLIT)
  float* ublk;        // u_{l+1} block (scaled block: value * s_out)
  float* urec;        // its record: [scale][max_r |v_l| 100 (1 - s'(a_l))]
  PointScale* ps;
  const float* splice;  // LDS: this lane's u_0 splice values, [kSpliceF][kThreads]
  float vm;           // running max |v_l| 100 (1 - s')
  int lane, half;
  bool l3;            // layer 3: the skip connection carries u_0 into rows >= 217 of u_4 (no second-order term there)

  __device__ __forceinline__ void begin() {
    if (lane < 32) urec[lane] = ps->s_out;
    vm = 0.0f;
  }
  __device__ __forceinline__ void end() {        // the record's second half, once the layer's maximum is known
    vm = __builtin_fmaxf(vm, __shfl_xor(vm, 32));
    if (lane >= 32) urec[lane] = vm;
  }
  __device__ __forceinline__ void a(int r) {
    v = prev[r] * ps->inv_in;
    s1 = 1.0f - __builtin_amdgcn_exp2f(grad_times<GP>(h, r, -100.0f * 1.44269504088896341f));   // s'(a) from h (svs_mlp_dev.h)
    pin(v); pin(s1);
  }
  template <bool LATE = false>     // LATE: the stores wait for store_slot() (behind the tile's LDS-DMA pieces)
  __device__ __forceinline__ void b(int tp, int r) {
    float u = v * s1;
    float w = __builtin_fabsf(v) * (100.0f - 100.0f * s1);
    if (tp == 6 && r >= 12 && l3) {
      // local rows 25..31 of tile 6 (registers 13..15 of half 0, 12..15 of half 1) carry u_0[32..38]
      const bool sp = half == 1 || r >= 13;
      const float us = splice[(16 + (r - 12)) * kThreads];
      u = sp ? us : u;
      w = sp ? 0.0f : w;
    }
    pin(u); pin(w);
    emit<LATE>(tp, r, u, w);
  }
  template <bool LATE = false>
  __device__ __forceinline__ void emit(int tp, int r, float u, float w) {
    vm = __builtin_fmaxf(vm, w);
    ps->track(u);
    v8[r & 7] = u * ps->s_out;
    if ((r & 7) == 7) {
      const int k = 2 * tp + (r >> 3), q = r >> 3;
      if (SPLIT) {
        split8(v8, out->h[k], out->m[k]);
        pin(out->h[k], out->m[k]);
        up[q] = out->h[k];
        if (GP) uq[q] = out->m[k];
      } else if (GP) {
        split8(v8, up[q], uq[q]);
      } else {
        up[q] = hi8(v8, 1.0f);
      }
#ifdef SVS_EXP_U_HI       // experiment: a one-piece STORED u block (the in-register operand of the next layer keeps both)
      if (GP) uq[q] = (f16x8)(_Float16)0;
#endif
#ifdef SVS_EXP_U_FP8      // experiment: the stored mid piece rounded to 3 mantissa bits (what an e4m3 mid piece would keep)
      {
        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
        u16x8 bb = __builtin_bit_cast(u16x8, uq[q]); bb = (bb + (unsigned short)0x40) & (unsigned short)0xFF80;
        if (GP) uq[q] = __builtin_bit_cast(f16x8, bb);
      }
#endif
#if !(SVS_ABL & 2048)  // diagnostic: no u stores
      if (!LATE) store_grad<GP>(ublk, k, lane, up[q], uq[q]);
#endif
    }
  }
  // k-step s of the tile whose MFMAs cover this epilogue: the two (GP: four) stores of tile tp behind the last LDS-DMA piece
  __device__ __forceinline__ void store_slot(int tp, int s) {
#if SVS_ABL & 2048
    return;
#endif
    if (s == 9) store_piece(ublk, 2 * tp, lane, up[0], 0);
    if (GP && s == 11) store_piece(ublk, 2 * tp, lane, uq[0], 1);
    if (s == 15) store_grad<GP>(ublk, 2 * tp + 1, lane, up[1], uq[1]);
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { a(r); b(tp, r); }
  }
  __device__ __forceinline__ void splice_tile7() {   // layer 3, tile 7: u = u_0[0..31] (no MFMA)
#pragma unroll
    for (int r = 0; r < 16; ++r) emit(7, r, splice[r * kThreads], 0.0f);
  }
};

// layer l >= 1 of pass A; hblk: block l of hbuf.  LAST: no chunk follows the layer's last one.
template <bool SPLIT, bool LAST, bool GP>
__device__ __forceinline__ void pass_a_layer_h2(Stream& st, const Pieces2& in, PassAEpi<SPLIT, GP>& ep, const float* hblk, int lane) {
  // Per tile: the side tile h of tile t (two fragments: the hi plane; four with GP) is requested in front of it (its
  // epilogue runs during tile t+1); the next chunk's LDS-DMA pieces go behind k-steps 0..8 (Stream::prefetch_step); the u
  // stores of tile t-1's epilogue are issued in k-steps 9 .. 15: younger than every piece, they stay in flight across the
  // tile's barrier.
  ep.begin();
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t == 7 && ep.l3) break;
    TilePieces hload;
#if SVS_ABL & 1024     // diagnostic: no side-tile loads
    hload = ep.h;
#else
    load_tile_grad<GP>(hblk, t, lane, hload);
#endif
    const bool fetch = !(LAST && t == 7);
    f32x16 acc;
    if (!fetch) acc = tile_mma_h2<16>(st.cur_buf(), in, lane, [&](int s) { ep.a(s); }, [&](int s) { ep.b(t - 1, s); });
    else if (t == 0) acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, NoEpi(), NoEpi());
    else acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, [&](int s) { ep.a(s); },
                                            [&](int s) { ep.template b<true>(t - 1, s); ep.store_slot(t - 1, s); });
    ep.prev = acc; ep.h = hload;
    if (fetch) {
      if (t == 0) st.advance();
      else st.advance_keep<GP ? 4 : 2>();
    }
  }
  if (ep.l3) { ep.all(6); ep.splice_tile7(); }
  else ep.all(7);
  ep.end();
  ep.ps->next();
}

template <bool GP>
__global__ __launch_bounds__(kThreads, 1) void sdf_bwd_a_h2_kernel(SdfBwdAArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  float* splice = reinterpret_cast<float*>(smem + kLdsBytes) + threadIdx.x;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.src.P ? p : a.src.P - 1;
#ifdef SVS_EXP_STAGGER    // experiment: odd workgroups start SVS_EXP_STAGGER x 2 us late (are the sweeps phase-locked chip-wide?)
  if (blockIdx.x & 1) for (int i = 0; i < SVS_EXP_STAGGER; ++i) __builtin_amdgcn_s_sleep(64);
#endif

  st.prefetch<kChunk0F4>();
  PointScale ps;
  Pieces2 pa, pb;
  const size_t LS = block_stride();
  const float* hb = a.hbuf + (size_t)wtile * kBlockF;
  float* ub = a.ubuf + (size_t)wtile * kBlockF;
  const size_t T = (size_t)gridDim.x * kWaves;
  auto urec = [&](int l) { return record_ptr(a.ubuf, 9, T, l, wtile); };
  {
    float x0, x1, x2;
    load_point(a.src, p, x0, x1, x2);
    PosEnc pe;
    pe.compute(x0, x1, x2);
    float nb[3] = {a.d_grad[3 * pc], a.d_grad[3 * pc + 1], a.d_grad[3 * pc + 2]};
    if (p >= a.src.P || (a.clamp_mask && a.clamp_mask[pc])) { nb[0] = nb[1] = nb[2] = 0.0f; }
    // u_0[q] = d PE_q / d x_{c(q)} * nbar_{c(q)}
    PosEnc u0;
    float m0 = 0.0f;
#pragma unroll
    for (int q = 0; q < 40; ++q) {
      float coef = 1.0f; int c = q;
      if (q >= 3 && q < kPeDim) {
        const int f = (q - 3) / 6, w = (q - 3) % 6;
        const float sc = (float)(1 << f);
        c = w < 3 ? w : w - 3;
        coef = w < 3 ? sc * pe.v[q + 3] : -sc * pe.v[q - 3];
      }
      u0.v[q] = q < kPeDim ? coef * nb[c < 3 ? c : 0] : 0.0f;
      m0 = __builtin_fmaxf(m0, __builtin_fabsf(u0.v[q]));
    }
    ps.start(m0, 0.0f);
    // the skip splice of layer 3, parked in LDS: [0..15] tile 7 = u_0[rho(r) (+4)], [16..19] registers 12..15 of
    // tile 6 = u_0[32 + rho(r) (+4) - 25]
#pragma unroll
    for (int r = 0; r < 16; ++r) splice[r * kThreads] = half ? u0.v[rho(r) + 4] : u0.v[rho(r)];
#pragma unroll
    for (int r = 12; r < 16; ++r) {
      const int k0 = rho(r) - 25, k1 = rho(r) + 4 - 25;
      const float s0 = k0 >= 0 ? u0.v[32 + (k0 >= 0 ? k0 : 0)] : 0.0f;
      splice[(16 + r - 12) * kThreads] = half ? u0.v[32 + k1] : s0;
    }
    PosEnc us;
#pragma unroll
    for (int q = 0; q < 40; ++q) us.v[q] = u0.v[q] * ps.s_in;
    split_pe(us, half, pa);
    // u_0 (scaled block, scale s_in) and h_0 = PE (PAIR block) in PE order, as block fragments: k-steps 0..2 hold the 39
    // rows, the rest of both blocks stays zero (the caller zeroes them once): the B operands of the two products of dW_0
    float* pbk = a.pebuf + (size_t)wtile * kBlockF;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      f16x8 fh, fm;
      block_fragment<kPeDim>(u0.v, k, half, ps.s_in, fh, fm);
      store_grad<GP>(ub, k, lane, fh, fm);
      block_fragment<kPeDim>(pe.v, k, half, 1.0f, fh, fm);
      store_grad<GP>(pbk, k, lane, fh, fm);      // (GP = false: the weight gradient reads hi planes only)
    }
    store_record(urec(0), lane, ps.s_in, m0);
  }
  st.advance();

  {
    // ---- layer 0 (three k-steps per tile: the epilogue of tile t-1 follows tile t's MFMAs)
    PassAEpi<true, GP> ep;
    ep.out = &pb; ep.ublk = ub + LS; ep.urec = urec(1); ep.ps = &ps; ep.splice = splice;
    ep.lane = lane; ep.half = half; ep.l3 = false;
    ep.begin();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      TilePieces hload;
      load_tile_grad<GP>(hb, t, lane, hload);
      if (t < 7) st.prefetch<kChunk0F4>(); else st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma_h2<3>(st.cur_buf(), pa, lane);
      if (t > 0) ep.all(t - 1);
      ep.prev = acc; ep.h = hload;
      st.advance();
    }
    ep.all(7);
    ep.end();
    ps.next();
  }
  // ---- layers 1..6: pb -> pa, copied back (one code body for all layers)
  for (int l = 1; l < 7; ++l) {
    PassAEpi<true, GP> ep;
    ep.out = &pa; ep.ublk = ub + (size_t)(l + 1) * LS; ep.urec = urec(l + 1);
    ep.ps = &ps; ep.splice = splice; ep.lane = lane; ep.half = half; ep.l3 = l == 3;
    pass_a_layer_h2<true, false, GP>(st, pb, ep, hb + (size_t)l * LS, lane);
#pragma unroll
    for (int s = 0; s < 16; ++s) { pb.h[s] = pa.h[s]; pb.m[s] = pa.m[s]; }
  }
  // ---- layer 7: u_8 (needed for the row-0 gradient of lin8 and for pass B's a2_7) is stored, not split
  {
    PassAEpi<false, GP> ep;
    ep.out = nullptr; ep.ublk = ub + (size_t)8 * LS; ep.urec = urec(8);
    ep.ps = &ps; ep.splice = splice; ep.lane = lane; ep.half = half; ep.l3 = false;
    pass_a_layer_h2<false, true, GP>(st, pb, ep, hb + (size_t)7 * LS, lane);
  }
  publish_max(a.absmax, ps.gmax);
}

// ==============================================================================================================
// SDF MLP backward, pass B: hbar_8 = sbar W_8[0,:] + W_8[1:,:]^T fbar;  abar_l = hbar_{l+1} s'(a_l) + a2_l;
//                           hbar_l = W_l^T abar_l
// ==============================================================================================================
// A2 (the foreground network): the second-order source a2_l = v_l ghat_l 100 (1 - s'(a_l)) of svs_mlp_bwd.hip is re-formed here
// from what pass A stored, u_{l+1} = v_l s'(a_l):  a2_l = u_{l+1} ghat_l * 100 (1 - s') / s'  (s' from the h_{l+1} tile the
// stage reads anyway; s' = 0 only where softplus underflowed to h = 0, and there u is 0 too: the quotient is guarded, the
// product is 0).  Absolute errors stay those of the stored blocks: an error du of u becomes du ghat 100 (1 - s') / s' =
// du * 100 g(h_{l+1}) (1 - s'), the size an error of v itself would have.
template <bool FIRST, bool SPLIT, typename Net, bool A2, bool GP>
struct PassBEpi {
  static constexpr bool kG = A2 || FIRST;      // the stage reads a ghat tile (A2: ghat_l; FIRST: ghat_7 = W8[0,:] s'(a_7))
  f32x16 prev;
  TilePieces h, u, g;      // h_{l+1} (softplus' only), u_{l+1} (value * its block's scale), ghat_l: both pieces with GP, else hi
  float v, s1, q;
  float v8[8];
  f16x8 ap[2], aq[2];      // the pieces of abar being stored (aq: mid, GP)
  Pieces2* out;
  float* ablk;             // abar_l block (scaled block: value * s_out)
  float* arec;             // its record
  PointScale* ps;
  float sbar, u_inv100;    // u_inv100: 100 / (scale the u block was stored under)
  int lane, half;
  bool l4;            // producing abar_3: rows >= 217 of h_4 are the PE splice
#if SVS_ABL & 65536
  unsigned* cyc;
#endif
  __device__ __forceinline__ void a(int r) {
    v = prev[r] * ps->inv_in;
#if SVS_ABL & 32768    // diagnostic: no softplus' arithmetic
    float e = 0.5f;
#else
    float e = __builtin_amdgcn_exp2f(grad_times<GP>(h, r, -100.0f * 1.44269504088896341f));   // 1 - s'
#endif
    // (A2: s' >= 2^-24 keeps the quotient of b() finite; where softplus' underflowed u is 0 and so is the product)
    if (A2) e = __builtin_fminf(e, 0.99999994f);
    s1 = 1.0f - e;
    if (A2) q = e * u_inv100;
    pin(v); pin(s1);
    if (A2) pin(q);
  }
  template <bool LATE = false>
  __device__ __forceinline__ void b(int tp, int r) {
    float o = v * s1;
#if SVS_ABL & 16384    // diagnostic: no second-order / sbar terms in the epilogue
    if (false) {
#else
    if (A2) {
#endif
      const float gg = grad_mix<GP>(g, r);
      const float t = grad_times<GP>(u, r, gg);
      o = __builtin_fmaf(t, q * __builtin_amdgcn_rcpf(s1), o);
      if (FIRST) o = __builtin_fmaf(sbar, gg, o);
    } else if (FIRST) {
      o = __builtin_fmaf(sbar, grad_mix<GP>(g, r), o);      // ghat_7 = W8[0,:] s'(a_7)
    }
    if (l4 && tp > Net::kSpliceTile) o = 0.0f;
    if (l4 && tp == Net::kSpliceTile) {
      const bool z0 = rho(r) >= Net::kSpliceLocal, z1 = rho(r) + 4 >= Net::kSpliceLocal;
      if (z0 || z1) { if (half ? z1 : z0) o = 0.0f; }
    }
    pin(o);
    ps->track(o);
    v8[r & 7] = o * ps->s_out;
    if ((r & 7) == 7) {
      const int k = 2 * tp + (r >> 3), q2 = r >> 3;
      if (SPLIT) {
        split8(v8, out->h[k], out->m[k]);
        pin(out->h[k], out->m[k]);
        ap[q2] = out->h[k];
        if (GP) aq[q2] = out->m[k];
      } else if (GP) {
        split8(v8, ap[q2], aq[q2]);
      } else {
        ap[q2] = hi8(v8, 1.0f);
      }
#if !(SVS_ABL & 8192)  // diagnostic: no abar stores
      if (!LATE) store_grad<GP>(ablk, k, lane, ap[q2], aq[q2]);
#endif
    }
  }
  __device__ __forceinline__ void store_slot(int tp, int s) {     // behind the tile's LDS-DMA pieces
#if SVS_ABL & 8192
    return;
#endif
    if (s == 9) store_grad<GP>(ablk, 2 * tp, lane, ap[0], aq[0]);
    if (s == 15) store_grad<GP>(ablk, 2 * tp + 1, lane, ap[1], aq[1]);
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { a(r); b(tp, r); }
  }
};

// one stage of pass B: in -> (W^T in) fused with abar_l (h_{l+1} from hblk; A2: u_{l+1} from ublk with its record urec and
// ghat_l from gblk; FIRST without A2: ghat_7 from gblk).  next_floor: the scale floor of the NEXT stage's output.
template <bool FIRST, bool SPLIT, bool LAST_STAGE, typename Net, bool A2, bool GP>
__device__ __forceinline__ void pass_b_stage_h2(Stream& st, const Pieces2& in, PassBEpi<FIRST, SPLIT, Net, A2, GP>& ep, const float* hblk,
                                                const float* ublk, const float* urec, const float* gblk, float next_floor, int lane) {
  // Per tile: the next chunk's LDS-DMA pieces behind k-steps 0..8 (Stream::prefetch_step), then -- younger than every
  // piece, left in flight across the tile's barrier -- the 2 abuf stores of tile t-1's epilogue (k-steps 9, 15; 4 with GP)
  // and the loads of the side tiles (hi planes: two fragments each; with GP both planes: four) h, u, ghat of tile t+1,
  // which the epilogue of tile t+1 consumes during tile t+2.
  constexpr bool kG = A2 || FIRST;
  store_record(ep.arec, lane, ep.ps->s_out, 0.0f);
  ep.u_inv100 = A2 ? 100.0f * PointScale::inv_pow2(load_scale(urec, lane)) : 0.0f;
  TilePieces hnext, unext, gnext;
  load_tile_grad<GP>(hblk, 0, lane, hnext);
  if (A2) load_tile_grad<GP>(ublk, 0, lane, unext);
  if (kG) load_tile_grad<GP>(gblk, 0, lane, gnext);
  constexpr int kPer = GP ? 4 : 2;
  constexpr int kLoads = (SVS_ABL & 4096) ? 0 : kPer * (1 + (A2 ? 1 : 0) + (kG ? 1 : 0));
  constexpr int kStores = (SVS_ABL & 8192) ? 0 : (GP ? 4 : 2);
#if SVS_ABL & 65536    // diagnostic: cycles per tile, MFMA part and wait + barrier part, summed over the stages
  uint64_t t0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const TilePieces hcur = hnext, ucur = unext, gcur = gnext;
    auto side = [&](int s) {
      if (t == 7 || (SVS_ABL & 4096)) return;      // (4096: diagnostic, no side-tile loads)
      if (s == 10) hnext.h[0] = load_piece(hblk, 2 * (t + 1), lane);
      if (s == 11) hnext.h[1] = load_piece(hblk, 2 * (t + 1) + 1, lane);
      if (A2 && s == 12) unext.h[0] = load_piece(ublk, 2 * (t + 1), lane);
      if (A2 && s == 13) unext.h[1] = load_piece(ublk, 2 * (t + 1) + 1, lane);
      if (kG && s == 14) gnext.h[0] = load_piece(gblk, 2 * (t + 1), lane);
      if (kG && s == 15) gnext.h[1] = load_piece(gblk, 2 * (t + 1) + 1, lane);
      if (GP) {      // the mid planes, in the same gaps
        if (s == 10) hnext.m[0] = load_piece(hblk, 2 * (t + 1), lane, 1);
        if (s == 11) hnext.m[1] = load_piece(hblk, 2 * (t + 1) + 1, lane, 1);
        if (A2 && s == 12) unext.m[0] = load_piece(ublk, 2 * (t + 1), lane, 1);
        if (A2 && s == 13) unext.m[1] = load_piece(ublk, 2 * (t + 1) + 1, lane, 1);
        if (kG && s == 14) gnext.m[0] = load_piece(gblk, 2 * (t + 1), lane, 1);
        if (kG && s == 15) gnext.m[1] = load_piece(gblk, 2 * (t + 1) + 1, lane, 1);
      }
    };
    const bool fetch = !(LAST_STAGE && t == 7);
    f32x16 acc;
    if (!fetch) acc = tile_mma_h2<16>(st.cur_buf(), in, lane, [&](int s) { ep.a(s); }, [&](int s) { ep.b(t - 1, s); });
    else if (t == 0) acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, NoEpi(), NoEpi(), side);
    else acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, [&](int s) { ep.a(s); },
                                            [&](int s) { ep.template b<true>(t - 1, s); ep.store_slot(t - 1, s); }, side);
    ep.prev = acc; ep.h = hcur;
    if (A2) ep.u = ucur;
    if (kG) ep.g = gcur;
#if SVS_ABL & 65536
    asm volatile("" : "+v"(ep.prev[15]));
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
#endif
    if (fetch) {
      if (t == 0) st.advance_keep<kLoads>();
      else if (t < 7) st.advance_keep<kLoads + kStores>();
      else st.advance_keep<kStores>();
    }
#if SVS_ABL & 65536
    const uint64_t t2 = __builtin_amdgcn_s_memtime();
    ep.cyc[t] += (unsigned)(t1 - t0); ep.cyc[8 + t] += (unsigned)(t2 - t1);
    t0 = t2;
#endif
  }
  ep.all(7);
#if SVS_ABL & 65536
  asm volatile("" : "+v"(ep.v8[7]));
  ep.cyc[16] += (unsigned)(__builtin_amdgcn_s_memtime() - t0);
#endif
  ep.ps->floor_m = next_floor;
  ep.ps->next();
}

template <typename Net, bool A2, bool GP>
__global__ __launch_bounds__(kThreads, 1) void sdf_bwd_b_h2_kernel(SdfBwdBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int bx = a.reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
  const int wtile = bx * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.P ? p : a.P - 1;
  st.prefetch<kChunkF4>();
  float sbar = (a.d_sdf && p < a.P) ? a.d_sdf[pc] : 0.0f;
  if (a.clamp_mask && a.clamp_mask[pc]) sbar = 0.0f;
  if (half == 0 && a.sbar_out) a.sbar_out[p] = sbar;
  const size_t LS = block_stride();
  const float* hb = a.hbuf + (size_t)wtile * kBlockF;
  // A2: ghat_l = block l of gbuf; else only ghat_7, at w0 + tile * w0_stride (the background network keeps no other block)
  const float* gb = A2 ? a.gbuf + (size_t)wtile * kBlockF : a.w0 + (size_t)wtile * a.w0_stride - 7 * LS;
  const float* ub = A2 ? a.ubuf + (size_t)wtile * kBlockF : nullptr;
  float* ab = a.abuf + (size_t)wtile * kBlockF;
  const size_t T = (size_t)gridDim.x * kWaves;
  auto arec = [&](int l) { return record_ptr(a.abuf, 8, T, l, wtile); };
  auto urec = [&](int l) { return A2 ? record_ptr(a.ubuf, 9, T, l, wtile) : nullptr; };
  // >= max_r |a2_l| of this lane's point: (max_r |v_l| 100 (1 - s'), pass A) x (max_r |ghat_l|, sdf_full); with |sbar| (the
  // other additive input, first stage) the floor of the scale abar_l is split under
  const float asb = __builtin_fabsf(sbar);
  auto floor_of = [&](int l) {
    return A2 ? __builtin_fmaxf(asb, load_max(urec(l + 1), lane) * load_max(record_ptr(a.gbuf, 8, T, l, wtile), lane)) : asb;
  };
  const bool has_f = a.feat_bar && wtile < a.n_feat_tiles;
#if SVS_ABL & 65536
  unsigned cyc[17];
  for (int i = 0; i < 17; ++i) cyc[i] = 0;
  const uint64_t k0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  PointScale ps;
  Pieces2 pa, pb;
  {
    // fbar (a scaled block under its recorded scale) is the first operand as it stands
    const float fl = floor_of(7);
    if (has_f) {
      const float* fb = a.feat_bar + (size_t)wtile * kBlockF;
      // (feat_bar belongs to the launch over the ray samples only: its tile count is n_feat_tiles padded to workgroups)
      const float* frec = record_ptr(a.feat_bar, 1, (size_t)((a.n_feat_tiles + kWaves - 1) / kWaves * kWaves), 0, wtile);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        pa.h[k] = load_piece(fb, k, lane);
        pa.m[k] = GP ? load_piece(fb, k, lane, 1) : (f16x8)(_Float16)0.0f;
      }
      ps.start_stored(load_scale(frec, lane), load_max(frec, lane), fl);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) { pa.h[k] = (f16x8)(_Float16)0.0f; pa.m[k] = (f16x8)(_Float16)0.0f; }
      ps.start(0.0f, fl);
    }
    ps.gmax = 0.0f;      // fbar is not an operand of the SDF weight-gradient GEMMs (rgb_bwd publishes its maximum)
  }
  st.advance();
  {
    // hbar_8 fused with abar_7 (ghat_7 = W8[0,:] s'(a_7), stored unscaled)
    PassBEpi<true, true, Net, A2, GP> ep;
    ep.out = &pb; ep.ablk = ab + 7 * LS; ep.arec = arec(7); ep.ps = &ps; ep.sbar = sbar; ep.lane = lane; ep.half = half; ep.l4 = false;
#if SVS_ABL & 65536
    ep.cyc = cyc;
#endif
    pass_b_stage_h2<true, true, false, Net, A2, GP>(st, pa, ep, hb + 7 * LS, A2 ? ub + 8 * LS : nullptr, urec(8), gb + 7 * LS,
                                                    floor_of(6), lane);
  }
  // layers 7..2: in = abar_l (pb), out = abar_{l-1} (pa, copied back: one code body for all layers)
  for (int l = 7; l >= 2; --l) {
    PassBEpi<false, true, Net, A2, GP> ep;
    ep.out = &pa; ep.ablk = ab + (size_t)(l - 1) * LS; ep.arec = arec(l - 1); ep.ps = &ps; ep.sbar = 0.0f; ep.lane = lane; ep.half = half;
    ep.l4 = l == 4;
#if SVS_ABL & 65536
    ep.cyc = cyc;
#endif
    pass_b_stage_h2<false, true, false, Net, A2, GP>(st, pb, ep, hb + (size_t)(l - 1) * LS, A2 ? ub + (size_t)l * LS : nullptr, urec(l),
                                                     A2 ? gb + (size_t)(l - 1) * LS : nullptr, floor_of(l >= 2 ? l - 2 : 0), lane);
#pragma unroll
    for (int s = 0; s < 16; ++s) { pb.h[s] = pa.h[s]; pb.m[s] = pa.m[s]; }
  }
  {
    // layer 1: abar_1 (in pb) -> abar_0, stored only
    PassBEpi<false, false, Net, A2, GP> ep;
    ep.out = nullptr; ep.ablk = ab; ep.arec = arec(0); ep.ps = &ps; ep.sbar = 0.0f; ep.lane = lane; ep.half = half; ep.l4 = false;
#if SVS_ABL & 65536
    ep.cyc = cyc;
#endif
    pass_b_stage_h2<false, false, true, Net, A2, GP>(st, pb, ep, hb, A2 ? ub + LS : nullptr, urec(1), A2 ? gb : nullptr, asb, lane);
  }
  publish_max(a.absmax, ps.gmax);
#if SVS_ABL & 65536
  {
    const uint64_t k1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (threadIdx.x == 0 && a.sbar_out) {     // per tile index: MFMA part x8, wait part x8, last epilogue, total, real time
      float* o = a.sbar_out + (size_t)bx * kWaves * kTilePts;
      for (int i = 0; i < 17; ++i) o[i] = (float)cyc[i];
      o[17] = (float)(k1 - k0); o[18] = (float)(r1 - r0);
    }
  }
#endif
}

// d loss / d W_8[0,:] = sum_p (sbar_p h_8[:,p] + u_8[:,p]);  d loss / d b_8[0] = sum_p sbar_p  (svs_mlp_bwd.hip) on the fp16x2
// block forms: h_8 = pair block 7 of hbuf, u_8 = scaled block 8 of ubuf with its per-point scale (GP: both pieces).  Wave w of a workgroup
// owns k-steps 4w..4w+3 (64 of the 256 rows), grid-strides over the tiles with the 12 loads of a tile in flight; the sum
// over the 32 points of a lane half goes through LDS once per workgroup, then float atomics into out[257].
template <bool GP>
__global__ __launch_bounds__(256) void lin8_row0_h2_kernel(const float* __restrict__ hbuf, const float* __restrict__ ubuf,
                                                           const float* __restrict__ sbar, int n_tiles, int n_tiles_pad, int P,
                                                           float* __restrict__ out, det::Ticket ticket) {
  __shared__ float red[4][32][65];
  const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
  float bsum = 0.0f;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int p = t * 32 + (lane & 31);
    const float sb = p < P ? sbar[p] : 0.0f;
    const float* h = hbuf + ((size_t)7 * n_tiles_pad + t) * kBlockF;
    f16x8 hh[4], hm[4], uh[4], um[4];
    float us = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hh[k] = load_piece(h, 4 * quarter + k, lane, 0); hm[k] = load_piece(h, 4 * quarter + k, lane, 1); }
    if (ubuf) {
      const float* u = ubuf + ((size_t)8 * n_tiles_pad + t) * kBlockF;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uh[k] = load_piece(u, 4 * quarter + k, lane);
        um[k] = GP ? load_piece(u, 4 * quarter + k, lane, 1) : (f16x8)(_Float16)0.0f;
      }
      us = p < P ? PointScale::inv_pow2(load_scale(record_ptr(ubuf, 9, (size_t)n_tiles_pad, 8, t), lane)) : 0.0f;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) { uh[k] = (f16x8)(_Float16)0.0f; um[k] = (f16x8)(_Float16)0.0f; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[8 * k + j] += sb * ((float)hh[k][j] + (float)hm[k][j]) + us * ((float)uh[k][j] + (float)um[k][j]);
    if (quarter == 0 && lane < 32) bsum += sb;
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) red[quarter][i][lane] = acc[i];
  __syncthreads();
  det::wait_turn(ticket, blockIdx.x);
  // thread -> (quarter, element i = 8 k + j, half): sum its 32 points; row = 16 (4 q + k) + 8 (j >> 2) + 4 half + (j & 3)
  {
    const int q = threadIdx.x >> 6, i = (threadIdx.x >> 1) & 31, hf = threadIdx.x & 1;
    float v = 0.0f;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) v += red[q][i][32 * hf + c];
    const int k = i >> 3, j = i & 7;
    atomicAdd(&out[16 * (4 * q + k) + 8 * (j >> 2) + 4 * hf + (j & 3)], v);
  }
  if (quarter == 0) {
    for (int d = 16; d >= 1; d >>= 1) bsum += __shfl_xor(bsum, d);
    if (lane == 0) atomicAdd(&out[256], bsum);
  }
  det::pass_turn(ticket, blockIdx.x);
}

int launch_lin8_row0_h2(const float* hbuf, const float* ubuf, const float* sbar, int n_points, int n_tiles_pad, float* out257,
                        bool gp, hipStream_t s) {
  const int n_tiles = (n_points + 31) / 32;
  // one workgroup per CU: every workgroup ends with 257 float atomics onto the same addresses, and with a workgroup per tile
  // (800 at 256 rays) their contention was most of the launch (26 -> 17 us at 256 rays, 44 -> 37 at 1024; 128: 17 / 45)
  const int grid = n_tiles < 256 ? n_tiles : 256;
  const det::Ticket ticket{det::take_slots(1), 0u, (unsigned)grid};
  if (gp) lin8_row0_h2_kernel<true><<<grid, 256, 0, s>>>(hbuf, ubuf, sbar, n_tiles, n_tiles_pad, n_points, out257, ticket);
  else lin8_row0_h2_kernel<false><<<grid, 256, 0, s>>>(hbuf, ubuf, sbar, n_tiles, n_tiles_pad, n_points, out257, ticket);
  return check_launch("svs_lin8_row0_grad");
}

int launch_rgb_bwd_h2(const RgbBwdArgs& a, bool gp, hipStream_t s) {
  static int once = set_lds(rgb_bwd_h2_kernel<true>, kLdsBytes, "svs_rgb_bwd") | set_lds(rgb_bwd_h2_kernel<false>, kLdsBytes, "svs_rgb_bwd");
  if (once) return once;
  const int grid = (a.P + kWgPts - 1) / kWgPts;
  if (gp) rgb_bwd_h2_kernel<true><<<grid, kThreads, kLdsBytes, s>>>(a);
  else rgb_bwd_h2_kernel<false><<<grid, kThreads, kLdsBytes, s>>>(a);
  return check_launch("svs_rgb_bwd");
}
int launch_sdf_bwd_a_h2(const SdfBwdAArgs& a, bool gp, hipStream_t s) {
  constexpr int lds = kLdsBytes + kSpliceF * kThreads * (int)sizeof(float);
  static int once = set_lds(sdf_bwd_a_h2_kernel<true>, lds, "svs_sdf_bwd_a") | set_lds(sdf_bwd_a_h2_kernel<false>, lds, "svs_sdf_bwd_a");
  if (once) return once;
  const int grid = (a.src.P + kWgPts - 1) / kWgPts;
  if (gp) sdf_bwd_a_h2_kernel<true><<<grid, kThreads, lds, s>>>(a);
  else sdf_bwd_a_h2_kernel<false><<<grid, kThreads, lds, s>>>(a);
  return check_launch("svs_sdf_bwd_a");
}
int launch_sdf_bwd_b_h2(const SdfBwdBArgs& a_in, bool gp, hipStream_t s) {
  static int once = set_lds(sdf_bwd_b_h2_kernel<NetFg, true, true>, kLdsBytes, "svs_sdf_bwd_b") |
                    set_lds(sdf_bwd_b_h2_kernel<NetFg, true, false>, kLdsBytes, "svs_sdf_bwd_b");
  if (once) return once;
  static const int reverse_env = [] { const char* e = getenv("SVS_BWD_B_REVERSE"); return e ? atoi(e) : 0; }();
  SdfBwdBArgs a = a_in;
  a.reverse = reverse_env;
  const int grid = (a.P + kWgPts - 1) / kWgPts;
  if (gp) sdf_bwd_b_h2_kernel<NetFg, true, true><<<grid, kThreads, kLdsBytes, s>>>(a);
  else sdf_bwd_b_h2_kernel<NetFg, true, false><<<grid, kThreads, kLdsBytes, s>>>(a);
  return check_launch("svs_sdf_bwd_b");
}
int launch_bg_bwd_b_h2(const SdfBwdBArgs& a, bool gp, hipStream_t s) {
  static int once = set_lds(sdf_bwd_b_h2_kernel<NetBg, false, true>, kLdsBytes, "svs_bg_sdf_bwd") |
                    set_lds(sdf_bwd_b_h2_kernel<NetBg, false, false>, kLdsBytes, "svs_bg_sdf_bwd");
  if (once) return once;
  const int grid = (a.P + kWgPts - 1) / kWgPts;
  if (gp) sdf_bwd_b_h2_kernel<NetBg, false, true><<<grid, kThreads, kLdsBytes, s>>>(a);
  else sdf_bwd_b_h2_kernel<NetBg, false, false><<<grid, kThreads, kLdsBytes, s>>>(a);
  return check_launch("svs_bg_sdf_bwd");
}

}  // namespace mlp
}  // namespace svs
