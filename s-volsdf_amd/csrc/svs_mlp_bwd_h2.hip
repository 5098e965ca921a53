// fp16x2 variants of the training-backward sweeps of svs_mlp_bwd.hip (same algebra, same buffers): every layer
// product runs on v_mfma_f32_32x32x16_f16 with two-piece fp16 operands (svs_mlp_h2_dev.h), the previous tile's
// epilogue interleaved with the current tile's MFMAs.
//
// Gradients span many orders of magnitude and are typically far below fp16's range, so every point (= MFMA column
// = lane) carries its own power-of-two scale: an operand is split as true * s with s chosen so that the point's
// largest element is ~2^4, and the accumulator is multiplied by 1/s in the epilogue (both exact).  The scale used
// to split the operand a layer produces is derived from the maximum of the operand the layer consumed (the new
// maximum is only known once all eight tiles are done), which leaves 2^11 of headroom for growth across one layer;
// additive external inputs (a2 in pass B) enter through a per-point floor of the maximum.  All buffers written to
// memory hold true (unscaled) float32 values.  The kernels also publish the global maxima of the gradient-like
// operands of the weight-gradient GEMMs (svs_wgrad.hip, fp16x2 path) with one atomic max per wave.
#include "svs_mlp_h2_dev.h"
#include "svs_mlp_host.h"
#include "svs_mlp_bwd_args.h"
#include "svs_mlp_h2_trunk.h"
#include "svs_mlp_bwd_h2_dev.h"

namespace svs {
namespace mlp {

// ==============================================================================================================
// radiance MLP backward
// ==============================================================================================================
// zbar_{l-1} = rbar_l * [r_l > 0] for one tile: store (true units), track, split
struct RgbBwdEpi {
  f32x16 prev, r;
  float v8[8];
  LateStore ls;
  Pieces2* out;
  float* zblk;
  PointScale* ps;
  int lane;
  __device__ __forceinline__ void b(int tp, int rr) {
    float v = prev[rr] * ps->inv_in;
    v = r[rr] > 0.0f ? v : 0.0f;
    pin(v);
    ps->track(v);
    ls.put(rr, v);
    v8[rr & 7] = v * ps->s_out;
    if ((rr & 7) == 7) {
      split8(v8, out->h[2 * tp + (rr >> 3)], out->m[2 * tp + (rr >> 3)]);
      pin(out->h[2 * tp + (rr >> 3)], out->m[2 * tp + (rr >> 3)]);
    }
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) b(tp, rr);
    ls.all(zblk, tp, lane);
  }
};

// rbar_l = W_l^T zbar_l fused with zbar_{l-1} (l = 3..1)
__device__ __forceinline__ void rgb_bwd_layer_h2(Stream& st, const Pieces2& in, Pieces2& out, const float* rblk, float* zblk,
                                                 PointScale& ps, int lane) {
  RgbBwdEpi ep;
  ep.out = &out; ep.zblk = zblk; ep.ps = &ps; ep.lane = lane;
  // Per tile: the next chunk's 9 LDS-DMA pieces behind k-steps 0..8 (Stream::prefetch_step), then -- younger than every
  // piece, so that the barrier leaves them in flight (LateStore) -- the 4 zbuf stores of tile t-1's epilogue (k-steps
  // 9, 11, 13, 15) and the 4 loads of r tile t+1 (k-steps 10, 12, 14, 15), which the epilogue of tile t+1 consumes
  // during tile t+2.
  f32x16 rnext = load_tile(rblk, 0, lane);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const f32x16 rcur = rnext;
    auto rload = [&](int s) {
      const int q = s == 10 ? 0 : s == 12 ? 1 : s == 14 ? 2 : s == 15 ? 3 : -1;
      if (t < 7 && q >= 0) load_tile_quarter(rblk, t + 1, lane, q, rnext);
    };
    f32x16 acc;
    if (t == 0) acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, NoEpi(), NoEpi(), rload);
    else acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, NoEpi(), [&](int s) { ep.b(t - 1, s); },
                                            [&](int s) { ep.ls.step(s, ep.zblk, t - 1, lane); rload(s); });
    ep.prev = acc; ep.r = rcur;
    if (t == 0) st.advance_keep<4>();
    else if (t < 7) st.advance_keep<8>();
    else st.advance_keep<4>();
  }
  ep.all(7);
  ps.next();
}

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
  {  // zbar_4: rows 0..2 live in registers 0..2 of half 0, tile 0
    f32x16 z4 = (f32x16)(0.0f);
    z4[0] = dz[0]; z4[1] = dz[1]; z4[2] = dz[2];
    store_tile(zb + 4 * LS, 0, lane, z4);
  }
  st.advance();
  Pieces2 pa, pb;
  // rbar_4 = W_4^T zbar_4 (K = 3: float32 MFMA from the short W4T chunk), masked by r_4 > 0 -> zbar_3
  st.prefetch<kChunkF4>();
  {
    const f32x4* c = st.cur_buf();
    const float* r4 = rb + 3 * LS;
    float* z3 = zb + 3 * LS;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x16 r = load_tile(r4, t, lane);
      const f32x4 w = c[t * 64 + lane];
      f32x16 acc = (f32x16)(0.0f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], dz[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], dz[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], dz[2], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[i] = r[i] > 0.0f ? acc[i] : 0.0f; ps.track(acc[i]); }
      store_tile(z3, t, lane, acc);
      split_tile_scaled(acc, t, pa, ps.s_out);
    }
  }
  ps.next();
  st.advance();
  // layers 3..1
  rgb_bwd_layer_h2(st, pa, pb, rb + 2 * LS, zb + 2 * LS, ps, lane);
  rgb_bwd_layer_h2(st, pb, pa, rb + 1 * LS, zb + 1 * LS, ps, lane);
  rgb_bwd_layer_h2(st, pa, pb, rb, zb, ps, lane);
  // layer 0: the input gradients from zbar_0 (feature rows: tiles 0..7, extras: tile 8)
  float* fb = a.feat_bar + (size_t)wtile * kBlockF;
  float fmax = 0.0f;
  {
    f32x16 prev;
    f32x4 q4;
    auto slice = [&](int tp, int r) {
      const float v = prev[r] * ps.inv_in;
      fmax = __builtin_fmaxf(fmax, __builtin_fabsf(v));
      q4[r & 3] = v;
      if ((r & 3) == 3) SVS_STREAM_STORE(q4, reinterpret_cast<f32x4*>(fb) + (4 * tp + (r >> 2)) * 64 + lane);
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
  publish_max(a.absmax + 1, ps.gmax);
  publish_max(a.absmax + 2, fmax);
}

// ==============================================================================================================
// SDF MLP backward, pass A: u_0 = J_PE nbar;  v_l = W_l u_l;  u_{l+1} = v_l s'(a_l);  a2_l = v_l g(h_{l+1}) s''(a_l)
// ==============================================================================================================
constexpr int kSpliceF = 20;     // floats per lane kept in LDS for the skip splice of pass A (tile 7 + 4 registers of tile 6)

template <bool SPLIT>
struct PassAEpi {
  f32x16 prev, h, g;
  float v, s1;
  float v8[8];
  f32x4 qu, qa;
  Pieces2* out;       // u_{l+1} as the next operand (SPLIT)
  float* ublk;        // u_{l+1} block
  float* a2blk;
  PointScale* ps;
  const float* splice;  // LDS: this lane's u_0 splice values, [kSpliceF][kThreads]
  float a2m;          // running max |a2|
  int lane, half;
  bool l3;            // layer 3: the skip connection carries u_0 into rows >= 217 of u_4, a2 is zero there

  __device__ __forceinline__ void a(int r) {
    v = prev[r] * ps->inv_in;
    s1 = dsoftplus_from_h(h[r]);
    pin(v); pin(s1);
  }
  __device__ __forceinline__ void b(int tp, int r) {
    float u = v * s1;
    float a2 = v * g[r] * (100.0f * (1.0f - s1));     // g = ghat_l = g(h_{l+1}) s'(a_l); s'' = 100 s' (1 - s')
    if (tp == 6 && r >= 12 && l3) {
      // local rows 25..31 of tile 6 (registers 13..15 of half 0, 12..15 of half 1) carry u_0[32..38]
      const bool sp = half == 1 || r >= 13;
      const float us = splice[(16 + (r - 12)) * kThreads];
      u = sp ? us : u;
      a2 = sp ? 0.0f : a2;
    }
    pin(u); pin(a2);
    emit(tp, r, u, a2);
  }
  __device__ __forceinline__ void emit(int tp, int r, float u, float a2) {
    a2m = __builtin_fmaxf(a2m, __builtin_fabsf(a2));
    ps->track(u);
    qa[r & 3] = a2;
    if ((r & 3) == 3) SVS_STREAM_STORE(qa, reinterpret_cast<f32x4*>(a2blk) + (4 * tp + (r >> 2)) * 64 + lane);
    qu[r & 3] = u;
    if ((r & 3) == 3) SVS_STREAM_STORE(qu, reinterpret_cast<f32x4*>(ublk) + (4 * tp + (r >> 2)) * 64 + lane);
    if (SPLIT) {
      v8[r & 7] = u * ps->s_out;
      if ((r & 7) == 7) {
        split8(v8, out->h[2 * tp + (r >> 3)], out->m[2 * tp + (r >> 3)]);
        pin(out->h[2 * tp + (r >> 3)], out->m[2 * tp + (r >> 3)]);
      }
    }
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { a(r); b(tp, r); }
  }
  __device__ __forceinline__ void splice_tile7() {   // layer 3, tile 7: u = u_0[0..31], a2 = 0 (no MFMA)
#pragma unroll
    for (int r = 0; r < 16; ++r) emit(7, r, splice[r * kThreads], 0.0f);
  }
};

// layer l >= 1 of pass A; hblk/gblk: blocks l of hbuf / gbuf.  LAST: no chunk follows the layer's last one.
template <bool SPLIT, bool LAST>
__device__ __forceinline__ void pass_a_layer_h2(Stream& st, const Pieces2& in, PassAEpi<SPLIT>& ep, const float* hblk,
                                                const float* gblk, int lane) {
  // Per tile: the side tiles h, ghat of tile t are requested in front of it (their epilogue runs during tile t+1; a third
  // register set for requesting them behind the pieces, as the other sweeps do, does not fit: 110 spills); the next
  // chunk's LDS-DMA pieces go behind k-steps 0..8 (Stream::prefetch_step); the a2 / u stores of tile t-1's epilogue are
  // issued in k-steps 3, 7, 11, 15 (two each): the last four are younger than every piece and stay in flight across the
  // tile's barrier.
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t == 7 && ep.l3) break;
    const f32x16 hload = load_tile(hblk, t, lane), gload = load_tile(gblk, t, lane);
    const bool fetch = !(LAST && t == 7);
    f32x16 acc;
    if (!fetch) acc = tile_mma_h2<16>(st.cur_buf(), in, lane, [&](int s) { ep.a(s); }, [&](int s) { ep.b(t - 1, s); });
    else if (t == 0) acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, NoEpi(), NoEpi());
    else acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, [&](int s) { ep.a(s); }, [&](int s) { ep.b(t - 1, s); });
    ep.prev = acc; ep.h = hload; ep.g = gload;
    if (fetch) {
      if (t == 0) st.advance();
      else st.advance_keep<4>();
    }
  }
  if (ep.l3) { ep.all(6); ep.splice_tile7(); }
  else ep.all(7);
  ep.ps->next();
}

__global__ __launch_bounds__(kThreads, 1) void sdf_bwd_a_h2_kernel(SdfBwdAArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  float* splice = reinterpret_cast<float*>(smem + kLdsBytes) + threadIdx.x;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.src.P ? p : a.src.P - 1;

  st.prefetch<kChunk0F4>();
  PointScale ps;
  Pieces2 pa, pb;
  const size_t LS = block_stride();
  const float* hb = a.hbuf + (size_t)wtile * kBlockF;
  const float* gb = a.gbuf + (size_t)wtile * kBlockF;
  float* ub = a.ubuf + (size_t)wtile * kBlockF;
  float* a2 = a.a2buf + (size_t)wtile * kBlockF;
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
    // u_0 / PE in PE order as 2-tile accumulator-layout blocks (rows q = 32*tile + rho(r) + 4*half)
    auto store_pe_block = [&](float* block, const float* vec40) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q0 = 32 * t + rho(r), q1 = q0 + 4;
          const float a0 = q0 < kPeDim ? vec40[q0 < 40 ? q0 : 39] : 0.0f;
          const float a1 = q1 < kPeDim ? vec40[q1 < 40 ? q1 : 39] : 0.0f;
          v[r] = half ? a1 : a0;
        }
        store_tile(block, t, lane, v);
      }
    };
    store_pe_block(ub, u0.v);
    store_pe_block(a.pebuf + (size_t)wtile * kBlockF, pe.v);
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
  }
  st.advance();

  float a2m = 0.0f;
  {
    // ---- layer 0 (three k-steps per tile: the epilogue of tile t-1 follows tile t's MFMAs)
    PassAEpi<true> ep;
    ep.out = &pb; ep.ublk = ub + LS; ep.a2blk = a2; ep.ps = &ps; ep.splice = splice; ep.a2m = 0.0f;
    ep.lane = lane; ep.half = half; ep.l3 = false;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x16 hload = load_tile(hb, t, lane), gload = load_tile(gb, t, lane);
      if (t < 7) st.prefetch<kChunk0F4>(); else st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma_h2<3>(st.cur_buf(), pa, lane);
      if (t > 0) ep.all(t - 1);
      ep.prev = acc; ep.h = hload; ep.g = gload;
      st.advance();
    }
    ep.all(7);
    ps.next();
    a2m = ep.a2m;
  }
  // ---- layers 1..6: pb -> pa, copied back (one code body for all layers)
  for (int l = 1; l < 7; ++l) {
    PassAEpi<true> ep;
    ep.out = &pa; ep.ublk = ub + (size_t)(l + 1) * LS; ep.a2blk = a2 + (size_t)l * LS; ep.ps = &ps;
    ep.splice = splice; ep.a2m = a2m; ep.lane = lane; ep.half = half; ep.l3 = l == 3;
    pass_a_layer_h2<true, false>(st, pb, ep, hb + (size_t)l * LS, gb + (size_t)l * LS, lane);
    a2m = ep.a2m;
#pragma unroll
    for (int s = 0; s < 16; ++s) { pb.h[s] = pa.h[s]; pb.m[s] = pa.m[s]; }
  }
  // ---- layer 7: u_8 (only needed for the row-0 gradient of lin8) is stored, not split
  {
    PassAEpi<false> ep;
    ep.out = nullptr; ep.ublk = ub + (size_t)8 * LS; ep.a2blk = a2 + (size_t)7 * LS; ep.ps = &ps;
    ep.splice = splice; ep.a2m = a2m; ep.lane = lane; ep.half = half; ep.l3 = false;
    pass_a_layer_h2<false, true>(st, pb, ep, hb + (size_t)7 * LS, gb + (size_t)7 * LS, lane);
    a2m = ep.a2m;
  }
  a2m = __builtin_fmaxf(a2m, __shfl_xor(a2m, 32));
  if (half == 0) a.a2max[p] = a2m;
  publish_max(a.absmax, ps.gmax);
}

// ==============================================================================================================
// SDF MLP backward, pass B: hbar_8 = sbar W_8[0,:] + W_8[1:,:]^T fbar;  abar_l = hbar_{l+1} s'(a_l) + a2_l;
//                           hbar_l = W_l^T abar_l
// ==============================================================================================================
template <bool FIRST, bool SPLIT, typename Net = NetFg, bool A2 = true>
struct PassBEpi {
  f32x16 prev, h, a2, w0;
  float v, s1;
  float v8[8];
  LateStore ls;
  Pieces2* out;
  float* ablk;
  PointScale* ps;
  float sbar;
  int lane, half;
  bool l4;            // producing abar_3: rows >= 217 of h_4 are the PE splice
  __device__ __forceinline__ void a(int r) {
    v = prev[r] * ps->inv_in;
    s1 = dsoftplus_from_h(h[r]);
    pin(v); pin(s1);
  }
  __device__ __forceinline__ void b(int tp, int r) {
    float o = v * s1;
    if (A2) o += a2[r];
    if (FIRST) o += sbar * w0[r];      // w0 = ghat_7 = W8[0,:] s'(a_7)
    if (l4 && tp > Net::kSpliceTile) o = 0.0f;
    if (l4 && tp == Net::kSpliceTile) {
      const bool z0 = rho(r) >= Net::kSpliceLocal, z1 = rho(r) + 4 >= Net::kSpliceLocal;
      if (z0 || z1) { if (half ? z1 : z0) o = 0.0f; }
    }
    pin(o);
    ps->track(o);
    ls.put(r, o);
    if (SPLIT) {
      v8[r & 7] = o * ps->s_out;
      if ((r & 7) == 7) {
        split8(v8, out->h[2 * tp + (r >> 3)], out->m[2 * tp + (r >> 3)]);
        pin(out->h[2 * tp + (r >> 3)], out->m[2 * tp + (r >> 3)]);
      }
    }
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { a(r); b(tp, r); }
    ls.all(ablk, tp, lane);
  }
};

// one stage of pass B: in -> (W^T in) fused with abar of block `blk` (h, a2 from blocks blk; FIRST: + sbar W8[0,:])
template <bool FIRST, bool SPLIT, bool LAST_STAGE, typename Net = NetFg, bool A2 = true>
__device__ __forceinline__ void pass_b_stage_h2(Stream& st, const Pieces2& in, PassBEpi<FIRST, SPLIT, Net, A2>& ep, const float* hblk,
                                                const float* a2blk, const float* w0blk, int lane) {
  // Per tile: the next chunk's LDS-DMA pieces behind k-steps 0..8 (Stream::prefetch_step), then -- younger than every
  // piece, left in flight across the tile's barrier (LateStore) -- the 4 abuf stores of tile t-1's epilogue (k-steps 9,
  // 11, 13, 15) and the loads of the side tiles h, a2 (FIRST: and ghat_7) of tile t+1, which the epilogue of tile t+1
  // consumes during tile t+2.
  f32x16 hnext = load_tile(hblk, 0, lane), anext, wnext;
  if (A2) anext = load_tile(a2blk, 0, lane);
  if (FIRST) wnext = load_tile(w0blk, 0, lane);
  constexpr int kLoads = 4 + (A2 ? 4 : 0) + (FIRST ? 4 : 0);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const f32x16 hcur = hnext, acur = anext, wcur = wnext;
    auto side = [&](int s) {
      if (t == 7) return;
      const int qh = s == 9 ? 0 : s == 11 ? 1 : s == 13 ? 2 : s == 15 ? 3 : -1;   // beside the stores
      const int qa = s == 10 ? 0 : s == 12 ? 1 : s == 14 ? 2 : s == 15 ? 3 : -1;
      if (qh >= 0) load_tile_quarter(hblk, t + 1, lane, qh, hnext);
      if (A2 && qa >= 0) load_tile_quarter(a2blk, t + 1, lane, qa, anext);
      if (FIRST && qa >= 0) load_tile_quarter(w0blk, t + 1, lane, qa, wnext);
    };
    const bool fetch = !(LAST_STAGE && t == 7);
    f32x16 acc;
    if (!fetch) acc = tile_mma_h2<16>(st.cur_buf(), in, lane, [&](int s) { ep.a(s); }, [&](int s) { ep.b(t - 1, s); }, 0,
                                       [&](int s) { ep.ls.step(s, ep.ablk, t - 1, lane); });
    else if (t == 0) acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, NoEpi(), NoEpi(), side);
    else acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, [&](int s) { ep.a(s); }, [&](int s) { ep.b(t - 1, s); },
                                            [&](int s) { ep.ls.step(s, ep.ablk, t - 1, lane); side(s); });
    ep.prev = acc; ep.h = hcur;
    if (A2) ep.a2 = acur;
    if (FIRST) ep.w0 = wcur;
    if (fetch) {
      if (t == 0) st.advance_keep<kLoads>();
      else if (t < 7) st.advance_keep<kLoads + 4>();
      else st.advance_keep<4>();
    }
  }
  ep.all(7);
  ep.ps->next();
}

template <typename Net, bool A2>
__global__ __launch_bounds__(kThreads, 1) void sdf_bwd_b_h2_kernel(SdfBwdBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.P ? p : a.P - 1;
  st.prefetch<kChunkF4>();
  float sbar = (a.d_sdf && p < a.P) ? a.d_sdf[pc] : 0.0f;
  if (a.clamp_mask && a.clamp_mask[pc]) sbar = 0.0f;
  if (half == 0 && a.sbar_out) a.sbar_out[p] = sbar;
  const size_t LS = block_stride();
  const float* hb = a.hbuf + (size_t)wtile * kBlockF;
  const float* w0 = a.w0 + (size_t)wtile * a.w0_stride;
  const float* a2 = A2 ? a.a2buf + (size_t)wtile * kBlockF : nullptr;
  float* ab = a.abuf + (size_t)wtile * kBlockF;
  const bool has_f = a.feat_bar && wtile < a.n_feat_tiles;
  PointScale ps;
  Pieces2 pa, pb;
  {
    // fbar (true units) -> first operand; the scale floor covers the additive inputs a2 and sbar * W8[0,:]
    f32x16 f[8];
    float m0 = 0.0f;
    if (has_f) {
      load_tile_regs(a.feat_bar + (size_t)wtile * kBlockF, f, lane);
#pragma unroll
      for (int i = 0; i < 128; ++i) m0 = __builtin_fmaxf(m0, __builtin_fabsf(f[i / 16][i % 16]));
      m0 = __builtin_fmaxf(m0, __shfl_xor(m0, 32));
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) f[t] = (f32x16)(0.0f);
    }
    ps.start(m0, __builtin_fmaxf(A2 ? a.a2max[p] : 0.0f, __builtin_fabsf(sbar)));
    ps.gmax = 0.0f;      // fbar is not an operand of the SDF weight-gradient GEMMs (rgb_bwd publishes its maximum)
#pragma unroll
    for (int t = 0; t < 8; ++t) split_tile_scaled(f[t], t, pa, ps.s_in);
  }
  st.advance();
  {
    // hbar_8 fused with abar_7 (gbuf block 7 = ghat_7 = W8[0,:] s'(a_7) in accumulator layout)
    PassBEpi<true, true, Net, A2> ep;
    ep.out = &pb; ep.ablk = ab + 7 * LS; ep.ps = &ps; ep.sbar = sbar; ep.lane = lane; ep.half = half; ep.l4 = false;
    pass_b_stage_h2<true, true, false, Net, A2>(st, pa, ep, hb + 7 * LS, A2 ? a2 + 7 * LS : nullptr, w0, lane);
  }
  // layers 7..2: in = abar_l (pb), out = abar_{l-1} (pa, copied back: one code body for all layers)
  for (int l = 7; l >= 2; --l) {
    PassBEpi<false, true, Net, A2> ep;
    ep.out = &pa; ep.ablk = ab + (size_t)(l - 1) * LS; ep.ps = &ps; ep.sbar = 0.0f; ep.lane = lane; ep.half = half; ep.l4 = l == 4;
    pass_b_stage_h2<false, true, false, Net, A2>(st, pb, ep, hb + (size_t)(l - 1) * LS, A2 ? a2 + (size_t)(l - 1) * LS : nullptr, nullptr, lane);
#pragma unroll
    for (int s = 0; s < 16; ++s) { pb.h[s] = pa.h[s]; pb.m[s] = pa.m[s]; }
  }
  {
    // layer 1: abar_1 (in pb) -> abar_0, stored only
    PassBEpi<false, false, Net, A2> ep;
    ep.out = nullptr; ep.ablk = ab; ep.ps = &ps; ep.sbar = 0.0f; ep.lane = lane; ep.half = half; ep.l4 = false;
    pass_b_stage_h2<false, false, true, Net, A2>(st, pb, ep, hb, a2, nullptr, lane);
  }
  publish_max(a.absmax, ps.gmax);
}

int launch_rgb_bwd_h2(const RgbBwdArgs& a, hipStream_t s) {
  static int once = set_lds(rgb_bwd_h2_kernel, kLdsBytes, "svs_rgb_bwd");
  if (once) return once;
  rgb_bwd_h2_kernel<<<(a.P + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, s>>>(a);
  return check_launch("svs_rgb_bwd");
}
int launch_sdf_bwd_a_h2(const SdfBwdAArgs& a, hipStream_t s) {
  constexpr int lds = kLdsBytes + kSpliceF * kThreads * (int)sizeof(float);
  static int once = set_lds(sdf_bwd_a_h2_kernel, lds, "svs_sdf_bwd_a");
  if (once) return once;
  sdf_bwd_a_h2_kernel<<<(a.src.P + kWgPts - 1) / kWgPts, kThreads, lds, s>>>(a);
  return check_launch("svs_sdf_bwd_a");
}
int launch_sdf_bwd_b_h2(const SdfBwdBArgs& a, hipStream_t s) {
  static int once = set_lds(sdf_bwd_b_h2_kernel<NetFg, true>, kLdsBytes, "svs_sdf_bwd_b");
  if (once) return once;
  sdf_bwd_b_h2_kernel<NetFg, true><<<(a.P + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, s>>>(a);
  return check_launch("svs_sdf_bwd_b");
}
int launch_bg_bwd_b_h2(const SdfBwdBArgs& a, hipStream_t s) {
  static int once = set_lds(sdf_bwd_b_h2_kernel<NetBg, false>, kLdsBytes, "svs_bg_sdf_bwd");
  if (once) return once;
  sdf_bwd_b_h2_kernel<NetBg, false><<<(a.P + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, s>>>(a);
  return check_launch("svs_bg_sdf_bwd");
}

}  // namespace mlp
}  // namespace svs
