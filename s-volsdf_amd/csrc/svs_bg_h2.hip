// The inverted-sphere background networks of VolSDFNetworkBG (volsdf/model/network_bg.py:31-35, 85-103) on fp16x2
// MFMAs (the float32-MFMA forms behind the same entry points: svs_bg_f32.hip): bg_implicit_network (4-D points, PE-10 = 84 inputs, 8 x 256 softplus, skip at 4, no weight-norm; output =
// [density logit, 256 features]) and bg_rendering_network (mode 'nerf': cat[PE4(view)(27), feature(256)] -> 128 ReLU
// -> 3 sigmoid).  Same machinery as svs_mlp_h2.hip (shared trunk: svs_mlp_h2_trunk.h with the NetBg geometry).
#include "svs_bg_args.h"
#include "svs_mlp_host.h"
#include "svs_mlp_bwd_h2_dev.h"

namespace svs {
namespace mlp {

template <bool TRAIN, bool GP>      // GP: the blocks kept for the backward in the both-pieces format (svs_blocks_h2.h)
__global__ __launch_bounds__(kThreads, 1) void bg_sdf_h2_kernel(BgSdfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream;
  st.buf = reinterpret_cast<f32x4*>(smem);
  st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.P ? p : a.P - 1;

  st.prefetch<kBgChunk0F4>();
  PosEncBg pe;
  {
    const f32x4 x = reinterpret_cast<const f32x4*>(a.pts)[pc];
    pe.compute(x[0], x[1], x[2], x[3]);
  }
  float* hb = TRAIN ? a.hbuf + (size_t)wtile * kBlockF : nullptr;
  Pieces2 x, xn;
  if (TRAIN) {
    // h_0 = PE (84 rows, 6 k-steps) in PE order as block fragments, the rest zero: the B operand of dW_0
    float* pb = a.pebuf + (size_t)wtile * kBlockF;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      f16x8 fh = (f16x8)(_Float16)0.0f, fm = (f16x8)(_Float16)0.0f;
      if (k < NetBg::kSteps0) block_fragment<kBgPeDim>(pe.v, k, half, 1.0f, fh, fm);
      store_grad<GP>(pb, k, lane, fh, fm);
    }
  }
  f32x16 y8[8];
  forward_trunk_h2<TRAIN, NetBg>(st, x, xn, y8, pe, lane, half, hb);
  // ---- head: current chunk = VEC (W8 row 0 in C-layout order as float32, b8[0])
  st.prefetch<kChunkF4>();                       // FEAT tile 0
  const float out0 = sdf_head(st.cur_buf(), y8, lane);
  if (TRAIN) {
    const f32x4* w_ptr = st.cur_buf() + kHdrF4 + lane;
    float* g7 = a.ghat7 + (size_t)wtile * kBlockF;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x16 g;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = w_ptr[(4 * t + q) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[4 * q + j] = w[j] * dsoftplus_from_h(y8[t][4 * q + j]);
      }
      float v8[8];
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = g[8 * sh + j];
        store_grad8<GP>(g7, 2 * t + sh, lane, v8);       // ghat_7, unscaled
      }
    }
  }
  if (!TRAIN) {     // (with TRAIN the trunk already split h_8 into x: its pieces are what hbuf stores)
#pragma unroll
    for (int t = 0; t < 8; ++t) split_tile(y8[t], t, x);
  }
  st.advance();
  if (half == 0 && p < a.P) a.out0[p] = out0;
  // ---- feature vector = rows 1..256 of lin8 (no activation); tile t-1 is stored while tile t's MFMAs run
  float* ft = a.feat_tiles + (size_t)wtile * kBlockF;     // a PAIR block (svs_blocks_h2.h)
  f32x16 prev;
  float v8[8];
  auto store_slice = [&](int tp, int r) {
    v8[r & 7] = prev[r];
    if ((r & 7) == 7) {
      f16x8 fh, fm;
      split8(v8, fh, fm);
      store_piece(ft, 2 * tp + (r >> 3), lane, fh, 0);
      store_piece(ft, 2 * tp + (r >> 3), lane, fm, 1);
    }
  };
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < 7) st.prefetch<kChunkF4>();
    f32x16 acc;
    if (t == 0) acc = tile_mma_h2<16>(st.cur_buf(), x, lane);
    else acc = tile_mma_h2<16>(st.cur_buf(), x, lane, NoEpi(), [&](int s) { store_slice(t - 1, s); });
    prev = acc;
    if (t < 7) st.advance();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) store_slice(7, r);
}

// ------------------------------------------------------------------------------------------------------------
// bg_rendering_network, mode 'nerf' (network.py:170-190 with bmvs.yaml:70-77)
// ------------------------------------------------------------------------------------------------------------
template <bool GP>      // GP: r_1 kept with both pieces
__global__ __launch_bounds__(kThreads, 1) void bg_rgb_h2_kernel(BgRgbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  BgRgbStream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.P ? p : a.P - 1;

  st.prefetch<kBgRgbChunk0F4>();
  const float* vd = a.view + 3 * (size_t)(a.view_S > 0 ? pc / a.view_S : pc);
  // PE-4 of the view direction: [d(3), sin(2^f d)(3), cos(2^f d)(3), f = 0..3] = 27 entries, padded to 32
  float ex[32];
  {
    const float d[3] = {vd[0], vd[1], vd[2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) ex[c] = d[c];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float s, co;
        sincosf(d[c] * (float)(1 << f), &s, &co);
        ex[3 + 6 * f + c] = s;
        ex[6 + 6 * f + c] = co;
      }
#pragma unroll
    for (int q = 27; q < 32; ++q) ex[q] = 0.0f;
  }
  Pieces2 x, xn;
  float eb[16];   // fragments of k-steps 16, 17: rows 16 s' + rho(j) (+4) of the 32 extra rows
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int j = 0; j < 8; ++j) eb[8 * s + j] = half ? ex[16 * s + rho(j) + 4] : ex[16 * s + rho(j)];
    split8(eb + 8 * s, x.h[16 + s], x.m[16 + s]);
  }
  {
    const float* ft = a.feat_tiles + (size_t)wtile * kBlockF;    // a pair block: the operand as it is
#pragma unroll
    for (int k = 0; k < 16; ++k) { x.h[k] = load_piece(ft, k, lane, 0); x.m[k] = load_piece(ft, k, lane, 1); }
  }
  float* rb = a.rbuf ? a.rbuf + (size_t)wtile * kBgRbufF : nullptr;
  if (rb) {
    // the 32 extra rows as ONE accumulator-layout tile (register r = 8 s + j holds row 16 s + rho(j) (+4) = rho(r) (+4)):
    // the single-tile B operand of the weight-gradient GEMM's narrow job
    f32x4* d = reinterpret_cast<f32x4*>(rb + (size_t)kBlockF) + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
      v[0] = eb[4 * q]; v[1] = eb[4 * q + 1]; v[2] = eb[4 * q + 2]; v[3] = eb[4 * q + 3];
      d[q * 64] = v;
    }
  }
  st.advance();
  // ---- layer 0: 283 -> 128 (4 tiles), ReLU
  {
    f32x16 prev;
    float v8[8];
    auto slice = [&](int tp, int r) {
      float v = __builtin_fmaxf(prev[r], 0.0f);
      pin(v);
      v8[r & 7] = v;
      if ((r & 7) == 7) {
        const int k = 2 * tp + (r >> 3);
        split8(v8, xn.h[k], xn.m[k]);
        pin(xn.h[k], xn.m[k]);
        if (rb) store_grad<GP>(rb, k, lane, xn.h[k], xn.m[k]);    // r_1 (ReLU mask, weight gradient's B operand)
      }
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < 3) st.prefetch<kBgRgbChunk0F4>(); else st.prefetch<kChunkF4>();
      f32x16 acc;
      if (t == 0) acc = tile_mma_h2<18>(st.cur_buf(), x, lane);
      else acc = tile_mma_h2<18>(st.cur_buf(), x, lane, NoEpi(), [&](int s) { if (s < 16) slice(t - 1, s); });
      prev = acc;
      st.advance();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) slice(3, r);
    if (rb) {
      // k-steps 8..15 (rows 128..255) of the r_1 block stay zero (the weight-gradient GEMM reads whole 256-row blocks)
#pragma unroll
      for (int k = 8; k < 16; ++k) store_grad<GP>(rb, k, lane, (f16x8)(_Float16)0.0f, (f16x8)(_Float16)0.0f);
    }
  }
  // ---- layer 1: 128 -> 3 as one tile (rows 0..2 live in registers 0..2 of lanes 0..31), sigmoid
  const f32x16 acc = tile_mma_h2<8>(st.cur_buf(), xn, lane);
  if (half == 0 && p < a.P) {
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb[3 * p + c] = 1.0f / (1.0f + __expf(-acc[c]));
  }
}

// ------------------------------------------------------------------------------------------------------------
// bg_rendering_network backward: zbar_1 = d_rgb * sigmoid', zbar_0 = (W1^T zbar_1) * [r_1 > 0],
// fbar = W0[:, 27:]^T zbar_0 (the view directions get no gradient).  Per-point scaling as in svs_mlp_bwd_h2.hip.
// ------------------------------------------------------------------------------------------------------------
template <bool GP>
__global__ __launch_bounds__(kThreads, 1) void bg_rgb_bwd_h2_kernel(BgRgbBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const bool livep = p < a.P;
  const int pc = livep ? p : a.P - 1;
  const float* rb = a.rbuf + (size_t)wtile * kBgRbufF;
  float* zb = a.zbuf + (size_t)wtile * 2 * kBlockF;
  // records behind the slots, [block][tile][64] as in every buffer (the SLOTS of this small buffer are [tile][block])
  const size_t T = (size_t)gridDim.x * kWaves;
  float* zrec0 = record_ptr(a.zbuf, 2, T, 0, wtile);
  float* zrec1 = record_ptr(a.zbuf, 2, T, 1, wtile);

  st.prefetch<kW1TF4>();
  float dz[3];
  float m0 = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float o = a.rgb[3 * pc + c];
    dz[c] = (livep && half == 0) ? a.d_rgb[3 * pc + c] * o * (1.0f - o) : 0.0f;
    m0 = __builtin_fmaxf(m0, __builtin_fabsf(dz[c]));
  }
  m0 = __builtin_fmaxf(m0, __shfl_xor(m0, 32));
  PointScale ps;
  ps.start(m0, 0.0f);
  {  // zbar_1: rows 0..2 = elements 0..2 of k-step 0 of half 0 (a half block; the rest stays zero)
    float z8[8] = {dz[0] * ps.s_in, dz[1] * ps.s_in, dz[2] * ps.s_in, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    store_grad8<GP>(zb + (size_t)kBlockF, 0, lane, z8);
    store_record(zrec1, lane, ps.s_in, m0);
  }
  st.advance();
  Pieces2 pz;
  st.prefetch<kChunkF4>();
  {
    // rbar_1 = W_1^T zbar_1 (K = 3: float32 MFMA from the short W1T chunk), masked by r_1 > 0 -> zbar_0 (4 tiles)
    const f32x4* c = st.cur_buf();
    store_record(zrec0, lane, ps.s_out, 0.0f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      TilePieces r;
      load_tile_hi(rb, t, lane, r);
      const f32x4 w = c[t * 64 + lane];
      f32x16 acc = (f32x16)(0.0f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], dz[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], dz[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], dz[2], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[i] = hi_at(r, i) > 0.0f ? acc[i] : 0.0f; ps.track(acc[i]); }
      split_tile_scaled(acc, t, pz, ps.s_out);
      store_grad<GP>(zb, 2 * t, lane, pz.h[2 * t], pz.m[2 * t]);              // zbar_0: a scaled block under s_out
      store_grad<GP>(zb, 2 * t + 1, lane, pz.h[2 * t + 1], pz.m[2 * t + 1]);
    }
  }
  ps.next();
  st.advance();
  // fbar: 8 tiles of feature rows, K = 128 (8 k-steps)
  float* fb = a.feat_bar + (size_t)wtile * kBlockF;     // a half block under the predicted scale s_f (see rgb_bwd_h2_kernel)
  float fmax = 0.0f;
  const float s_f = ps.s_out;
  f32x16 prev;
  float v8[8];
  auto slice = [&](int tp, int r) {
    const float v = prev[r] * ps.inv_in;
    fmax = __builtin_fmaxf(fmax, __builtin_fabsf(v));
    v8[r & 7] = v * s_f;
    if ((r & 7) == 7) store_grad8<GP>(fb, 2 * tp + (r >> 3), lane, v8);
  };
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < 7) st.prefetch<kChunkF4>();
    const f32x16 acc = tile_mma_h2<8>(st.cur_buf(), pz, lane);
    if (t > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) slice(t - 1, r);
    }
    prev = acc;
    if (t < 7) st.advance();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) slice(7, r);
  fmax = __builtin_fmaxf(fmax, __shfl_xor(fmax, 32));
  store_record(record_ptr(a.feat_bar, 1, T, 0, wtile), lane, s_f, fmax);
  publish_max(a.absmax + 1, ps.gmax);
  publish_max(a.absmax + 2, fmax);
}

}  // namespace mlp
}  // namespace svs

using namespace svs;
using namespace svs::mlp;

extern "C" {

size_t svs_bg_rbuf_bytes(int n_points) { return (size_t)wave_tiles(n_points) * kBgRbufF * sizeof(float); }

// bg_implicit_network (network_bg.py:85-88): pts (P,4) -> out0 (P) = output[:,0], feat_tiles (svs_feat_tiles_bytes);
// training: hbuf (svs_sdf_hbuf_bytes) and ghat7 (svs_block_bytes(P,1)) for the backward, both or neither.
int svs_bg_sdf_eval(const float* pts, int n_points, const float* stream, int precision, float* out0, float* feat_tiles,
                    float* hbuf, float* ghat7, float* pebuf, void* hip_stream) {
  if (!is_h2(precision) && precision != kFmtF32) { set_error("svs_bg_sdf_eval: unknown precision %d", precision); return SVS_EINVAL; }
  if (!pts || n_points <= 0 || !stream || !out0 || !feat_tiles || (!hbuf != !ghat7) || (!hbuf != !pebuf)) {
    set_error("svs_bg_sdf_eval: null/invalid argument"); return SVS_EINVAL;
  }
  BgSdfArgs a{pts, n_points, reinterpret_cast<const f32x4*>(stream), out0, feat_tiles, hbuf, ghat7, pebuf};
  if (precision == kFmtF32) return launch_bg_sdf_f32(a, (hipStream_t)hip_stream);
  static int once = set_lds(bg_sdf_h2_kernel<false, false>, kLdsBytes, "svs_bg_sdf_eval") |
                    set_lds(bg_sdf_h2_kernel<true, false>, kLdsBytes, "svs_bg_sdf_eval") |
                    set_lds(bg_sdf_h2_kernel<true, true>, kLdsBytes, "svs_bg_sdf_eval");
  if (once) return once;
  const dim3 grid((n_points + kWgPts - 1) / kWgPts);
  if (!hbuf) bg_sdf_h2_kernel<false, false><<<grid, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  else if (precision == kFmtF16x2) bg_sdf_h2_kernel<true, true><<<grid, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  else bg_sdf_h2_kernel<true, false><<<grid, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_bg_sdf_eval");
}

// bg_rendering_network (network_bg.py:91-93): view_dirs (n_rays,3) when view_S > 0 (points per ray) else (P,3)
int svs_bg_rgb_eval(int n_points, const float* view_dirs, int view_S, const float* feat_tiles, const float* stream,
                    int precision, float* rgb, float* rbuf, void* hip_stream) {
  if (!is_h2(precision) && precision != kFmtF32) { set_error("svs_bg_rgb_eval: unknown precision %d", precision); return SVS_EINVAL; }
  if (n_points <= 0 || !view_dirs || !feat_tiles || !stream || !rgb || view_S < 0 || (view_S > 0 && n_points % view_S)) {
    set_error("svs_bg_rgb_eval: null/invalid argument"); return SVS_EINVAL;
  }
  BgRgbArgs a{n_points, view_dirs, view_S, feat_tiles, reinterpret_cast<const f32x4*>(stream), rgb, rbuf};
  if (precision == kFmtF32) return launch_bg_rgb_f32(a, (hipStream_t)hip_stream);
  constexpr int lds = 2 * kBgRgbBufF4 * 16;
  static int once = set_lds(bg_rgb_h2_kernel<true>, lds, "svs_bg_rgb_eval") | set_lds(bg_rgb_h2_kernel<false>, lds, "svs_bg_rgb_eval");
  if (once) return once;
  const int grid = (n_points + kWgPts - 1) / kWgPts;
  if (rbuf && precision == kFmtF16x2) bg_rgb_h2_kernel<true><<<grid, kThreads, lds, (hipStream_t)hip_stream>>>(a);
  else bg_rgb_h2_kernel<false><<<grid, kThreads, lds, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_bg_rgb_eval");
}

// bg_rendering_network backward (stream: which = 8): zbuf = 2 blocks per tile, ZERO-INITIALISED by the caller once
int svs_bg_rgb_bwd(int n_points, const float* d_rgb, const float* rgb, const float* rbuf, const float* stream, int precision,
                   float* zbuf, float* feat_bar, float* absmax, void* hip_stream) {
  if (!is_h2(precision) && precision != kFmtF32) { set_error("svs_bg_rgb_bwd: unknown precision %d", precision); return SVS_EINVAL; }
  if (n_points <= 0 || !d_rgb || !rgb || !rbuf || !stream || !zbuf || !feat_bar || (!absmax && is_h2(precision))) {
    set_error("svs_bg_rgb_bwd: null/invalid argument"); return SVS_EINVAL;
  }
  BgRgbBwdArgs a{n_points, d_rgb, rgb, rbuf, reinterpret_cast<const f32x4*>(stream), zbuf, feat_bar, absmax};
  if (precision == kFmtF32) return launch_bg_rgb_bwd_f32(a, (hipStream_t)hip_stream);
  static int once = set_lds(bg_rgb_bwd_h2_kernel<true>, kLdsBytes, "svs_bg_rgb_bwd") | set_lds(bg_rgb_bwd_h2_kernel<false>, kLdsBytes, "svs_bg_rgb_bwd");
  if (once) return once;
  const int grid = (n_points + kWgPts - 1) / kWgPts;
  if (precision == kFmtF16x2) bg_rgb_bwd_h2_kernel<true><<<grid, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  else bg_rgb_bwd_h2_kernel<false><<<grid, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_bg_rgb_bwd");
}

// bg_implicit_network backward (stream: which = 6): d_out0 (P) = d loss / d output[:,0], feat_bar (1 block per tile),
// hbuf / ghat7 from svs_bg_sdf_eval -> abuf (8 blocks per tile: abar_0..abar_7), sbar_out (padded P)
int svs_bg_sdf_bwd(int n_points, const float* d_out0, const float* feat_bar, const float* hbuf, const float* ghat7,
                   const float* stream, int precision, float* abuf, float* sbar_out, float* absmax, void* hip_stream) {
  if (!is_h2(precision) && precision != kFmtF32) { set_error("svs_bg_sdf_bwd: unknown precision %d", precision); return SVS_EINVAL; }
  if (n_points <= 0 || n_points % 32 || !d_out0 || !feat_bar || !hbuf || !ghat7 || !stream || !abuf || !sbar_out ||
      (!absmax && is_h2(precision))) {
    set_error("svs_bg_sdf_bwd: null/invalid argument (n_points must be a multiple of 32)"); return SVS_EINVAL;
  }
  SdfBwdBArgs a{n_points, d_out0, nullptr, feat_bar, n_points / 32, hbuf, nullptr, nullptr, nullptr,
                reinterpret_cast<const f32x4*>(stream), abuf, sbar_out, absmax, ghat7, (size_t)kBlockF};
  if (precision == kFmtF32) return launch_bg_bwd_b_f32(a, (hipStream_t)hip_stream);
  return launch_bg_bwd_b_h2(a, precision == kFmtF16x2, (hipStream_t)hip_stream);
}

}  // extern "C"
