// The inverted-sphere background networks of VolSDFNetworkBG (volsdf/model/network_bg.py:31-35, 85-103) on float32 MFMAs
// (v_mfma_f32_32x32x2_f32): what SVS_MLP_PRECISION=f32 runs for config 4.  Same streams (svs_pack_stream with precision 0),
// same buffers and the same arrangement of tiles as the fp16x2 kernels of svs_bg_h2.hip; every activation block is a float32
// block in the layout of svs_mlp_dev.h (store_tile / load_tile), which is what the float32 weight-gradient kernel of
// svs_wgrad.hip and svs_lin8_row0_grad read.  Built from the float32 machinery of the foreground networks (svs_mlp.hip,
// svs_mlp_bwd.hip: one tile = 128 k-steps of 2 input rows, epilogue behind the tile's MFMAs, stores deferred by one tile).
#include "svs_bg_args.h"
#include "svs_mlp_host.h"

namespace svs {
namespace mlp {

// layer 0 of bg_implicit_network: the B operand of k-step s is PE row 2s (lanes 0-31) / 2s+1 (lanes 32-63); 48 k-steps
// (84 inputs padded to 96: the chunk is as large as its fp16x2 form, svs_mlp_layout.h kBgChunk0F4)
__device__ __forceinline__ f32x16 tile_mma_pe_bg(const f32x4* __restrict__ chunk, const PosEncBg& pe, int lane, int half) {
  f32x16 acc;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const f32x4 b = chunk[r4 * 64 + lane];
    acc[4 * r4 + 0] = b[0]; acc[4 * r4 + 1] = b[1]; acc[4 * r4 + 2] = b[2]; acc[4 * r4 + 3] = b[3];
  }
  const f32x4* a_ptr = chunk + kHdrF4 + lane;
#pragma unroll
  for (int s4 = 0; s4 < 12; ++s4) {
    const f32x4 a = a_ptr[s4 * 64];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = 4 * s4 + j;
      const float b = half ? pe.v[2 * s + 1] : pe.v[2 * s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b, acc, 0, 0, 0);
    }
  }
  return acc;
}

// the PE rows spliced into the layer-4 input (NetBg: rows 172..255): tile 7 = PE[0..31], tile 6 = PE[32..63], tile 5 local rows
// 12..31 = PE[64..83] (svs_mlp_h2_trunk.h TrunkEpi::b / splice_full_tiles; 1/sqrt(2) folded into the packed W4)
__device__ __forceinline__ void splice_skip_bg(f32x16* y, const PosEncBg& pe, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row0 = rho(r), row1 = rho(r) + 4;
    y[7][r] = half ? pe.v[row1] : pe.v[row0];
    y[6][r] = half ? pe.v[32 + row1] : pe.v[32 + row0];
    const int l0 = row0 - NetBg::kSpliceLocal, l1 = row1 - NetBg::kSpliceLocal;
    if (l0 >= 0 || l1 >= 0) {
      const float v0 = l0 >= 0 ? pe.v[64 + (l0 >= 0 ? l0 : 0)] : y[5][r];
      const float v1 = l1 >= 0 ? pe.v[64 + (l1 >= 0 ? l1 : 0)] : y[5][r];
      y[5][r] = half ? v1 : v0;
    }
  }
}
// rows >= 172 of a layer-4 input quantity that does not flow into lin3: tiles 6, 7 whole, tile 5 from local row 12 on
__device__ __forceinline__ void zero_splice_rows_bg(f32x16& v5, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool z0 = rho(r) >= NetBg::kSpliceLocal, z1 = rho(r) + 4 >= NetBg::kSpliceLocal;
    if (z0 || z1) { if (half ? z1 : z0) v5[r] = 0.0f; }
  }
}

// layers 0..7 (forward_trunk of svs_mlp_dev.h with the background network's geometry); on return x holds h_8
template <bool HBUF>
__device__ __forceinline__ void forward_trunk_bg(Stream& st, f32x16* x, f32x16* y, const PosEncBg& pe, int lane, int half,
                                                 float* __restrict__ hbuf) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (HBUF && t > 0) store_tile(hbuf, t - 1, lane, x[t - 1]);
    if (t < 7) st.prefetch<kBgChunk0F4>(); else st.prefetch<kChunkF4>();
    const f32x16 acc = tile_mma_pe_bg(st.cur_buf(), pe, lane, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = softplus100(acc[r]);
    st.advance();
  }
  if (HBUF) store_tile(hbuf, 7, lane, x[7]);
  for (int l = 1; l < 8; ++l) {
    float* hb = hbuf + (size_t)l * block_stride();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t == NetBg::kSpliceTile + 1 && l == 3) break;   // lin3 has 172 outputs = 6 tiles (the last one partial)
      if (HBUF && t > 0) store_tile(hb, t - 1, lane, y[t - 1]);
      st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma<128>(st.cur_buf(), x, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) y[t][r] = softplus100(acc[r]);
      st.advance();
    }
    if (l == 3) {
      splice_skip_bg(y, pe, half);
      if (HBUF) { store_tile(hb, 5, lane, y[5]); store_tile(hb, 6, lane, y[6]); store_tile(hb, 7, lane, y[7]); }
    } else if (HBUF) {
      store_tile(hb, 7, lane, y[7]);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = y[t];
  }
}

// bg_implicit_network (network_bg.py:85-88): out0, the feature vector; TRAIN: h_1..h_8, ghat_7 and the PE block as well
template <bool TRAIN>
__global__ __launch_bounds__(kThreads, 1) void bg_sdf_f32_kernel(BgSdfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
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
  if (TRAIN) {
    // h_0 = PE (84 rows) in PE order as a block (rows q = 32 tile + rho(r) + 4 half), zero beyond: the B operand of dW_0
    float* pb = a.pebuf + (size_t)wtile * kBlockF;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x16 v;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q0 = 32 * t + rho(r), q1 = q0 + 4;
        const float a0 = q0 < kBgPeDim ? pe.v[q0 < 96 ? q0 : 95] : 0.0f;
        const float a1 = q1 < kBgPeDim ? pe.v[q1 < 96 ? q1 : 95] : 0.0f;
        v[r] = half ? a1 : a0;
      }
      store_tile(pb, t, lane, v);
    }
  }
  st.advance();
  float* hb = TRAIN ? a.hbuf + (size_t)wtile * kBlockF : nullptr;
  f32x16 x[8], y[8];
  forward_trunk_bg<TRAIN>(st, x, y, pe, lane, half, hb);
  // ---- head: current chunk = VEC (W8 row 0 in C-layout order, b8[0])
  st.prefetch<kChunkF4>();                       // FEAT tile 0
  const float out0 = sdf_head(st.cur_buf(), x, lane);
  if (TRAIN) {
    // ghat_7 = W8[0,:] * softplus'(a_7): pass B's seed
    const f32x4* w_ptr = st.cur_buf() + kHdrF4 + lane;
#pragma unroll
    for (int s4 = 0; s4 < 32; ++s4) {
      const f32x4 w = w_ptr[s4 * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[s4 / 4][4 * (s4 % 4) + j] = w[j] * dsoftplus_from_h(x[s4 / 4][4 * (s4 % 4) + j]);
    }
    store_tile_regs(a.ghat7 + (size_t)wtile * kBlockF, y, lane);
  }
  st.advance();
  if (half == 0 && p < a.P) a.out0[p] = out0;
  // ---- feature vector = rows 1..256 of lin8 (no activation)
  float* ft = a.feat_tiles + (size_t)wtile * kBlockF;
  f32x16 pend;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t > 0) store_tile(ft, t - 1, lane, pend);     // deferred store (see forward_trunk)
    if (t < 7) st.prefetch<kChunkF4>();
    pend = tile_mma<128>(st.cur_buf(), x, lane);
    if (t < 7) st.advance();
  }
  store_tile(ft, 7, lane, pend);
}

// bg_rendering_network, mode 'nerf' (network.py:170-190 with bmvs.yaml:70-77): cat[PE4(view)(27), feature(256)] -> 128 ReLU
// -> 3, sigmoid.  The 32 (27 + padding) view rows are k-steps 128..143 of the layer-0 chunks.
__global__ __launch_bounds__(kThreads, 1) void bg_rgb_f32_kernel(BgRgbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  BgRgbStream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.P ? p : a.P - 1;

  st.prefetch<kBgRgbChunk0F4>();
  const float* vd = a.view + 3 * (size_t)(a.view_S > 0 ? pc / a.view_S : pc);
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
  float eb[16];   // B operands of k-steps 128..143: rows rho(r) / rho(r) + 4 of the 32 extra rows
#pragma unroll
  for (int r = 0; r < 16; ++r) eb[r] = half ? ex[rho(r) + 4] : ex[rho(r)];
  f32x16 x[8], y[4];
  load_tile_regs(a.feat_tiles + (size_t)wtile * kBlockF, x, lane);
  float* rb = a.rbuf ? a.rbuf + (size_t)wtile * kBgRbufF : nullptr;
  if (rb) {
    // the 32 extra rows as ONE accumulator-layout tile: the B operand of the weight-gradient kernel's extra columns
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
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (rb && t > 0) store_tile(rb, t - 1, lane, y[t - 1]);
    if (t < 3) st.prefetch<kBgRgbChunk0F4>(); else st.prefetch<kChunkF4>();
    const f32x4* chunk = st.cur_buf();
    f32x16 acc = tile_mma<128>(chunk, x, lane);
    const f32x4* a_ptr = chunk + kHdrF4 + 2048 + lane;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const f32x4 w = a_ptr[s4 * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], eb[4 * s4 + j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) y[t][r] = __builtin_fmaxf(acc[r], 0.0f);
    st.advance();
  }
  if (rb) {
    store_tile(rb, 3, lane, y[3]);
    // rows 128..255 of the r_1 block stay zero (the weight-gradient kernel reads whole 256-row blocks)
#pragma unroll
    for (int t = 4; t < 8; ++t) store_tile(rb, t, lane, (f32x16)(0.0f));
  }
  // ---- layer 1: 128 -> 3 as one tile (rows 0..2 live in registers 0..2 of lanes 0..31), sigmoid
  const f32x16 acc = tile_mma<64>(st.cur_buf(), y, lane);
  if (half == 0 && p < a.P) {
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb[3 * p + c] = 1.0f / (1.0f + __expf(-acc[c]));
  }
}

// bg_rendering_network backward: zbar_1 = d_rgb * sigmoid', zbar_0 = (W1^T zbar_1) * [r_1 > 0], fbar = W0[:, 27:]^T zbar_0
// (the view directions get no gradient)
__global__ __launch_bounds__(kThreads, 1) void bg_rgb_bwd_f32_kernel(BgRgbBwdArgs a) {
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

  st.prefetch<kW1TF4>();
  float dz[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float o = a.rgb[3 * pc + c];
    dz[c] = (livep && half == 0) ? a.d_rgb[3 * pc + c] * o * (1.0f - o) : 0.0f;
  }
  {  // zbar_1: rows 0..2 live in registers 0..2 of half 0, tile 0 (the other tiles of the block stay zero)
    f32x16 z1 = (f32x16)(0.0f);
    z1[0] = dz[0]; z1[1] = dz[1]; z1[2] = dz[2];
    store_tile(zb + (size_t)kBlockF, 0, lane, z1);
  }
  st.advance();
  st.prefetch<kChunkF4>();
  f32x16 y[4];
  {
    // rbar_1 = W_1^T zbar_1 (K = 3, the short W1T chunk), masked by r_1 > 0 -> zbar_0 (4 tiles)
    const f32x4* c = st.cur_buf();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x16 r = load_tile(rb, t, lane);
      const f32x4 w = c[t * 64 + lane];
      f32x16 acc = (f32x16)(0.0f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], dz[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], dz[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], dz[2], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) y[t][i] = r[i] > 0.0f ? acc[i] : 0.0f;
      store_tile(zb, t, lane, y[t]);
    }
  }
  st.advance();
  // fbar: 8 tiles of feature rows, K = 128 (64 k-steps)
  float* fb = a.feat_bar + (size_t)wtile * kBlockF;
  f32x16 pend;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t > 0) store_tile(fb, t - 1, lane, pend);
    if (t < 7) st.prefetch<kChunkF4>();
    pend = tile_mma<64>(st.cur_buf(), y, lane);
    if (t < 7) st.advance();
  }
  store_tile(fb, 7, lane, pend);
}

// bg_implicit_network backward: ordinary backprop (pass B of svs_mlp_bwd.hip without the second-order blocks; the seed
// block ghat_7 of a tile is at w0 + tile * w0_stride; the PE splice rows are the background network's)
__global__ __launch_bounds__(kThreads, 1) void bg_bwd_b_f32_kernel(SdfBwdBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.P ? p : a.P - 1;
  st.prefetch<kChunkF4>();
  const float sbar = (a.d_sdf && p < a.P) ? a.d_sdf[pc] : 0.0f;
  if (half == 0 && a.sbar_out) a.sbar_out[p] = sbar;
  const size_t LS = block_stride();
  const float* hb = a.hbuf + (size_t)wtile * kBlockF;
  const float* g7 = a.w0 + (size_t)wtile * a.w0_stride;
  float* ab = a.abuf + (size_t)wtile * kBlockF;
  const bool has_f = a.feat_bar && wtile < a.n_feat_tiles;
  f32x16 x[8], y[8];
  if (has_f) load_tile_regs(a.feat_bar + (size_t)wtile * kBlockF, y, lane);
  else {
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = (f32x16)(0.0f);
  }
  st.advance();
  // hbar_8 = W8[1:,:]^T fbar + sbar * W8[0,:], fused with abar_7 = hbar_8 * s'(a_7)  (ghat_7 = W8[0,:] * s'(a_7))
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t > 0) store_tile(ab + 7 * LS, t - 1, lane, x[t - 1]);
    const f32x16 w0 = load_tile(g7, t, lane);
    const f32x16 h = load_tile(hb + 7 * LS, t, lane);
    st.prefetch<kChunkF4>();
    const f32x16 acc = tile_mma<128>(st.cur_buf(), y, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[t][i] = acc[i] * dsoftplus_from_h(h[i]) + sbar * w0[i];
    st.advance();
  }
  store_tile(ab + 7 * LS, 7, lane, x[7]);
  for (int l = 7; l >= 1; --l) {
    // x = abar_l; hbar_l = W_l^T abar_l, fused with abar_{l-1} = hbar_l * s'(a_{l-1})
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t > 0) store_tile(ab + (size_t)(l - 1) * LS, t - 1, lane, y[t - 1]);
      const f32x16 h = load_tile(hb + (size_t)(l - 1) * LS, t, lane);
      if (!(l == 1 && t == 7)) st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma<128>(st.cur_buf(), x, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) y[t][i] = acc[i] * dsoftplus_from_h(h[i]);
      if (l == 4 && t > NetBg::kSpliceTile) y[t] = (f32x16)(0.0f);          // abar_3 rows >= 172: the PE splice rows of h_4
      if (l == 4 && t == NetBg::kSpliceTile) zero_splice_rows_bg(y[t], half);
      if (!(l == 1 && t == 7)) st.advance();
    }
    store_tile(ab + (size_t)(l - 1) * LS, 7, lane, y[7]);
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = y[t];
  }
}

int launch_bg_sdf_f32(const BgSdfArgs& a, hipStream_t s) {
  static int once = set_lds(bg_sdf_f32_kernel<false>, kLdsBytes, "svs_bg_sdf_eval") | set_lds(bg_sdf_f32_kernel<true>, kLdsBytes, "svs_bg_sdf_eval");
  if (once) return once;
  const dim3 grid((a.P + kWgPts - 1) / kWgPts);
  if (a.hbuf) bg_sdf_f32_kernel<true><<<grid, kThreads, kLdsBytes, s>>>(a);
  else bg_sdf_f32_kernel<false><<<grid, kThreads, kLdsBytes, s>>>(a);
  return check_launch("svs_bg_sdf_eval");
}
int launch_bg_rgb_f32(const BgRgbArgs& a, hipStream_t s) {
  constexpr int lds = 2 * kBgRgbBufF4 * 16;
  static int once = set_lds(bg_rgb_f32_kernel, lds, "svs_bg_rgb_eval");
  if (once) return once;
  bg_rgb_f32_kernel<<<(a.P + kWgPts - 1) / kWgPts, kThreads, lds, s>>>(a);
  return check_launch("svs_bg_rgb_eval");
}
int launch_bg_rgb_bwd_f32(const BgRgbBwdArgs& a, hipStream_t s) {
  static int once = set_lds(bg_rgb_bwd_f32_kernel, kLdsBytes, "svs_bg_rgb_bwd");
  if (once) return once;
  bg_rgb_bwd_f32_kernel<<<(a.P + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, s>>>(a);
  return check_launch("svs_bg_rgb_bwd");
}
int launch_bg_bwd_b_f32(const SdfBwdBArgs& a, hipStream_t s) {
  static int once = set_lds(bg_bwd_b_f32_kernel, kLdsBytes, "svs_bg_sdf_bwd");
  if (once) return once;
  bg_bwd_b_f32_kernel<<<(a.P + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, s>>>(a);
  return check_launch("svs_bg_sdf_bwd");
}

}  // namespace mlp
}  // namespace svs
