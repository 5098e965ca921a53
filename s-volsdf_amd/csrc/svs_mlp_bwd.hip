// Training backward of the fused MLPs (reverse-mode derivative of svs_mlp.hip, written by hand).
//
// The reference obtains these gradients from torch.autograd (loss.backward() at volsdf/vsdf.py:215), including the
// double-backward through `gradients = autograd.grad(sdf, x, create_graph=True)` (volsdf/model/network.py:115-121):
// the normals feed the radiance MLP and the eikonal loss, so d loss / d theta needs the derivative of a
// quantity that is itself a backward pass.
//
// Notation: h_{l+1} = softplus(a_l), a_l = W_l h_l + b_l (l = 0..7), sdf = a_8[0], feat = a_8[1:];
// gradient pass: g(h_8) = W_8[0,:], ghat_l = g(h_{l+1}) * s'(a_l), g(h_l) = W_l^T ghat_l, n = J_PE^T g(h_0).
// Given sbar = dL/dsdf, fbar = dL/dfeat, nbar = dL/dn:
//   pass A (bottom-up, "second-order sweep"):  u_0 = J_PE nbar;  v_l = W_l u_l;  u_{l+1} = v_l * s'(a_l);
//                                              a2_l = v_l * g(h_{l+1}) * s''(a_l) = v_l * ghat_l * 100 (1 - s'(a_l))
//   pass B (top-down, ordinary backprop):      hbar_8 = sbar W_8[0,:] + W_8[1:,:]^T fbar;
//                                              abar_l = hbar_{l+1} * s'(a_l) + a2_l;  hbar_l = W_l^T abar_l
//   weights (svs_wgrad):                       dW_l = abar_l h_l^T + ghat_l u_l^T,  db_l = sum abar_l
// with s' = 1 - exp(-100 h), s'' = 100 s' (1 - s').  Both passes reuse the forward machinery (transposed
// activations in registers, LDS-DMA weight streaming, float32 MFMA) with bias-free / transposed weight streams.
#include "svs_mlp_dev.h"
#include "svs_mlp_bwd_args.h"
#include "svs_ticket.h"

namespace svs {
namespace mlp {

// ==============================================================================================================
// radiance MLP backward
// ==============================================================================================================

__global__ __launch_bounds__(kThreads, 1) void rgb_bwd_kernel(RgbBwdArgs a) {
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
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float o = a.rgb[3 * pc + c];
    dz[c] = (livep && half == 0) ? a.d_rgb[3 * pc + c] * o * (1.0f - o) : 0.0f;   // through the sigmoid (network.py:189)
  }
  {  // zbar_4: rows 0..2 live in registers 0..2 of half 0, tile 0
    f32x16 z4 = (f32x16)(0.0f);
    z4[0] = dz[0]; z4[1] = dz[1]; z4[2] = dz[2];
    store_tile(zb + 4 * LS, 0, lane, z4);
  }
  st.advance();
  f32x16 x[8], y[8];
  // rbar_4 = W_4^T zbar_4 (3 k-steps per tile), masked by r_4 > 0 -> zbar_3.  All of r_4 is requested up front.
  st.prefetch<kChunkF4>();
  load_tile_regs(rb + 3 * LS, y, lane);
  {
    const f32x4* c = st.cur_buf();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 w = c[t * 64 + lane];
      f32x16 acc = (f32x16)(0.0f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], dz[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], dz[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], dz[2], acc, 0, 0, 0);
      x[t] = acc;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) y[t][i] = y[t][i] > 0.0f ? x[t][i] : 0.0f;
  store_tile_regs(zb + 3 * LS, y, lane);
  st.advance();
  // layers 3..1: rbar_l = W_l^T zbar_l, fused per tile with zbar_{l-1} = rbar_l * [r_l > 0]; the r_l tile is requested
  // before the tile's MFMAs and the zbar tile is stored at the top of the next tile
  for (int l = 3; l >= 1; --l) {
    const float* rblk = rb + (size_t)(l - 1) * LS;
    float* zblk = zb + (size_t)(l - 1) * LS;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t > 0) store_tile(zblk, t - 1, lane, x[t - 1]);
      const f32x16 r = load_tile(rblk, t, lane);
      st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma<128>(st.cur_buf(), y, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) x[t][i] = r[i] > 0.0f ? acc[i] : 0.0f;
      st.advance();
    }
    store_tile(zblk, 7, lane, x[7]);
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = x[t];
  }
  // layer 0: the input gradients from zbar_0 (feature rows: tiles 0..7, extras: tile 8)
  float* fb = a.feat_bar + (size_t)wtile * kBlockF;
  f32x16 pend;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (t > 0) store_tile(fb, t - 1, lane, pend);        // deferred store of the previous feature-gradient tile
    if (t < 8) st.prefetch<kChunkF4>();
    const f32x16 acc = tile_mma<128>(st.cur_buf(), y, lane);
    if (t < 8) { pend = acc; st.advance(); }
    else if (half == 1 && livep) {
      // extra rows 12,13,14 (normals, network.py:175) = local rows rho(4..6)+4 of the extras tile
      a.d_normals[3 * p] = acc[4]; a.d_normals[3 * p + 1] = acc[5]; a.d_normals[3 * p + 2] = acc[6];
    }
  }
}

// ==============================================================================================================
// SDF MLP backward, pass A
// ==============================================================================================================

// a 39-vector in PE order as a 2-tile accumulator-layout block (rows q = 32*tile + rho(r) + 4*half)
__device__ __forceinline__ void store_pe_block(float* __restrict__ block, const float* vec40, int lane, int half) {
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
}

__global__ __launch_bounds__(kThreads, 1) void sdf_bwd_a_kernel(SdfBwdAArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.src.P ? p : a.src.P - 1;

  st.prefetch<kChunk0F4>();
  float x0, x1, x2;
  load_point(a.src, p, x0, x1, x2);
  PosEnc pe;
  pe.compute(x0, x1, x2);
  float nb[3] = {a.d_grad[3 * pc], a.d_grad[3 * pc + 1], a.d_grad[3 * pc + 2]};
  if (p >= a.src.P || (a.clamp_mask && a.clamp_mask[pc])) { nb[0] = nb[1] = nb[2] = 0.0f; }
  // u_0[q] = d PE_q / d x_{c(q)} * nbar_{c(q)}
  PosEnc u0;
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
  }
  const size_t LS = block_stride();
  const float* hb = a.hbuf + (size_t)wtile * kBlockF;
  const float* gb = a.gbuf + (size_t)wtile * kBlockF;
  float* ub = a.ubuf + (size_t)wtile * kBlockF;
  float* a2 = a.a2buf + (size_t)wtile * kBlockF;
  store_pe_block(ub, u0.v, lane, half);
  store_pe_block(a.pebuf + (size_t)wtile * kBlockF, pe.v, lane, half);
  st.advance();

  f32x16 x[8], y[8];
  auto epilogue = [&](int l, int t, const f32x16& v, const f32x16& h, const f32x16& g, f32x16& u_next, f32x16& a2v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float s1 = dsoftplus_from_h(h[i]);
      u_next[i] = v[i] * s1;
      a2v[i] = v[i] * g[i] * (100.0f * (1.0f - s1));      // g = ghat = g(h_{l+1}) s', and s'' = 100 s' (1 - s')
    }
    if (l == 3 && t == 6) zero_splice_rows_tile6(a2v, half);
  };
  // Stores of tile t-1 (a2 and u) are issued at the top of tile t, before its loads and weight prefetch.
  f32x16 pend_a2;
  // layer 0
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t > 0) { store_tile(a2, t - 1, lane, pend_a2); store_tile(ub + LS, t - 1, lane, x[t - 1]); }
    const f32x16 h = load_tile(hb, t, lane), g = load_tile(gb, t, lane);   // in flight during the MFMAs
    if (t < 7) st.prefetch<kChunk0F4>(); else st.prefetch<kChunkF4>();
    const f32x16 v = tile_mma_pe(st.cur_buf(), u0, lane, half);
    __builtin_amdgcn_sched_barrier(0);   // keep the epilogue (and its vmcnt wait) behind the MFMAs, see Stream
    epilogue(0, t, v, h, g, x[t], pend_a2);
    st.advance();
  }
  store_tile(a2, 7, lane, pend_a2);
  store_tile(ub + LS, 7, lane, x[7]);
  for (int l = 1; l < 8; ++l) {
    float* a2l = a2 + (size_t)l * LS;
    float* ul = ub + (size_t)(l + 1) * LS;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t == 7 && l == 3) break;
      if (t > 0) {
        store_tile(a2l, t - 1, lane, pend_a2);
        if (!(l == 3 && t - 1 == 6)) store_tile(ul, t - 1, lane, y[t - 1]);
      }
      const f32x16 h = load_tile(hb + (size_t)l * LS, t, lane), g = load_tile(gb + (size_t)l * LS, t, lane);
      if (!(l == 7 && t == 7)) st.prefetch<kChunkF4>();
      const f32x16 v = tile_mma<128>(st.cur_buf(), x, lane);
      __builtin_amdgcn_sched_barrier(0);
      epilogue(l, t, v, h, g, y[t], pend_a2);
      if (!(l == 7 && t == 7)) st.advance();
    }
    if (l == 3) {
      splice_skip(y, u0, half);                       // u_4 rows >= 217 carry u_0 (skip connection)
      store_tile(a2l, 6, lane, pend_a2);
      store_tile(a2l, 7, lane, (f32x16)(0.0f));
      store_tile(ul, 6, lane, y[6]);
      store_tile(ul, 7, lane, y[7]);
    } else {
      store_tile(a2l, 7, lane, pend_a2);
      store_tile(ul, 7, lane, y[7]);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = y[t];
  }
}

// ==============================================================================================================
// SDF MLP backward, pass B
// ==============================================================================================================

__global__ __launch_bounds__(kThreads, 1) void sdf_bwd_b_kernel(SdfBwdBArgs a) {
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
  const float* gb = a.gbuf + (size_t)wtile * kBlockF;
  const float* a2 = a.a2buf + (size_t)wtile * kBlockF;
  float* ab = a.abuf + (size_t)wtile * kBlockF;
  const bool has_f = a.feat_bar && wtile < a.n_feat_tiles;
  f32x16 x[8], y[8];
  if (has_f) load_tile_regs(a.feat_bar + (size_t)wtile * kBlockF, y, lane);
  else {
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = (f32x16)(0.0f);
  }
  st.advance();
  // hbar_8 = W8[1:,:]^T fbar + sbar * W8[0,:], fused with abar_7 = hbar_8 * s'(a_7) + a2_7 (gbuf block 7 holds
  // ghat_7 = W8[0,:] * s'(a_7)); the tiles of h_8 / a2_7 / ghat_7 are requested before the MFMAs of the tile
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t > 0) store_tile(ab + 7 * LS, t - 1, lane, x[t - 1]);    // deferred store
    const f32x16 w0 = load_tile(gb + 7 * LS, t, lane);
    const f32x16 h = load_tile(hb + 7 * LS, t, lane);
    const f32x16 s2 = load_tile(a2 + 7 * LS, t, lane);
    st.prefetch<kChunkF4>();
    const f32x16 acc = tile_mma<128>(st.cur_buf(), y, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[t][i] = acc[i] * dsoftplus_from_h(h[i]) + sbar * w0[i] + s2[i];
    st.advance();
  }
  store_tile(ab + 7 * LS, 7, lane, x[7]);
  for (int l = 7; l >= 1; --l) {
    // x = abar_l; hbar_l = W_l^T abar_l, fused with abar_{l-1} = hbar_l * s'(a_{l-1}) + a2_{l-1}
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t > 0) store_tile(ab + (size_t)(l - 1) * LS, t - 1, lane, y[t - 1]);   // deferred store
      const f32x16 h = load_tile(hb + (size_t)(l - 1) * LS, t, lane);
      const f32x16 s2 = load_tile(a2 + (size_t)(l - 1) * LS, t, lane);
      if (!(l == 1 && t == 7)) st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma<128>(st.cur_buf(), x, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) y[t][i] = acc[i] * dsoftplus_from_h(h[i]) + s2[i];
      if (l == 4 && t == 7) y[7] = (f32x16)(0.0f);          // abar_3 rows >= 217: the PE splice rows of h_4
      if (l == 4 && t == 6) zero_splice_rows_tile6(y[6], half);
      if (!(l == 1 && t == 7)) st.advance();
    }
    store_tile(ab + (size_t)(l - 1) * LS, 7, lane, y[7]);
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = y[t];
  }
}

// d loss / d W_8[0,:] = sum_p (sbar_p h_8[:,p] + u_8[:,p]);  d loss / d b_8[0] = sum_p sbar_p.
// HBM-bound (two activation blocks per tile are read once).  A wave owns a quarter of the 256 rows (8 float4 per lane
// and block) and grid-strides over the tiles, with the 16 loads of a tile in flight together; the sum over the 32
// points of a lane half goes through LDS once per workgroup, then float atomics into out[257] (index 256 = bias).
__global__ __launch_bounds__(256) void lin8_row0_kernel(const float* __restrict__ hbuf, const float* __restrict__ ubuf,
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
    const float live = p < P ? 1.0f : 0.0f;
    const f32x4* h = reinterpret_cast<const f32x4*>(hbuf + ((size_t)7 * n_tiles_pad + t) * kBlockF) + quarter * 8 * 64 + lane;
    f32x4 hv[8], uv[8];
    if (ubuf) {
      const f32x4* u = reinterpret_cast<const f32x4*>(ubuf + ((size_t)8 * n_tiles_pad + t) * kBlockF) + quarter * 8 * 64 + lane;
#pragma unroll
      for (int i4 = 0; i4 < 8; ++i4) uv[i4] = u[i4 * 64];
    } else {
#pragma unroll
      for (int i4 = 0; i4 < 8; ++i4) uv[i4] = (f32x4)(0.0f);
    }
#pragma unroll
    for (int i4 = 0; i4 < 8; ++i4) hv[i4] = h[i4 * 64];
#pragma unroll
    for (int i4 = 0; i4 < 8; ++i4)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * i4 + j] += sb * hv[i4][j] + live * uv[i4][j];
    if (quarter == 0 && lane < 32) bsum += sb;
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) red[quarter][i][lane] = acc[i];
  __syncthreads();
  det::wait_turn(ticket, blockIdx.x);
  // thread -> (quarter, register i, half): sum its 32 points
  {
    const int q = threadIdx.x >> 6, i = (threadIdx.x >> 1) & 31, hf = threadIdx.x & 1;
    float v = 0.0f;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) v += red[q][i][32 * hf + c];
    const int ii = 32 * q + i;                       // accumulator register index 0..127 of the block
    atomicAdd(&out[32 * (ii / 16) + rho(ii % 16) + 4 * hf], v);
  }
  if (quarter == 0) {
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) bsum += __shfl_xor(bsum, d);
    if (lane == 0) atomicAdd(&out[256], bsum);
  }
  det::pass_turn(ticket, blockIdx.x);
}

// ==============================================================================================================
// kernel-order weight gradients -> parameter gradients (incl. the weight-norm backward, network.py:64-65)
// ==============================================================================================================
struct UnpackArgs {
  const float* dWk;    // [256][ldw] kernel-order gradient of the EFFECTIVE weight
  const float* dbk;    // [256] or nullptr
  int ldw;
  int map;             // 0 identity, 1 SDF lin4 (skip splice, 1/sqrt2), 2 radiance lin0 (extras first)
  int rows, cols, row_off;   // parameter shape; row_off: first parameter row covered (SDF lin8: 1)
  const float* v; const float* g;     // weight_v (or plain weight), weight_g or nullptr
  float* grad_v; float* grad_g; float* grad_b;
  const float* row0;   // SDF lin8: [257] gradient of row 0 (+ bias at [256]) or nullptr
};

__device__ __forceinline__ int kernel_col(int map, int c) {
  if (map == 1) return c < 217 ? c : (c < 249 ? 224 + (c - 217) : 217 + (c - 249));
  if (map == 2) return c < 15 ? 256 + c : c - 15;
  if (map == 3) {     // bg lin4: cat[h(172), PE(84)] -> rows 224.. = PE[0..31], 192.. = PE[32..63], 172..191 = PE[64..83]
    if (c < 172) return c;
    const int e = c - 172;
    return e < 32 ? 224 + e : (e < 64 ? 192 + (e - 32) : 172 + (e - 64));
  }
  if (map == 4) return c < 27 ? 256 + c : c - 27;     // bg radiance lin0: cat[PE4(view)(27), feature(256)]
  return c;
}

// one wave per parameter row
__device__ __forceinline__ void unpack_row(const UnpackArgs& a, int o, int lane) {
  if (o >= a.rows) return;
  const float fac = (a.map == 1 || a.map == 3) ? 0.70710678118654752f : 1.0f;
  const bool is_row0 = a.row0 && o == 0;
  const int ko = o - a.row_off;
  auto dweff = [&](int c) -> float {
    if (is_row0) return a.row0[c];
    return a.dWk[(size_t)ko * a.ldw + kernel_col(a.map, c)] * fac;
  };
  if (lane == 0 && a.grad_b) a.grad_b[o] = is_row0 ? a.row0[256] : (a.dbk ? a.dbk[ko] : 0.0f);
  if (!a.g) {
    for (int c = lane; c < a.cols; c += 64) a.grad_v[(size_t)o * a.cols + c] = dweff(c);
    return;
  }
  double nn = 0.0, dot = 0.0;
  for (int c = lane; c < a.cols; c += 64) {
    const double vv = (double)a.v[(size_t)o * a.cols + c];
    nn += vv * vv; dot += vv * (double)dweff(c);
  }
  for (int d = 32; d >= 1; d >>= 1) { nn += __shfl_xor(nn, d); dot += __shfl_xor(dot, d); }
  const double nrm = __builtin_sqrt(nn);
  const float gg = a.g[o];
  const float k1 = (float)((double)gg / nrm), k2 = (float)((double)gg * dot / (nrm * nn));
  if (lane == 0) a.grad_g[o] = (float)(dot / nrm);
  for (int c = lane; c < a.cols; c += 64)
    a.grad_v[(size_t)o * a.cols + c] = k1 * dweff(c) - k2 * a.v[(size_t)o * a.cols + c];
}

__global__ __launch_bounds__(256) void unpack_wgrad_kernel(UnpackArgs a) {
  unpack_row(a, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// all layers of a step in one launch: block b belongs to the job whose block range contains it
constexpr int kMaxUnpackJobs = 24;
struct UnpackMulti {
  UnpackArgs job[kMaxUnpackJobs];
  int first_block[kMaxUnpackJobs + 1];
  int n;
};
__global__ __launch_bounds__(256) void unpack_wgrad_multi_kernel(UnpackMulti m) {
  int j = 0;
  while (j + 1 < m.n && (int)blockIdx.x >= m.first_block[j + 1]) ++j;
  unpack_row(m.job[j], ((int)blockIdx.x - m.first_block[j]) * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

}  // namespace mlp
}  // namespace svs

using namespace svs;
using namespace svs::mlp;

namespace {
int tiles_of(int n_points) { return (n_points + kWgPts - 1) / kWgPts * kWaves; }
template <typename K>
int set_lds_b(K kernel, int bytes, const char* who) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e)); return (int)e; }
  return SVS_OK;
}
}  // namespace

struct svs_unpack_job {          // include/svolsdf_hip.h
  const float *dWk, *dbk;
  int ldw, map, rows, cols, row_off;
  const float *weight_v, *weight_g, *row0;
  float *grad_v, *grad_g, *grad_b;
};

extern "C" {

// blocks_per_tile slots of kBlockF floats per wave tile, followed by one 64-float record per slot (the per-point scales and
// maxima of scaled blocks, csrc/svs_blocks_h2.h)
size_t svs_block_bytes(int n_points, int blocks_per_tile) {
  return (size_t)tiles_of(n_points) * blocks_per_tile * (kBlockF + 64) * sizeof(float);
}
size_t svs_rgb_zbuf_bytes(int n_points) { return svs_block_bytes(n_points, 5); }
size_t svs_sdf_ubuf_bytes(int n_points) { return svs_block_bytes(n_points, 9); }
size_t svs_sdf_gbuf_bytes(int n_points) { return svs_block_bytes(n_points, 8); }

int svs_rgb_bwd(int n_points, const float* d_rgb, const float* rgb, const float* rbuf, const float* stream, int precision,
                float* zbuf, float* feat_bar, float* d_normals, float* absmax, void* hip_stream) {
  if (!d_rgb || !rgb || !rbuf || !stream || !zbuf || !feat_bar || !d_normals || n_points <= 0) {
    set_error("svs_rgb_bwd: null/invalid argument"); return SVS_EINVAL;
  }
  RgbBwdArgs a{n_points, d_rgb, rgb, rbuf, reinterpret_cast<const f32x4*>(stream), zbuf, feat_bar, d_normals, absmax};
  if (is_h2(precision)) {
    if (!absmax) { set_error("svs_rgb_bwd: fp16x2 needs absmax"); return SVS_EINVAL; }
    return launch_rgb_bwd_h2(a, precision == kFmtF16x2, (hipStream_t)hip_stream);
  }
  if (precision != kFmtF32) { set_error("svs_rgb_bwd: unknown precision %d", precision); return SVS_EINVAL; }
  static int once = set_lds_b(rgb_bwd_kernel, kLdsBytes, "svs_rgb_bwd");
  if (once) return once;
  rgb_bwd_kernel<<<(n_points + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_rgb_bwd");
}

int svs_sdf_bwd_a(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                  int S, int n_rays, const float* d_grad, const unsigned char* clamp_mask, const float* hbuf,
                  const float* gbuf, const float* stream, int precision, float* ubuf, float* a2buf, float* pebuf,
                  float* absmax, void* hip_stream) {
  SdfBwdAArgs a;
  const bool h2 = is_h2(precision);      // fp16x2: gbuf is not read and a2buf not written (pass B re-forms a2)
  if (n_points < 0 || n_rays < 0 || (n_points == 0 && n_rays == 0) || (n_points > 0 && !points) ||
      (n_rays > 0 && !(cam && dirs && z && S > 0)) || !d_grad || !hbuf || (!h2 && !gbuf) || !stream || !ubuf || (!h2 && !a2buf) ||
      !pebuf) {
    set_error("svs_sdf_bwd_a: null/invalid argument"); return SVS_EINVAL;
  }
  a.src.pts = points; a.src.cam = cam; a.src.dirs = dirs; a.src.z = z; a.src.cam_stride = cam_stride;
  a.src.S = S > 0 ? S : 1; a.src.n_ray = n_rays * (S > 0 ? S : 0); a.src.P = a.src.n_ray + n_points;
  a.d_grad = d_grad; a.clamp_mask = clamp_mask; a.hbuf = hbuf; a.gbuf = gbuf;
  a.stream = reinterpret_cast<const f32x4*>(stream); a.ubuf = ubuf; a.a2buf = a2buf; a.pebuf = pebuf;
  a.absmax = absmax;
  if (h2) {
    if (!absmax) { set_error("svs_sdf_bwd_a: fp16x2 needs absmax"); return SVS_EINVAL; }
    return launch_sdf_bwd_a_h2(a, precision == kFmtF16x2, (hipStream_t)hip_stream);
  }
  if (precision != kFmtF32) { set_error("svs_sdf_bwd_a: unknown precision %d", precision); return SVS_EINVAL; }
  static int once = set_lds_b(sdf_bwd_a_kernel, kLdsBytes, "svs_sdf_bwd_a");
  if (once) return once;
  sdf_bwd_a_kernel<<<(a.src.P + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_sdf_bwd_a");
}

int svs_sdf_bwd_b(int n_points, const float* d_sdf, const unsigned char* clamp_mask, const float* feat_bar,
                  int n_feat_points, const float* hbuf, const float* gbuf, const float* a2buf, const float* ubuf,
                  const float* stream, int precision, float* abuf, float* sbar_out, float* absmax, void* hip_stream) {
  const bool h2 = is_h2(precision);      // fp16x2: a2 is re-formed from ubuf and gbuf (+ their records); float32: read from a2buf
  if (!hbuf || !gbuf || (h2 ? !ubuf : !a2buf) || !stream || !abuf || !sbar_out || n_points <= 0 || n_feat_points % 32) {
    set_error("svs_sdf_bwd_b: null/invalid argument (n_feat_points must be a multiple of 32)"); return SVS_EINVAL;
  }
  SdfBwdBArgs a{n_points, d_sdf, clamp_mask, feat_bar, n_feat_points / 32, hbuf, gbuf, a2buf, ubuf,
                reinterpret_cast<const f32x4*>(stream) + kSdfTrainPassBF4, abuf, sbar_out, absmax,
                gbuf + 7 * (size_t)tiles_of(n_points) * kBlockF, (size_t)kBlockF};     // w0 = ghat_7: block 7 of gbuf ([block][tile])
  if (h2) {
    if (!absmax) { set_error("svs_sdf_bwd_b: fp16x2 needs absmax"); return SVS_EINVAL; }
    return launch_sdf_bwd_b_h2(a, precision == kFmtF16x2, (hipStream_t)hip_stream);
  }
  if (precision != kFmtF32) { set_error("svs_sdf_bwd_b: unknown precision %d", precision); return SVS_EINVAL; }
  static int once = set_lds_b(sdf_bwd_b_kernel, kLdsBytes, "svs_sdf_bwd_b");
  if (once) return once;
  sdf_bwd_b_kernel<<<(n_points + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_sdf_bwd_b");
}

int svs_lin8_row0_grad(const float* hbuf, const float* ubuf, const float* sbar, int n_points, int precision, float* out257,
                       void* hip_stream) {
  if (!hbuf || !sbar || !out257 || n_points <= 0) { set_error("svs_lin8_row0_grad: bad argument"); return SVS_EINVAL; }
  if (is_h2(precision)) return launch_lin8_row0_h2(hbuf, ubuf, sbar, n_points, tiles_of(n_points), out257, precision == kFmtF16x2, (hipStream_t)hip_stream);
  if (precision != kFmtF32) { set_error("svs_lin8_row0_grad: unknown precision %d", precision); return SVS_EINVAL; }
  const int n_tiles = (n_points + 31) / 32;
  const int grid = n_tiles < 1024 ? n_tiles : 1024;
  const det::Ticket ticket{det::take_slots(1), 0u, (unsigned)grid};
  lin8_row0_kernel<<<grid, 256, 0, (hipStream_t)hip_stream>>>(hbuf, ubuf, sbar, n_tiles, tiles_of(n_points), n_points, out257, ticket);
  return check_launch("svs_lin8_row0_grad");
}

int svs_unpack_wgrad(const float* dWk, const float* dbk, int ldw, int map, int rows, int cols, int row_off,
                     const float* weight_v, const float* weight_g, const float* row0, float* grad_v, float* grad_g,
                     float* grad_b, void* hip_stream) {
  if (!dWk || !weight_v || !grad_v || rows < 1 || cols < 1 || map < 0 || map > 4 || (weight_g && !grad_g)) {
    set_error("svs_unpack_wgrad: bad argument"); return SVS_EINVAL;
  }
  UnpackArgs a{dWk, dbk, ldw, map, rows, cols, row_off, weight_v, weight_g, grad_v, grad_g, grad_b, row0};
  unpack_wgrad_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_unpack_wgrad");
}

int svs_unpack_wgrad_multi(const svs_unpack_job* jobs, int n_jobs, void* hip_stream) {
  if (!jobs || n_jobs < 1 || n_jobs > kMaxUnpackJobs) { set_error("svs_unpack_wgrad_multi: 1..%d jobs", kMaxUnpackJobs); return SVS_EINVAL; }
  UnpackMulti m;
  m.n = n_jobs;
  int blocks = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const svs_unpack_job& q = jobs[i];
    if (!q.dWk || !q.weight_v || !q.grad_v || q.rows < 1 || q.cols < 1 || q.map < 0 || q.map > 4 || (q.weight_g && !q.grad_g)) {
      set_error("svs_unpack_wgrad_multi: bad job %d", i); return SVS_EINVAL;
    }
    m.job[i] = UnpackArgs{q.dWk, q.dbk, q.ldw, q.map, q.rows, q.cols, q.row_off, q.weight_v, q.weight_g, q.grad_v, q.grad_g, q.grad_b, q.row0};
    m.first_block[i] = blocks;
    blocks += (q.rows + 3) / 4;
  }
  m.first_block[n_jobs] = blocks;
  unpack_wgrad_multi_kernel<<<blocks, 256, 0, (hipStream_t)hip_stream>>>(m);
  return check_launch("svs_unpack_wgrad_multi");
}

}  // extern "C"
