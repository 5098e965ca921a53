// Fused SDF ("implicit") and radiance ("rendering") MLP kernels for gfx950.
//
// Reference semantics: volsdf/model/network.py:71-131 (ImplicitNetwork) and :170-190
// (RenderingNetwork) of cvlab-stonybrook/s-volsdf; positional encoding volsdf/model/embedder.py:10-36.
//
// Design (DESIGN.md "MLP kernels"):
//   * activations live TRANSPOSED in registers: a wave owns 32 points (MFMA columns = lane & 31) and the
//     256 features of a layer are the rows of eight 32x32 accumulator tiles (8 x 16 = 128 VGPRs/lane).
//     The accumulator layout of v_mfma_f32_32x32x2_f32 (row = (r&3)+8(r>>2)+4(lane>>5)) is exactly the
//     B-operand layout of the next layer's MFMAs once the K order of the weights is permuted to match,
//     so a layer's output feeds the next layer with no LDS round trip and no cross-lane traffic;
//   * weights are pre-packed (svs_mlp_pack) in the order the MFMAs consume them, as 36 KiB "chunks"
//     (one output tile = 32 output features x K = 256), streamed global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4) into a double buffer shared by the 4 waves of a workgroup, and read back
//     with conflict-free ds_read_b128;
//   * exact float32 MFMA (v_mfma_f32_32x32x2_f32): the 1e-4 parity bound of the path rules out bf16.
#include "svs_mlp_host.h"
#include <cstdlib>
#include "svs_mlp_args.h"

namespace svs {
namespace mlp {


// ImplicitNetwork.get_sdf_vals (network.py:125-131), no grad: the sampler's evaluation.
__global__ __launch_bounds__(kThreads, 1) void sdf_only_kernel(SdfOnlyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.gate && a.gate[(size_t)(blockIdx.x * kWgPts / a.gate_points) * a.gate_stride] == 0) return;
  Stream st;
  st.g = a.stream;
  st.buf = reinterpret_cast<f32x4*>(smem);
  st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int p = (blockIdx.x * kWaves + wave) * kTilePts + (lane & 31);

  st.prefetch<kChunk0F4>();   // chunk 0 -> buffer 0 (overlaps the positional encoding below)
  float x0, x1, x2;
  load_point(a.src, p, x0, x1, x2);
  PosEnc pe;
  pe.compute(x0, x1, x2);
  st.advance();

  f32x16 x[8], y[8];
  forward_trunk<false>(st, x, y, pe, lane, half, nullptr);
  // the VEC chunk was prefetched by the last tile of layer 7
  float sdf = sdf_head(st.cur_buf(), x, lane);
  if (a.sphere_radius > 0.0f && p < a.clamp_n) {
    const float nrm = __builtin_sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
    sdf = __builtin_fminf(sdf, a.sphere_scale * (a.sphere_radius - nrm));
  }
  if (half == 0 && p < a.src.P) a.sdf[p] = sdf;
}

// ------------------------------------------------------------------------------------------------------
// ImplicitNetwork.get_outputs (network.py:105-123): sdf, feature vector and d sdf / d x in one launch.
// The input gradient is the reverse-mode product through the same MLP (the reference's autograd.grad,
// :115-121): g(h_8) = W8[0,:], g(a_l) = g(h_{l+1}) * softplus'(a_l), g(h_l) = W_l^T g(a_l).
// ------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(kThreads, 1) void sdf_full_kernel(SdfFullArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream;
  st.buf = reinterpret_cast<f32x4*>(smem);
  st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);

  st.prefetch<kChunk0F4>();
  float x0, x1, x2;
  load_point(a.src, p, x0, x1, x2);
  PosEnc pe;
  pe.compute(x0, x1, x2);
  st.advance();

  float* hb = a.hbuf + (size_t)wtile * kBlockF;          // block l of this tile: + l * block_stride()
  f32x16 x[8], y[8];
  forward_trunk<true>(st, x, y, pe, lane, half, hb);

  // ---- head: current chunk = VEC (W8 row 0 in C-layout order, b8[0])
  st.prefetch<kChunkF4>();                       // FEAT tile 0
  float sdf = sdf_head(st.cur_buf(), x, lane);
  {
    const f32x4* w_ptr = st.cur_buf() + kHdrF4 + lane;
#pragma unroll
    for (int s4 = 0; s4 < 32; ++s4) {
      const f32x4 w = w_ptr[s4 * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[s4 / 4][4 * (s4 % 4) + j] = w[j];   // g(h_8)
    }
  }
  st.advance();
  float* gb = a.gbuf ? a.gbuf + (size_t)wtile * kBlockF : nullptr;
  // ---- feature vector = rows 1..256 of lin8
  float* ft = a.feat_tiles ? a.feat_tiles + (size_t)wtile * 128 * 64 : nullptr;
  {
    f32x16 pend;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (ft && t > 0) store_tile(ft, t - 1, lane, pend);   // deferred store (see forward_trunk)
      st.prefetch<kChunkF4>();                                // FEAT t+1, or reverse L7 tile 0
      pend = tile_mma<128>(st.cur_buf(), x, lane);
      st.advance();
    }
    if (ft) store_tile(ft, 7, lane, pend);
  }
  // g(a_7) = g(h_8) * softplus'(a_7), h_8 still in x
#pragma unroll
  for (int i = 0; i < 128; ++i) y[i / 16][i % 16] *= dsoftplus_from_h(x[i / 16][i % 16]);
  if (gb) store_tile_regs(gb + 7 * block_stride(), y, lane);       // ghat_7 = W8[0,:] * softplus'(a_7)

  // ---- reverse layers 7..1
  f32x16 skip7;          // g(PE[0..31]) from the skip connection (tile 7 of g(h_4 spliced))
  f32x16 skip6;          // tile 6; only local rows 25..31 are PE[32..38]
  for (int l = 7; l >= 1; --l) {
    const float* hblk = hb + (size_t)(l - 1) * block_stride();
    float* gblk = gb ? gb + (size_t)(l - 1) * block_stride() : nullptr;
    f32x16 pend;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (gblk && t > 0) store_tile(gblk, t - 1, lane, pend);   // ghat_{l-1} tile t-1, deferred store
      const f32x16 h = load_tile(hblk, t, lane);   // issued before the MFMAs: arrives while they run
      st.prefetch<kChunkF4>();                     // next reverse chunk (the last one prefetches REV0 tile 0)
      const f32x16 acc = tile_mma<128>(st.cur_buf(), y, lane);   // g(h_l) rows 32t..32t+31
      __builtin_amdgcn_sched_barrier(0);   // keep the epilogue (and its vmcnt wait) behind the MFMAs
      if (l == 4 && t == 7) skip7 = acc;
      if (l == 4 && t == 6) skip6 = acc;
      // g(a_{l-1}) = g(h_l) * softplus'(a_{l-1}),  softplus' from the stored h_l
#pragma unroll
      for (int i = 0; i < 16; ++i) x[t][i] = acc[i] * dsoftplus_from_h(h[i]);
      pend = x[t];
      st.advance();
    }
    if (gblk) store_tile(gblk, 7, lane, pend);
    if (l == 4) {
      // rows >= 217 of h_4 are the PE splice, not softplus outputs: they do not flow into lin3
      x[7] = (f32x16)(0.0f);
      zero_splice_rows_tile6(x[6], half);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = x[t];
  }
  // ---- reverse layer 0: g(PE) = W0^T g(a_0) (+ skip), 2 tiles
  st.prefetch<kChunkF4>();
  f32x16 gpe0 = tile_mma<128>(st.cur_buf(), y, lane);
  st.advance();
  f32x16 gpe1 = tile_mma<128>(st.cur_buf(), y, lane);
  gpe0 += skip7;
  gpe1 += skip6;

  // ---- d/dx of the positional encoding
  float dx0 = 0.0f, dx1 = 0.0f, dx2 = 0.0f;
  // PE entry q: q<3 identity; else f=(q-3)/6, w=(q-3)%6: w<3 sin(2^f x_w) else cos(2^f x_{w-3})
  auto accum = [&](int q, float g, bool active) {
    if (q < 0 || q >= kPeDim) return;
    float coef = 1.0f;
    int c = q;
    if (q >= 3) {
      const int f = (q - 3) / 6, w = (q - 3) % 6;
      const float sc = (float)(1 << f);
      c = w < 3 ? w : w - 3;
      coef = w < 3 ? sc * pe.v[q + 3] : -sc * pe.v[q - 3];
    }
    const float term = active ? coef * g : 0.0f;
    if (c == 0) dx0 += term; else if (c == 1) dx1 += term; else dx2 += term;
  };
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    accum(rho(r), gpe0[r], half == 0);
    accum(rho(r) + 4, gpe0[r], half == 1);
    accum(rho(r) >= 25 ? 32 + rho(r) - 25 : -1, gpe1[r], half == 0);
    accum(rho(r) + 4 >= 25 ? 32 + rho(r) + 4 - 25 : -1, gpe1[r], half == 1);
  }
  dx0 += __shfl_xor(dx0, 32); dx1 += __shfl_xor(dx1, 32); dx2 += __shfl_xor(dx2, 32);

  bool clamped = false;
  if (a.sphere_radius > 0.0f && p < a.clamp_n) {
    const float nrm = __builtin_sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
    const float sphere = a.sphere_scale * (a.sphere_radius - nrm);
    if (sphere < sdf) {
      sdf = sphere;
      const float k = -a.sphere_scale / nrm;
      dx0 = k * x0; dx1 = k * x1; dx2 = k * x2;
      clamped = true;
    }
  }
  if (half == 0 && p < a.src.P && a.clamp_mask) a.clamp_mask[p] = clamped ? 1 : 0;
  if (half == 0 && p < a.src.P) {
    a.sdf[p] = sdf;
    a.grad[3 * p + 0] = dx0; a.grad[3 * p + 1] = dx1; a.grad[3 * p + 2] = dx2;
  }
}

// pair-block layout of the fp16x2 kernels (svs_blocks_h2.h) -> row-major (P, 256): value = hi + mid
__global__ void pair_tiles_to_rows_kernel(const float* __restrict__ tiles, int P, float* __restrict__ rows) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over P*256
  if (idx >= (size_t)P * 256) return;
  const int p = (int)(idx / 256), f = (int)(idx % 256);
  const int wtile = p / 32, col = p % 32;
  // row f = 16 s + 8 (j >> 2) + 4 half + (j & 3)
  const int s = f >> 4, w = f & 15, half = (w >> 2) & 1, j = (w & 3) + 4 * (w >> 3), lane = col + 32 * half;
  const _Float16* blk = reinterpret_cast<const _Float16*>(tiles + (size_t)wtile * 128 * 64);
  const int b = lane >> 5, a = (lane >> 2) & 7, q = lane & 3;
  const int slot = 16 * (a >> 1) + 8 * ((a & 1) ^ (s & 1)) + 4 * b + q;        // piece_slot() of svs_blocks_h2.h
  const size_t e = ((size_t)s * 64 + slot) * 8 + j;
  rows[idx] = (float)blk[e] + (float)blk[e + 1024 * 8];
}

// wave-tile layout -> row-major (P, 256)
__global__ void tiles_to_rows_kernel(const float* __restrict__ tiles, int P, float* __restrict__ rows) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over P*256
  if (idx >= (size_t)P * 256) return;
  const int p = (int)(idx / 256), f = (int)(idx % 256);
  const int wtile = p / 32, col = p % 32;
  const int t = f / 32, local = f % 32;
  const int half = (local >> 2) & 1, r = (local & 3) + 4 * (local >> 3);
  const int i = 16 * t + r, lane = col + 32 * half;
  rows[idx] = tiles[(size_t)wtile * 128 * 64 + ((size_t)(i / 4) * 64 + lane) * 4 + (i % 4)];
}

// ------------------------------------------------------------------------------------------------------
// RenderingNetwork.forward, mode 'idr' (network.py:170-190): cat[x, PE1(view), normal, feature] -> 4 x 256 ReLU -> 3, sigmoid
// ------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(kThreads, 1) void rgb_kernel(RgbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  RgbStream st;
  st.g = a.stream; st.buf = reinterpret_cast<f32x4*>(smem); st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
  const int pc = p < a.src.P ? p : a.src.P - 1;

  st.prefetch<kRgbChunk0F4>();
  float x0, x1, x2;
  load_point(a.src, p, x0, x1, x2);
  const float* vd = a.view + 3 * (size_t)(a.view_S > 0 ? pc / a.view_S : pc);
  const float d0 = vd[0], d1 = vd[1], d2 = vd[2];
  // extra rows: [x(3), d(3), sin d(3), cos d(3), n(3), 0]
  float ex[16];
  ex[0] = x0; ex[1] = x1; ex[2] = x2; ex[3] = d0; ex[4] = d1; ex[5] = d2;
  ex[6] = sinf(d0); ex[7] = sinf(d1); ex[8] = sinf(d2); ex[9] = cosf(d0); ex[10] = cosf(d1); ex[11] = cosf(d2);
  ex[12] = a.normals[3 * pc]; ex[13] = a.normals[3 * pc + 1]; ex[14] = a.normals[3 * pc + 2]; ex[15] = 0.0f;
  float eb[8];   // B operands of k-steps 128..135: rows rho(r) / rho(r)+4 of the 16 extra rows
#pragma unroll
  for (int r = 0; r < 8; ++r) eb[r] = half ? ex[rho(r) + 4] : ex[rho(r)];

  f32x16 x[8], y[8];
  load_tile_regs(a.feat_tiles + (size_t)wtile * 128 * 64, x, lane);
  // rbuf: [block 0..3][wave tile][128*64] then the extras [wave tile][1024] (same [block][tile] layout as the SDF buffers)
  const size_t LS = block_stride();
  float* rb = a.rbuf ? a.rbuf + (size_t)wtile * kBlockF : nullptr;
  if (rb) {
    f32x4* d = reinterpret_cast<f32x4*>(a.rbuf + 4 * LS + (size_t)wtile * 1024) + lane;
    f32x4 v0, v1; v0[0] = eb[0]; v0[1] = eb[1]; v0[2] = eb[2]; v0[3] = eb[3]; v1[0] = eb[4]; v1[1] = eb[5]; v1[2] = eb[6]; v1[3] = eb[7];
    d[0] = v0; d[64] = v1; d[128] = (f32x4)(0.0f); d[192] = (f32x4)(0.0f);
  }
  st.advance();

  // ---- layer 0: 271 -> 256
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (rb && t > 0) store_tile(rb, t - 1, lane, y[t - 1]);     // deferred store (see forward_trunk)
    if (t < 7) st.prefetch<kRgbChunk0F4>(); else st.prefetch<kChunkF4>();
    const f32x4* chunk = st.cur_buf();
    f32x16 acc = tile_mma<128>(chunk, x, lane);
    const f32x4* a_ptr = chunk + kHdrF4 + 2048 + lane;
#pragma unroll
    for (int s4 = 0; s4 < 2; ++s4) {
      const f32x4 w = a_ptr[s4 * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], eb[4 * s4 + j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) y[t][r] = __builtin_fmaxf(acc[r], 0.0f);
    st.advance();
  }
  if (rb) store_tile(rb, 7, lane, y[7]);
  // ---- layers 1..3
  for (int l = 1; l < 4; ++l) {
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = y[t];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (rb && t > 0) store_tile(rb + (size_t)l * LS, t - 1, lane, y[t - 1]);
      st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma<128>(st.cur_buf(), x, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) y[t][r] = __builtin_fmaxf(acc[r], 0.0f);
      st.advance();
    }
    if (rb) store_tile(rb + (size_t)l * LS, 7, lane, y[7]);
  }
  // ---- layer 4: 256 -> 3 as one tile (rows 0..2 live in registers 0..2 of lanes 0..31), sigmoid
  const f32x16 acc = tile_mma<128>(st.cur_buf(), y, lane);
  if (half == 0 && p < a.src.P) {
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb[3 * p + c] = 1.0f / (1.0f + __expf(-acc[c]));
  }
}

}  // namespace mlp
}  // namespace svs

using namespace svs;
using namespace svs::mlp;


extern "C" {

int svs_sdf_vals(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                 int S, int n_rays, const float* stream, int precision, float sphere_radius, float sphere_scale,
                 int clamp_n, float* sdf, const int* gate, int gate_points, int gate_stride, void* hip_stream) {
  SdfOnlyArgs a;
  if (int rc = fill_src(a.src, points, n_points, cam, cam_stride, dirs, z, S, n_rays, "svs_sdf_vals")) return rc;
  if (!stream || !sdf) { set_error("svs_sdf_vals: null stream/sdf"); return SVS_EINVAL; }
  a.stream = reinterpret_cast<const f32x4*>(stream); a.sdf = sdf;
  a.sphere_radius = sphere_radius; a.sphere_scale = sphere_scale; a.gate = gate;
  a.gate_points = gate_points > 0 ? gate_points : 0x7fffff80; a.gate_stride = gate_stride;
  if (gate && a.gate_points % kWgPts) { set_error("svs_sdf_vals: gate_points must be a multiple of %d", kWgPts); return SVS_EINVAL; }
  a.clamp_n = clamp_n < 0 ? a.src.P : clamp_n;
  if (is_h2(precision)) return launch_sdf_only_h2(a, (hipStream_t)hip_stream);
  if (precision != kFmtF32) { set_error("svs_sdf_vals: unknown precision %d", precision); return SVS_EINVAL; }
  static int once = set_lds(sdf_only_kernel, kLdsBytes, "svs_sdf_vals");
  if (once) return once;
  sdf_only_kernel<<<(a.src.P + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_sdf_vals");
}

#ifdef SVS_EXPERIMENTAL_KERNELS
// svs_sdf_vals by the K-split-pair kernel (two waves per SIMD on the same 32 points, csrc/svs_mlp_h2p.hip): same stream
// (fp16x2), same arguments.  An experiment kept for A/B runs; measured slower than svs_sdf_vals.
int svs_sdf_vals_pair(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                      int S, int n_rays, const float* stream, float sphere_radius, float sphere_scale, int clamp_n, float* sdf,
                      const int* gate, int gate_points, int gate_stride, void* hip_stream) {
  SdfOnlyArgs a;
  if (int rc = fill_src(a.src, points, n_points, cam, cam_stride, dirs, z, S, n_rays, "svs_sdf_vals_pair")) return rc;
  if (!stream || !sdf) { set_error("svs_sdf_vals_pair: null stream/sdf"); return SVS_EINVAL; }
  a.stream = reinterpret_cast<const f32x4*>(stream); a.sdf = sdf;
  a.sphere_radius = sphere_radius; a.sphere_scale = sphere_scale; a.gate = gate;
  a.gate_points = gate_points > 0 ? gate_points : 0x7fffff80; a.gate_stride = gate_stride;
  if (gate && a.gate_points % kWgPts) { set_error("svs_sdf_vals_pair: gate_points must be a multiple of %d", kWgPts); return SVS_EINVAL; }
  a.clamp_n = clamp_n < 0 ? a.src.P : clamp_n;
  return launch_sdf_only_kp(a, (hipStream_t)hip_stream);
}
#endif

size_t svs_sdf_hbuf_bytes(int n_points) { return (size_t)wave_tiles(n_points) * 8 * 128 * 64 * sizeof(float); }
size_t svs_feat_tiles_bytes(int n_points) { return (size_t)wave_tiles(n_points) * 128 * 64 * sizeof(float); }
size_t svs_rgb_rbuf_bytes(int n_points) { return (size_t)wave_tiles(n_points) * kRbufF * sizeof(float); }

int svs_sdf_outputs(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs,
                    const float* z, int S, int n_rays, const float* stream, int precision, float sphere_radius,
                    float sphere_scale, int clamp_n, float* sdf, float* grad, float* feat_tiles, float* hbuf,
                    float* gbuf, unsigned char* clamp_mask, void* hip_stream) {
  SdfFullArgs a;
  if (int rc = fill_src(a.src, points, n_points, cam, cam_stride, dirs, z, S, n_rays, "svs_sdf_outputs")) return rc;
  if (!stream || !sdf || !grad || !hbuf) { set_error("svs_sdf_outputs: null stream/sdf/grad/hbuf"); return SVS_EINVAL; }
  a.stream = reinterpret_cast<const f32x4*>(stream); a.sdf = sdf; a.grad = grad; a.feat_tiles = feat_tiles; a.hbuf = hbuf;
  a.gbuf = gbuf; a.clamp_mask = clamp_mask;
  a.sphere_radius = sphere_radius; a.sphere_scale = sphere_scale;
  a.clamp_n = clamp_n < 0 ? a.src.P : clamp_n;
  if (is_h2(precision)) return launch_sdf_full_h2(a, precision == kFmtF16x2, (hipStream_t)hip_stream);
  if (precision != kFmtF32) { set_error("svs_sdf_outputs: unknown precision %d", precision); return SVS_EINVAL; }
  static int once = set_lds(sdf_full_kernel, kLdsBytes, "svs_sdf_outputs");
  if (once) return once;
  sdf_full_kernel<<<(a.src.P + kWgPts - 1) / kWgPts, kThreads, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_sdf_outputs");
}

int svs_tiles_to_rows(const float* tiles, int n_points, int precision, float* rows, void* hip_stream) {
  if (!tiles || !rows || n_points <= 0) { set_error("svs_tiles_to_rows: bad argument"); return SVS_EINVAL; }
  if (is_h2(precision)) {
    const size_t n2 = (size_t)n_points * 256;
    pair_tiles_to_rows_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, (hipStream_t)hip_stream>>>(tiles, n_points, rows);
    return check_launch("svs_tiles_to_rows");
  }
  const size_t n = (size_t)n_points * 256;
  tiles_to_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)hip_stream>>>(tiles, n_points, rows);
  return check_launch("svs_tiles_to_rows");
}

int svs_rgb_eval(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                 int S, int n_rays, const float* normals, const float* view_dirs, int view_S, const float* feat_tiles,
                 const float* stream, int precision, float* rgb, float* rbuf, void* hip_stream) {
  RgbArgs a;
  if (int rc = fill_src(a.src, points, n_points, cam, cam_stride, dirs, z, S, n_rays, "svs_rgb_eval")) return rc;
  if (!normals || !view_dirs || !feat_tiles || !stream || !rgb || view_S < 0 || (view_S > 0 && a.src.P % view_S)) {
    set_error("svs_rgb_eval: bad argument"); return SVS_EINVAL;
  }
  a.normals = normals; a.view = view_dirs; a.view_S = view_S; a.feat_tiles = feat_tiles;
  a.stream = reinterpret_cast<const f32x4*>(stream); a.rgb = rgb; a.rbuf = rbuf;
  if (is_h2(precision)) return launch_rgb_h2(a, precision == kFmtF16x2, (hipStream_t)hip_stream);
  if (precision != kFmtF32) { set_error("svs_rgb_eval: unknown precision %d", precision); return SVS_EINVAL; }
  constexpr int lds = 2 * kRgbBufF4 * 16;
  static int once = set_lds(rgb_kernel, lds, "svs_rgb_eval");
  if (once) return once;
  rgb_kernel<<<(a.src.P + kWgPts - 1) / kWgPts, kThreads, lds, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_rgb_eval");
}

}  // extern "C"
