// fp16x2 variants of the fused SDF ("implicit") and radiance ("rendering") MLP kernels of svs_mlp.hip: the same
// networks, the same transposed register-resident activations and LDS-DMA weight stream, with every layer product
// evaluated by v_mfma_f32_32x32x16_f16 on two-piece fp16 operands (svs_mlp_h2_dev.h) and the previous tile's
// epilogue (activation, operand split, stores) interleaved with the current tile's MFMAs.
//
// Reference semantics: volsdf/model/network.py:71-131 (ImplicitNetwork), :170-190 (RenderingNetwork).
#include "svs_mlp_h2_dev.h"
#include "svs_mlp_host.h"
#include "svs_mlp_args.h"
#include "svs_mlp_h2_trunk.h"
#include "svs_blocks_h2.h"

namespace svs {
namespace mlp {

// ImplicitNetwork.get_sdf_vals (network.py:125-131), no grad: the sampler's evaluation.
__global__ __launch_bounds__(kThreads, 1) void sdf_only_h2_kernel(SdfOnlyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.gate && a.gate[(size_t)(blockIdx.x * kWgPts / a.gate_points) * a.gate_stride] == 0) return;
  Stream st;
  st.g = a.stream;
  st.buf = reinterpret_cast<f32x4*>(smem);
  st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int p = (blockIdx.x * kWaves + wave) * kTilePts + (lane & 31);
#if SVS_ABL & 16   // diagnostic: cycle stamps of wave 0 replace the first outputs of the workgroup
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif

  st.prefetch<kChunk0F4>();   // chunk 0 -> buffer 0 (overlaps the positional encoding below)
  float x0, x1, x2;
  load_point(a.src, p, x0, x1, x2);
  const float r2 = x0 * x0 + x1 * x1 + x2 * x2;   // for the sphere clamp at the end: one live value instead of three
  PosEnc pe;
  pe.compute(x0, x1, x2);
#if SVS_ABL & 16
  asm volatile("" : "+v"(pe.v[38]));
  const uint64_t c1 = __builtin_amdgcn_s_memtime();
#endif

  Pieces2 x, xn;
  f32x16 y8[8];
  forward_trunk_h2<false>(st, x, xn, y8, pe, lane, half, nullptr);
#if SVS_ABL & 16
  asm volatile("" : "+v"(y8[7][15]));
  const uint64_t c2 = __builtin_amdgcn_s_memtime();
#endif
  // the VEC chunk was prefetched by the last tile of layer 7
  float sdf = sdf_head(st.cur_buf(), y8, lane);
  if (a.sphere_radius > 0.0f && p < a.clamp_n) {
    const float nrm = __builtin_sqrtf(r2);
    sdf = __builtin_fminf(sdf, a.sphere_scale * (a.sphere_radius - nrm));
  }
  if (half == 0 && p < a.src.P) a.sdf[p] = sdf;
#if SVS_ABL & 16
  asm volatile("" : "+v"(sdf));
  const uint64_t c3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = a.sdf + (size_t)blockIdx.x * kWgPts;
    o[0] = (float)(c1 - c0); o[1] = (float)(c2 - c1); o[2] = (float)(c3 - c2); o[3] = (float)(c3 - c0);
    o[4] = (float)(r3 - r0); o[5] = (float)(r0 & 0xffffff);
  }
#endif
}

// --------------------------------------------------------------------------------------------------------------
// ImplicitNetwork.get_outputs (network.py:105-123): sdf, feature vector and d sdf / d x in one launch (see
// sdf_full_kernel in svs_mlp.hip for the algebra of the gradient pass).
// --------------------------------------------------------------------------------------------------------------
// Epilogue of one tile of the gradient pass: ghat_{l-1} = g(a_{l-1}) = g(h_l) * softplus'(a_{l-1}) (softplus' from the
// stored h_l tile `h`) -> gbuf (training), split into the next operand.
// GBUF: training launch, ghat blocks are stored (compile-time: a run-time test put exec-mask branches into every tile and
// made hipcc's vmcnt bookkeeping take the minimum over both paths); GP: as both pieces (svs_blocks_h2.h), else the hi piece
template <bool GBUF, bool GP>
struct RevEpi {
  f32x16 prev;
  TilePieces h;       // the stored h_l tile (pair block)
  float d;            // softplus' of the current slice
  float v8[8];
  Pieces2* out;
  float* gblk;        // ghat_{l-1} block of gbuf (stored unscaled, format GP); GBUF only
  float gm;           // GBUF: running max |ghat_{l-1}| of this lane's rows -> the block's record (pass B's scale floor)
  int lane, half;
  bool l4;            // l == 4: rows >= 217 of h_4 are the PE splice, they do not flow into lin3

  // three slices, one per MFMA gap of a k-step: exp2 | 1 - e and the product | masks, split
  __device__ __forceinline__ void a(int r) {
#if SVS_ABL & 512      // diagnostic: no softplus' arithmetic in the reverse epilogue
    d = 0.5f;
#else
    d = __builtin_amdgcn_exp2f(grad_times<true>(h, r, -100.0f * 1.44269504088896341f));
#endif
    pin(d);
  }
  __device__ __forceinline__ void a2(int r) {
    d = prev[r] * (1.0f - d);
    pin(d);
  }
  template <bool DEFER = false>     // DEFER: the split of elements 8..15 is left to finish(tp) (see TrunkEpi::emit)
  __device__ __forceinline__ void b(int tp, int r) {
    float v = d;
    if (l4 && tp == 7) v = 0.0f;
    if (l4 && tp == 6) {
      const bool z0 = rho(r) >= 25, z1 = rho(r) + 4 >= 25;
      if (z0 || z1) { if (half ? z1 : z0) v = 0.0f; }
    }
    pin(v);
    if (GBUF) { gm = __builtin_fmaxf(gm, __builtin_fabsf(v)); pin(gm); }   // (pinned: LLVM re-associates the chain into a tree and keeps 16 values alive)
    v8[r & 7] = v;
    if ((r & 7) == 7 && !(DEFER && r == 15)) {
      const int k = 2 * tp + (r >> 3);
      split8(v8, out->h[k], out->m[k]);
      pin(out->h[k], out->m[k]);
      if (GBUF && !DEFER) store_grad<GP>(gblk, k, lane, out->h[k], out->m[k]);
    }
  }
  template <bool STORE = false>
  __device__ __forceinline__ void finish(int tp) {
    const int k = 2 * tp + 1;
    split8(v8, out->h[k], out->m[k]);
    pin(out->h[k], out->m[k]);
    if (GBUF && STORE) store_grad<GP>(gblk, k, lane, out->h[k], out->m[k]);
  }
  // the gbuf stores issued during tile t (see TrunkEpi::late_store): the pieces of k-steps 2(t-1) and 2(t-2)+1
  __device__ __forceinline__ void st(int t, int s) {
    if (!GBUF || (SVS_ABL & 256)) return;         // (256: diagnostic, no gbuf stores)
    if (s == 9) store_piece(gblk, 2 * (t - 1), lane, out->h[2 * (t - 1)], 0);
    if (t >= 2 && s == 11) store_piece(gblk, 2 * (t - 2) + 1, lane, out->h[2 * (t - 2) + 1], 0);
    if (GP && s == 13) store_piece(gblk, 2 * (t - 1), lane, out->m[2 * (t - 1)], 1);
    if (GP && t >= 2 && s == 14) store_piece(gblk, 2 * (t - 2) + 1, lane, out->m[2 * (t - 2) + 1], 1);
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { a(r); a2(r); b(tp, r); }
  }
};

// reverse of trunk layer l (7..1): in = g(a_l) pieces, out = g(a_{l-1}) pieces
template <bool GBUF, bool GP>
__device__ __forceinline__ void reverse_layer_h2(Stream& st, const Pieces2& in, Pieces2& out, int l, const float* hb,
                                                 float* gb, float* grec0, size_t rec_stride, f32x16& skip6, f32x16& skip7, int lane,
                                                 int half) {
  RevEpi<GBUF, GP> ep;
  ep.out = &out; ep.lane = lane; ep.half = half; ep.l4 = l == 4; ep.gm = 0.0f;
  const size_t LS = block_stride();
  ep.gblk = GBUF ? gb + (size_t)(l - 1) * LS : nullptr;
  const float* hblk = hb + (size_t)(l - 1) * LS;
  // h tile t + 1 (a pair: 4 fragments) is requested during tile t behind k-steps 10, 12, 14, 15 -- after the last LDS-DMA
  // piece (k-step 8), like the gbuf stores: in flight across the tile's barrier, complete one tile later, consumed by the
  // epilogue of tile t + 1 during tile t + 2.  (Requested in front of the tile's pieces, the barrier's vmcnt wait had to
  // sit out their HBM latency: 700 cycles per tile.)
  TilePieces hnext;
  load_tile_pair(hblk, 0, lane, hnext);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    // the h tile this tile's epilogue slices consume (ep.h, requested two tiles ago) is waited for HERE, before the tile's
    // LDS-DMA pieces are issued: hipcc does not count those (svs_mlp_dev.h), so a wait placed among them would also
    // wait for the pieces just issued
    if (t >= 1) pin(ep.h.h[0], ep.h.h[1]), pin(ep.h.m[0], ep.h.m[1]);
    const TilePieces hcur = hnext;
    auto hload = [&](int s) {
      if (t == 7 || (SVS_ABL & 128)) return;      // (128: diagnostic, no h loads)
      if (s == 10) hnext.h[0] = load_piece(hblk, 2 * (t + 1), lane, 0);
      if (s == 12) hnext.h[1] = load_piece(hblk, 2 * (t + 1) + 1, lane, 0);
      if (s == 14) hnext.m[0] = load_piece(hblk, 2 * (t + 1), lane, 1);
      if (s == 15) hnext.m[1] = load_piece(hblk, 2 * (t + 1) + 1, lane, 1);
    };
    // the next reverse chunk (the last one: REV0 tile 0) is fetched in pieces behind k-steps 0..8
    f32x16 acc;
    if (t == 0) acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, NoEpi(), NoEpi(), hload);
    else acc = tile_mma_h2_pf<16, kChunkF4>(st, in, lane, [&](int s) { ep.a(s); }, [&](int s) { ep.a2(s); },
                                            [&](int s) { ep.template b<true>(t - 1, s); ep.st(t, s); hload(s); },
                                            [&]() { if (t >= 2) ep.finish(t - 2); });
    if (l == 4 && t == 7) skip7 = acc;
    if (l == 4 && t == 6) skip6 = acc;
    ep.prev = acc;
    ep.h = hcur;
    // in flight across the barrier: the 4 h loads (t < 7) and the gbuf stores (training: 1 in tile 1, then 2; twice that
    // with GP) of this tile
    constexpr int per = GP ? 2 : 1;
    const int stores = !GBUF || t == 0 ? 0 : (t == 1 ? per : 2 * per);
    if (t < 7) {
      if (stores == 0) st.advance_keep<4>(); else if (stores == 1) st.advance_keep<5>(); else if (stores == 2) st.advance_keep<6>();
      else st.advance_keep<8>();
    } else {
      if (stores == 0) st.advance(); else if (stores == 2) st.advance_keep<2>(); else st.advance_keep<4>();
    }
  }
  ep.template finish<true>(6);
  ep.all(7);
  // the record of ghat_{l-1}: scale 1 (the block is stored unscaled), max |ghat_{l-1}| of the point
  if (GBUF) store_record(grec0 + (size_t)(l - 1) * rec_stride, lane, 1.0f, __builtin_fmaxf(ep.gm, __shfl_xor(ep.gm, 32)));
}

template <bool GBUF, bool GP>
__global__ __launch_bounds__(kThreads, 1) void sdf_full_h2_kernel(SdfFullArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stream st;
  st.g = a.stream;
  st.buf = reinterpret_cast<f32x4*>(smem);
  st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
  const int wtile = blockIdx.x * kWaves + wave;
  const int p = wtile * kTilePts + (lane & 31);
#if SVS_ABL & 16   // diagnostic: cycle stamps of wave 0 replace the workgroup's first gradient outputs
  uint64_t cs[8];
  cs[0] = __builtin_amdgcn_s_memtime();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
#define SVS_STAMP(i, var) { asm volatile("" : "+v"(var)); cs[i] = __builtin_amdgcn_s_memtime(); }
#else
#define SVS_STAMP(i, var)
#endif

  st.prefetch<kChunk0F4>();
  float x0, x1, x2;
  load_point(a.src, p, x0, x1, x2);
  PosEnc pe;
  pe.compute(x0, x1, x2);
  SVS_STAMP(1, pe.v[38])

  float* hb = a.hbuf + (size_t)wtile * kBlockF;                        // block l of this tile: + l * block_stride()
  float* gb = GBUF ? a.gbuf + (size_t)wtile * kBlockF : nullptr;
  // gbuf's records (behind its slots, like a scaled buffer's: svs_sdf_gbuf_bytes): [1.0][max_r |ghat_l| of the point] per block
  // and tile -- with the factor pass A leaves in u's records, pass B's bound of the a2 it re-forms (svs_mlp_bwd_h2.hip)
  const size_t rec_stride = (size_t)gridDim.x * kWaves * kRecF;
  float* grec0 = GBUF ? record_ptr(a.gbuf, 8, (size_t)gridDim.x * kWaves, 0, wtile) : nullptr;
  Pieces2 x, xn;
  float sdf;
  {
    f32x16 y8[8];
    forward_trunk_h2<true>(st, x, xn, y8, pe, lane, half, hb);
    SVS_STAMP(2, y8[7][15])
    // ---- head: current chunk = VEC (W8 row 0 in C-layout order as float32, b8[0])
    st.prefetch<kChunkF4>();                       // FEAT tile 0
    sdf = sdf_head(st.cur_buf(), y8, lane);
    // x = h_8 (input of the feature head, split by the trunk), xn = g(a_7) = W8[0,:] * softplus'(a_7): the VEC chunk's
    // buffer is overwritten two prefetches from now, so its weights are consumed here
    const f32x4* w_ptr = st.cur_buf() + kHdrF4 + lane;
    float gm7 = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x16 g;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = w_ptr[(4 * t + q) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g[4 * q + j] = w[j] * dsoftplus_from_h(y8[t][4 * q + j]);
          if (GBUF) gm7 = __builtin_fmaxf(gm7, __builtin_fabsf(g[4 * q + j]));
        }
      }
      // ghat_7 = W8[0,:] * softplus'(a_7), stored unscaled.  (x already holds the pieces of h_8: the trunk's last layer
      // split and stored them.)
      split_tile(g, t, xn);
      if (GBUF) {
        store_grad<GP>(gb + 7 * block_stride(), 2 * t, lane, xn.h[2 * t], xn.m[2 * t]);
        store_grad<GP>(gb + 7 * block_stride(), 2 * t + 1, lane, xn.h[2 * t + 1], xn.m[2 * t + 1]);
      }
    }
    if (GBUF) store_record(grec0 + 7 * rec_stride, lane, 1.0f, __builtin_fmaxf(gm7, __shfl_xor(gm7, 32)));
    st.advance();
  }
  SVS_STAMP(3, sdf)
  // ---- feature vector = rows 1..256 of lin8 (no activation), stored as a PAIR block (the radiance network's operand
  // and the B operand of its first weight gradient); tile t-1 is split and stored while tile t's MFMAs run
  float* ft = a.feat_tiles ? a.feat_tiles + (size_t)wtile * kBlockF : nullptr;
  {
    f32x16 prev;
    float v8[8];
    f16x8 fh[2], fm[2];
    auto slice = [&](int tp, int r) {             // stores behind the LDS-DMA pieces (k-steps 9, 11, 15, 15)
      v8[r & 7] = prev[r];
      if ((r & 7) == 7) { split8(v8, fh[r >> 3], fm[r >> 3]); pin(fh[r >> 3], fm[r >> 3]); }
      if (!ft) return;
      if (r == 9) store_piece(ft, 2 * tp, lane, fh[0], 0);
      if (r == 11) store_piece(ft, 2 * tp, lane, fm[0], 1);
      if (r == 15) { store_piece(ft, 2 * tp + 1, lane, fh[1], 0); store_piece(ft, 2 * tp + 1, lane, fm[1], 1); }
    };
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x16 acc;                                   // prefetching FEAT t+1, or reverse L7 tile 0
      if (t == 0) acc = tile_mma_h2_pf<16, kChunkF4>(st, x, lane, NoEpi(), NoEpi());
      else acc = tile_mma_h2_pf<16, kChunkF4>(st, x, lane, NoEpi(), [&](int s) { slice(t - 1, s); });
      prev = acc;
      if (ft && t > 0) st.advance_keep<4>();
      else st.advance();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) slice(7, r);
  }
  SVS_STAMP(4, sdf)
  // ---- reverse layers 7..1, operands ping-pong between xn and x
  f32x16 skip7 = (f32x16)(0.0f);   // g(PE[0..31]) from the skip connection (tile 7 of g(h_4 spliced))
  f32x16 skip6 = (f32x16)(0.0f);   // tile 6; only local rows 25..31 are PE[32..38]
  for (int l = 7; l >= 1; l -= 2) {
    reverse_layer_h2<GBUF, GP>(st, xn, x, l, hb, gb, grec0, rec_stride, skip6, skip7, lane, half);
    if (l > 1) reverse_layer_h2<GBUF, GP>(st, x, xn, l - 1, hb, gb, grec0, rec_stride, skip6, skip7, lane, half);
  }
  SVS_STAMP(5, skip7[0])
  // ---- reverse layer 0: g(PE) = W0^T g(a_0) (+ skip), 2 tiles; g(a_0) is in x
  st.prefetch<kChunkF4>();
  f32x16 gpe0 = tile_mma_h2<16>(st.cur_buf(), x, lane);
  st.advance();
  f32x16 gpe1 = tile_mma_h2<16>(st.cur_buf(), x, lane);
  gpe0 += skip7;
  gpe1 += skip6;

  // ---- d/dx of the positional encoding
  if (!GBUF) {
    // Render instance: the 39 encoding values are formed AGAIN here instead of being held through the reverse sweep (the
    // trunk consumed them at the skip splice; recomputing costs ~1.5 k instructions per wave, holding them made this instance
    // spill 136 bytes per lane: rounds 3-5; now 0).  The point index passes through an empty asm so that the two evaluations
    // are not merged back into one long-lived set of registers; same inputs, same routines: same bits.
    int p2 = p;
    asm volatile("" : "+v"(p2));
    load_point(a.src, p2, x0, x1, x2);          // (the point itself too: three more registers through the sweep otherwise)
    pe.compute(x0, x1, x2);
  }
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
#if SVS_ABL & 16
  SVS_STAMP(6, dx0)
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  if (threadIdx.x == 0) {   // pe, trunk, head, features, reverse 7..1, reverse 0 + Jacobian, total, real time
    float* o = a.grad + (size_t)blockIdx.x * kWgPts * 3;
    for (int i = 0; i < 6; ++i) o[i] = (float)(cs[i + 1] - cs[i]);
    o[6] = (float)(cs[6] - cs[0]); o[7] = (float)(r1 - r0);
  }
#endif
}

// --------------------------------------------------------------------------------------------------------------
// RenderingNetwork.forward, mode 'idr' (network.py:170-190)
// --------------------------------------------------------------------------------------------------------------
template <bool GP>      // GP: r_l is stored with both pieces (the weight gradient's three-product B operand), else its hi plane
struct RgbEpi {
  f32x16 prev;
  float v8[8];
  Pieces2* out;
  float* rblk;       // this layer's block of rbuf (a pair block; GP = false: its hi plane only) or nullptr
  int lane;
  __device__ __forceinline__ void b(int tp, int r) {
    float v;
    asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(prev[r]));   // ReLU without the canonicalising v_max of fmaxf (IEEE mode)
    pin(v);
    v8[r & 7] = v;
    if ((r & 7) == 7) {
      split8(v8, out->h[2 * tp + (r >> 3)], out->m[2 * tp + (r >> 3)]);
      pin(out->h[2 * tp + (r >> 3)], out->m[2 * tp + (r >> 3)]);
    }
    // r_l is read back by the backward's ReLU mask (hi plane) and as the weight gradient's B operand (GP: both planes);
    // stored behind the LDS-DMA pieces (k-steps 0..9): k-steps 11 .. 15
    if (rblk) {
      if (r == 11) store_piece(rblk, 2 * tp, lane, out->h[2 * tp], 0);
      if (GP && r == 13) store_piece(rblk, 2 * tp, lane, out->m[2 * tp], 1);
      if (r == 15) store_grad<GP>(rblk, 2 * tp + 1, lane, out->h[2 * tp + 1], out->m[2 * tp + 1]);
    }
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) b(tp, r);
  }
};

// One radiance layer.  The next chunk (N16NEXT_LAST float4 behind the layer's last tile) is fetched in pieces behind the
// first k-steps of each tile (Stream::prefetch_step: 9 or 10 pieces, k-steps 0..9); the rbuf stores of tile t-1's
// epilogue are issued in k-steps 11, 15: younger than every piece, they may stay in flight.
template <int KS, int N16NEXT_LAST, bool GP>
__device__ __forceinline__ void rgb_layer_h2(RgbStream& st, const Pieces2& in, Pieces2& out, float* rblk, int lane) {
  RgbEpi<GP> ep;
  ep.out = &out; ep.rblk = rblk; ep.lane = lane;
  constexpr int kNext = KS == 17 ? kRgbChunk0F4 : kChunkF4;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    auto relu = [&](int s) { if (s < 16) ep.b(t - 1, s); };
    f32x16 acc;
    if (t == 0) acc = tile_mma_h2_pf<KS, kNext>(st, in, lane, NoEpi(), NoEpi());
    else if (t < 7) acc = tile_mma_h2_pf<KS, kNext>(st, in, lane, NoEpi(), relu);
    else acc = tile_mma_h2_pf<KS, N16NEXT_LAST>(st, in, lane, NoEpi(), relu);
    ep.prev = acc;
    if (rblk && t > 0) st.template advance_keep<GP ? 4 : 2>();
    else st.advance();
  }
  ep.all(7);
}

template <bool GP>
__global__ __launch_bounds__(kThreads, 1) void rgb_h2_kernel(RgbArgs a) {
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
  float eb[8];   // fragment of k-step 16: rows rho(j) / rho(j)+4 of the 16 extra rows
#pragma unroll
  for (int r = 0; r < 8; ++r) eb[r] = half ? ex[rho(r) + 4] : ex[rho(r)];

  Pieces2 x, xn;
  split8(eb, x.h[16], x.m[16]);
  {
    const float* ft = a.feat_tiles + (size_t)wtile * kBlockF;   // a pair block: the operand as it is
#pragma unroll
    for (int k = 0; k < 16; ++k) { x.h[k] = load_piece(ft, k, lane, 0); x.m[k] = load_piece(ft, k, lane, 1); }
  }
  // rbuf: [block 0..3][wave tile][128*64] then the extras [wave tile][1024] (same [block][tile] layout as the SDF buffers)
  const size_t LS = block_stride();
  float* rb = a.rbuf ? a.rbuf + (size_t)wtile * kBlockF : nullptr;
  if (rb) {
    f32x4* d = reinterpret_cast<f32x4*>(a.rbuf + 4 * LS + (size_t)wtile * 1024) + lane;
    f32x4 v0, v1; v0[0] = eb[0]; v0[1] = eb[1]; v0[2] = eb[2]; v0[3] = eb[3]; v1[0] = eb[4]; v1[1] = eb[5]; v1[2] = eb[6]; v1[3] = eb[7];
    d[0] = v0; d[64] = v1; d[128] = (f32x4)(0.0f); d[192] = (f32x4)(0.0f);
  }
  st.advance();

  // ---- layer 0: 271 -> 256; layers 1..3; every output r_l (post-ReLU) optionally kept in rbuf
  rgb_layer_h2<17, kChunkF4, GP>(st, x, xn, rb, lane);
  rgb_layer_h2<16, kChunkF4, GP>(st, xn, x, rb ? rb + 1 * LS : nullptr, lane);
  rgb_layer_h2<16, kChunkF4, GP>(st, x, xn, rb ? rb + 2 * LS : nullptr, lane);
  rgb_layer_h2<16, kChunkF4, GP>(st, xn, x, rb ? rb + 3 * LS : nullptr, lane);   // prefetches lin4's chunk
  // ---- layer 4: 256 -> 3 as one tile (rows 0..2 live in registers 0..2 of lanes 0..31), sigmoid
  const f32x16 acc = tile_mma_h2<16>(st.cur_buf(), x, lane);
  if (half == 0 && p < a.src.P) {
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb[3 * p + c] = 1.0f / (1.0f + __expf(-acc[c]));
  }
}

}  // namespace mlp
}  // namespace svs

using namespace svs;
using namespace svs::mlp;

namespace svs {
namespace mlp {
// diagnostic builds with fewer registers must not become two workgroups per CU: pad the LDS request
constexpr int kLdsAbl = (SVS_ABL & 32) ? 0 : (SVS_ABL ? 20480 : 0);
int launch_sdf_only_h2(const SdfOnlyArgs& a, hipStream_t s) {
  static int once = set_lds(sdf_only_h2_kernel, kLdsBytes + kLdsAbl, "svs_sdf_vals");
  if (once) return once;
  sdf_only_h2_kernel<<<(a.src.P + kWgPts - 1) / kWgPts, kThreads, kLdsBytes + kLdsAbl, s>>>(a);
  return check_launch("svs_sdf_vals");
}
int launch_sdf_full_h2(const SdfFullArgs& a, bool grad_pair, hipStream_t s) {
  static int once = set_lds(sdf_full_h2_kernel<true, true>, kLdsBytes + kLdsAbl, "svs_sdf_outputs") |
                    set_lds(sdf_full_h2_kernel<true, false>, kLdsBytes + kLdsAbl, "svs_sdf_outputs") |
                    set_lds(sdf_full_h2_kernel<false, false>, kLdsBytes + kLdsAbl, "svs_sdf_outputs");
  if (once) return once;
  const int grid = (a.src.P + kWgPts - 1) / kWgPts;
  if (!a.gbuf) sdf_full_h2_kernel<false, false><<<grid, kThreads, kLdsBytes + kLdsAbl, s>>>(a);
  else if (grad_pair) sdf_full_h2_kernel<true, true><<<grid, kThreads, kLdsBytes + kLdsAbl, s>>>(a);     // training: ghat blocks stored
  else sdf_full_h2_kernel<true, false><<<grid, kThreads, kLdsBytes + kLdsAbl, s>>>(a);
  return check_launch("svs_sdf_outputs");
}
int launch_rgb_h2(const RgbArgs& a, bool grad_pair, hipStream_t s) {
  constexpr int lds = 2 * kRgbBufF4 * 16;
  static int once = set_lds(rgb_h2_kernel<true>, lds, "svs_rgb_eval") | set_lds(rgb_h2_kernel<false>, lds, "svs_rgb_eval");
  if (once) return once;
  const int grid = (a.src.P + kWgPts - 1) / kWgPts;
  if (grad_pair && a.rbuf) rgb_h2_kernel<true><<<grid, kThreads, lds, s>>>(a);
  else rgb_h2_kernel<false><<<grid, kThreads, lds, s>>>(a);
  return check_launch("svs_rgb_eval");
}
}  // namespace mlp
}  // namespace svs
