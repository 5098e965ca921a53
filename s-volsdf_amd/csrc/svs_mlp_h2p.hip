// K-split-pair variants of the fused fp16x2 MLP kernels (two waves per SIMD, svs_mlp_h2p_dev.h): same networks, same packed
// streams, same activation blocks, same results up to the order of the float32 accumulation.
// Reference semantics: volsdf/model/network.py:71-131 (ImplicitNetwork).
#include "svs_mlp_h2p_dev.h"
#include "svs_mlp_host.h"
#include "svs_mlp_args.h"
#include <type_traits>

namespace svs {
namespace mlp {
namespace kp {

// diagnostic build (-DKP_STAMP, never in the product): every wave accumulates the cycles it spends in the MFMA loops of the
// trunk tiles and in the send + wait + barrier that follows each, and writes them over its first outputs
#ifdef KP_STAMP
#define KP_T0() const uint64_t kp_t0 = __builtin_amdgcn_s_memtime()
#define KP_T1(acc) { const uint64_t kp_t1 = __builtin_amdgcn_s_memtime(); acc += (unsigned)(kp_t1 - kp_t0); }
__device__ unsigned kp_loop_cycles, kp_sync_cycles;
#else
#define KP_T0()
#define KP_T1(acc)
#endif

constexpr int kHeadXchg = 4 * 2 * 256;       // the sdf head's partial sums: [pair][role][64 lanes] floats

// Epilogue of one trunk tile for the 8 accumulator registers this wave owns (role 0: registers 0..7 = rows (j & 3) + 8 (j >> 2)
// + 4 half of the tile, role 1: the same + 16): softplus in three slices (exp2 | max, log2 | fma), the skip splice of layer 3,
// and the split into the wave's k-step `tp` of the next layer's operand -- or, LAST, the float32 copy y8.
template <bool LAST>
struct TrunkEpiP {
  float prev[8];
  SoftplusA sa;
  float v8[8];
  PiecesP* xn;
  float y8[LAST ? 64 : 1];   // LAST: [8 tiles][8], the wave's half of h_8 in float32
  const PosEnc* pe;
  int lane, half, role;
  bool splice;         // layer 3: rows >= 217 of the output are the PE splice (network.py:80-81)

  __device__ __forceinline__ void a(int j) {
    sa.lg = __builtin_amdgcn_exp2f(__builtin_fabsf(prev[j]) * (-100.0f * 1.44269504088896341f));
    pin(sa.lg);
  }
  __device__ __forceinline__ void a2(int j) {
    asm volatile("v_max_f32 %0, 0, %1" : "=v"(sa.mx) : "v"(prev[j]));
    sa.lg = __builtin_amdgcn_logf(1.0f + sa.lg);
    pin(sa.lg);
  }
  __device__ __forceinline__ void b(int tp, int j) {
    float v = softplus100_b(sa);
    if (tp == 6 && splice && j >= 4) {
      // tile 6, local rows 24 + (j & 3) + 4 half of role 1: rows >= 25 carry PE[32 + row - 25]
      const int q = j & 3;
      const float p0 = q >= 1 ? pe->v[32 + (q >= 1 ? q - 1 : 0)] : v;      // half 0: rows 24..27
      const float p1 = pe->v[35 + q];                                      // half 1: rows 28..31
      const float ps = half ? p1 : p0;
      v = role ? ps : v;
    }
    pin(v);
    v8[j] = v;
    if (LAST) y8[tp * 8 + j] = v;
  }
  __device__ __forceinline__ void finish(int tp) {
    if (!LAST) { split8(v8, xn->h[tp], xn->m[tp]); pin(xn->h[tp], xn->m[tp]); }
  }
  __device__ __forceinline__ void all(int tp) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { a(j); a2(j); b(tp, j); }
  }
  // layer 3, tile 7 = PE[0..31] (no MFMA): element j = PE[(j & 3) + 8 (j >> 2) + 16 role + 4 half]
  __device__ __forceinline__ void splice_tile7() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r0 = (j & 3) + 8 * (j >> 2);
      const float a0 = half ? pe->v[r0 + 4] : pe->v[r0];
      const float a1 = half ? pe->v[r0 + 20] : pe->v[r0 + 16];
      v8[j] = role ? a1 : a0;
    }
    finish(7);
  }
};

// one 256 -> 256 trunk layer (l >= 1).  On entry the layer's first chunk is current; on return the next layer's is.
template <bool LAST>
__device__ __forceinline__ void trunk_layer_p(StreamP& st, const PiecesP& x, TrunkEpiP<LAST>& ep, const Xchg& xc, int lane,
                                              unsigned& c_loop, unsigned& c_sync, Trace& tr) {
  float mine[8];
  const int n_tiles = ep.splice ? 7 : 8;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t == 7 && ep.splice) break;
    f32x4 r0, r1;
    if (t > 0) xc.recv((t - 1) & 1, lane, r0, r1);          // the partner's sums of tile t-1 (written before the last barrier)
    f32x16 acc;
    {
    KP_T0();
    if (t == 0) acc = tile_mma_p<kChunkF4>(st, x, lane, ep.role, NoE(), NoE(), NoE(), NoP(), tr);
    else acc = tile_mma_p<kChunkF4>(st, x, lane, ep.role, [&](int s) { ep.a(s); }, [&](int s) { ep.a2(s); },
                                    [&](int s) { ep.b(t - 1, s); },
                                    [&]() {
                                      if (t >= 2) ep.finish(t - 2);
#pragma unroll
                                      for (int j = 0; j < 4; ++j) { ep.prev[j] = mine[j] + r0[j]; ep.prev[4 + j] = mine[4 + j] + r1[j]; }
                                    }, tr);
    asm volatile("" : "+v"(acc));
    KP_T1(c_loop);
    }
    {
    KP_T0();
    own_half(acc, ep.role, mine);
    xc.send(acc, t & 1, lane);
    KP_MARK(tr);
    st.advance();
    KP_MARK(tr);
    KP_T1(c_sync);
    }
  }
  // the tail: the last tile's sums, its epilogue at once (tile indices are compile-time constants: a run-time index into the
  // piece arrays would push them into scratch)
  f32x4 r0, r1;
  xc.recv((n_tiles - 1) & 1, lane, r0, r1);
  auto tail = [&](auto last) {
    constexpr int T = decltype(last)::value;
    ep.finish(T - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { ep.prev[j] = mine[j] + r0[j]; ep.prev[4 + j] = mine[4 + j] + r1[j]; }
    ep.all(T);
    ep.finish(T);
  };
  if (ep.splice) { tail(std::integral_constant<int, 6>()); ep.splice_tile7(); }
  else tail(std::integral_constant<int, 7>());
}

// ImplicitNetwork.get_sdf_vals (network.py:125-131), no grad: the sampler's evaluation.
__global__ __launch_bounds__(kThreadsP, 1) void sdf_only_kp_kernel(SdfOnlyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.gate && a.gate[(size_t)(blockIdx.x * kWgPts / a.gate_points) * a.gate_stride] == 0) return;
  StreamP st;
  st.g = a.stream;
  st.buf = reinterpret_cast<f32x4*>(smem);
  st.cur = 1;
  const int lane = threadIdx.x & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int role = wave >> 2, pair = wave & 3;          // waves w and w + 4 share a SIMD and a point tile
  const int p = (blockIdx.x * 4 + pair) * kTilePts + (lane & 31);
  Xchg xc;
  xc.base = smem + kLdsBytes + pair * (2 * 2 * kXchgTile);
  xc.role = role;

  st.prefetch<kChunk0F4>();
  float x0, x1, x2;
  load_point(a.src, p, x0, x1, x2);
  const float r2 = x0 * x0 + x1 * x1 + x2 * x2;
  PosEnc pe;
  pe.compute(x0, x1, x2);

  PiecesP xa, xb;
  Trace tr;
  tr.on = false;
  const bool trace_wg = blockIdx.x == gridDim.x - 1;
#ifdef KP_TRACE
  tr.n = 0;
#endif
  unsigned c_loop = 0, c_sync = 0;
  {
    // ---- layer 0 : 39(48) -> 256.  K is three k-steps: both waves of a pair compute whole tiles and keep their half
    Pieces2 xpe;
    split_pe(pe, half, xpe);
    st.advance();
    TrunkEpiP<false> ep;
    ep.xn = &xa; ep.pe = &pe; ep.lane = lane; ep.half = half; ep.role = role; ep.splice = false;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t < 7) st.prefetch<kChunk0F4>(); else st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma_full<3, 0>(st, xpe, lane);
      own_half(acc, role, ep.prev);
      ep.all(t);
      ep.finish(t);
      st.advance();
    }
  }
  // ---- layers 1..6, operands ping-pong between xa and xb (layer 3 emits 217 rows + the skip splice)
  for (int l = 1; l < 7; l += 2) {
    // (one epilogue object per destination: a pointer that changes at run time would force the pieces into scratch)
    TrunkEpiP<false> ep;
    ep.pe = &pe; ep.lane = lane; ep.half = half; ep.role = role; ep.xn = &xb; ep.splice = l == 3;
    tr.on = trace_wg && l == 5; trunk_layer_p<false>(st, xa, ep, xc, lane, c_loop, c_sync, tr); tr.on = false;
    TrunkEpiP<false> ep2;
    ep2.pe = &pe; ep2.lane = lane; ep2.half = half; ep2.role = role; ep2.xn = &xa; ep2.splice = false;
    trunk_layer_p<false>(st, xb, ep2, xc, lane, c_loop, c_sync, tr);
  }
  // ---- layer 7: output kept in float32
  TrunkEpiP<true> ep7;
  ep7.xn = nullptr; ep7.pe = &pe; ep7.lane = lane; ep7.half = half; ep7.role = role; ep7.splice = false;
  trunk_layer_p<true>(st, xa, ep7, xc, lane, c_loop, c_sync, tr);
  const float* y8 = ep7.y8;
  // ---- head: current chunk = VEC (W8 row 0 in accumulator order as float32, b8[0]); this wave's 64 of the 256 terms
  float acc = 0.0f;
  {
    const f32x4* w_ptr = st.cur_buf() + kHdrF4 + lane;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q4 = 0; q4 < 2; ++q4) {
        const f32x4 w = w_ptr[(4 * t + 2 * role + q4) * 64];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_fmaf(w[c], y8[t * 8 + 4 * q4 + c], acc);
      }
  }
  acc += __shfl_xor(acc, 32);
  float* hx = reinterpret_cast<float*>(smem + kLdsBytes + kXchgBytes) + pair * 128;
  hx[role * 64 + lane] = acc;
  __syncthreads();
  float sdf = acc + hx[(role ^ 1) * 64 + lane] + st.cur_buf()[lane][0];
  if (a.sphere_radius > 0.0f && p < a.clamp_n) {
    const float nrm = __builtin_sqrtf(r2);
    sdf = __builtin_fminf(sdf, a.sphere_scale * (a.sphere_radius - nrm));
  }
  if (role == 0 && half == 0 && p < a.src.P) a.sdf[p] = sdf;
#ifdef KP_TRACE
  if (trace_wg && lane == 0) {
    for (int i = 0; i < 80; ++i) a.sdf[wave * 80 + i] = (float)(tr.t[i] - tr.t[0]);
    a.sdf[640 + wave] = (float)(tr.t[0] & 0xffffff);
  }
#endif
#ifdef KP_STAMP
  __syncthreads();
  if (lane == 0) { a.sdf[(size_t)blockIdx.x * kWgPts + 2 * wave] = (float)c_loop; a.sdf[(size_t)blockIdx.x * kWgPts + 2 * wave + 1] = (float)c_sync; }
#endif
}

}  // namespace kp

int launch_sdf_only_kp(const SdfOnlyArgs& a, hipStream_t s) {
  constexpr int lds = kp::kLdsP + kp::kHeadXchg;
  static int once = set_lds(kp::sdf_only_kp_kernel, lds, "svs_sdf_vals");
  if (once) return once;
  kp::sdf_only_kp_kernel<<<(a.src.P + kWgPts - 1) / kWgPts, kp::kThreadsP, lds, s>>>(a);
  return check_launch("svs_sdf_vals");
}

}  // namespace mlp
}  // namespace svs
