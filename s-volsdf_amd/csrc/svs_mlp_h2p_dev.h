// "K-split pairs": the fused fp16x2 MLP kernels at TWO waves per SIMD without giving up the 32-point MFMA shape.
//
// The 32-point kernels (svs_mlp_h2.hip) hold a wave's 256 x 32 layer input as 128 VGPRs of MFMA B fragments and the pieces
// being produced in another 128, which forces one wave per SIMD; every activation / split / store instruction is then issued
// between that wave's own MFMAs, in order, and the matrix core idles whenever a gap's vector work does not fit (0.57 busy
// forward-only, 0.39 with the reverse pass).  The 16-point experiment (svs_mlp_w16.hip) showed that
// shrinking the tile is not the way: v_mfma_f32_16x16x32_f16 holds the SIMD's vector issue port for 8 of its 16 cycles, and
// twice as many of them saturate that port together with the epilogue (same time as the 32-point kernel).
//
// Here the two waves of a SIMD (w and w + 4 of an 8-wave workgroup) work on the SAME 32 points and split the contraction:
// role 0 owns the even k-steps of every layer input (16 rows each), role 1 the odd ones -- 64 VGPRs of pieces instead of
// 128.  Both run v_mfma_f32_32x32x16_f16 over their 8 k-steps of every 32-row output tile; the two partial accumulators are
// summed through LDS: a wave sends the 8 accumulator registers the partner owns (role 0 keeps registers 0..7 = the next
// layer's k-step 2t, role 1 registers 8..15 = k-step 2t + 1 -- the accumulator layout IS the next operand's K order, as in
// the 32-point kernels) and receives the partner's 8 of its own.  Each wave therefore runs half of every tile's MFMAs and
// half of every tile's epilogue, symmetrically and without role branches; the matrix core sees 2 x 24 MFMAs of 32 cycles per
// tile = the same 1536 cycles, but one wave's softplus / split / stores issue while the other's MFMAs execute.
// Nothing changes outside the kernels: the packed weight streams (a wave reads every second k-step of a chunk), the
// activation blocks in HBM (a wave writes / reads the k-steps it owns) and the launch geometry (128 points per workgroup).
#pragma once
#include "svs_mlp_h2_dev.h"
#include "svs_blocks_h2.h"

namespace svs {
namespace mlp {
namespace kp {

// diagnostic build (-DKP_TRACE, never in the product): the waves of the LAST workgroup stamp s_memtime in front of every
// MFMA block and behind its vector block during one layer; the kernel writes the stamps over the first outputs
#ifdef KP_TRACE
struct Trace { unsigned t[80]; int n; bool on; };
__device__ __forceinline__ void trace_mark(Trace& tr) { if (tr.on && tr.n < 80) tr.t[tr.n++] = (unsigned)__builtin_amdgcn_s_memtime(); }
#define KP_MARK(tr) trace_mark(tr)
#else
struct Trace { bool on; };
#define KP_MARK(tr)
#endif

constexpr int kWavesP = 8;
constexpr int kThreadsP = kWavesP * 64;
constexpr int kXchgTile = 2048;                        // bytes one wave sends per tile: 8 registers x 64 lanes
constexpr int kXchgBytes = 4 * 2 * 2 * kXchgTile;      // [pair][parity][direction]: 32 KiB
constexpr int kLdsP = kLdsBytes + kXchgBytes;

struct PiecesP { f16x8 h[8], m[8]; };      // the wave's 8 k-steps of a layer input (global k-step 2 s + role): 64 VGPRs

// ---- weight ring (two 36-KiB chunk buffers), copied by the whole workgroup: wave w moves pieces w, w + 8, ...
struct StreamP {
  const f32x4* g;
  f32x4* buf;
  int cur;
  int wb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  __device__ __forceinline__ const f32x4* cur_buf() const { return buf + cur * kChunkF4; }
  template <int N16>
  __device__ __forceinline__ void piece(int i) {
    const int idx = i * kThreadsP + wb;
    if (i < (N16 + kThreadsP - 1) / kThreadsP && ((i + 1) * kThreadsP <= N16 || idx < N16)) {
      f32x4* dst = buf + (cur ^ 1) * kChunkF4 + idx;
      const unsigned lane_bytes = (threadIdx.x & 63u) * 16u;
      const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)dst;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :: "v"(lane_bytes), "s"(g + idx), "s"(lds_base) : "memory");
    }
  }
  template <int N16>
  __device__ __forceinline__ void prefetch() {
#pragma unroll
    for (int i = 0; i < (N16 + kThreadsP - 1) / kThreadsP; ++i) piece<N16>(i);
    g += N16;
  }
  template <int N16>
  __device__ __forceinline__ void done() { g += N16; }
  // own copies landed, then everyone's; N: the wave's youngest vector-memory operations that may stay in flight
  template <int N = 0>
  __device__ __forceinline__ void advance() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
    cur ^= 1;
  }
};

// ---- the exchange of partial accumulators between the two waves of a pair
struct Xchg {
  unsigned char* base;      // this pair's area: [parity][direction][2 KiB]
  int role;
  // send the 8 registers the partner owns (role 0 sends 8..15, role 1 sends 0..7) of tile parity `par`
  __device__ __forceinline__ void send(const f32x16& acc, int par, int lane) const {
    f32x4* d = reinterpret_cast<f32x4*>(base + (par * 2 + role) * kXchgTile) + lane;
    f32x4 v0, v1;
    if (role == 0) { v0 = {acc[8], acc[9], acc[10], acc[11]}; v1 = {acc[12], acc[13], acc[14], acc[15]}; }
    else { v0 = {acc[0], acc[1], acc[2], acc[3]}; v1 = {acc[4], acc[5], acc[6], acc[7]}; }
    d[0] = v0; d[64] = v1;
  }
  // the partner's partial sums of the 8 registers this wave owns (issue early: two ds_read_b128)
  __device__ __forceinline__ void recv(int par, int lane, f32x4& r0, f32x4& r1) const {
    const f32x4* s = reinterpret_cast<const f32x4*>(base + (par * 2 + (role ^ 1)) * kXchgTile) + lane;
    r0 = s[0]; r1 = s[64];
  }
};
// the 8 accumulator registers this wave owns
__device__ __forceinline__ void own_half(const f32x16& acc, int role, float* o) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = role ? acc[8 + j] : acc[j];
}

struct NoE { __device__ __forceinline__ void operator()(int) const {} };
struct NoP { __device__ __forceinline__ void operator()() const {} };

// One 32-row output tile over the wave's 8 k-steps (global k-step 2 s + role): role 0 starts from the bias block, role 1
// from zero.  ea / eb / ec(s): slices of the previous tile's epilogue behind the three MFMAs of k-step s; pre(): vector work
// issued while the tile waits for its first LDS reads; piece i of the next chunk's LDS-DMA goes behind k-step i.
template <int N16NEXT, typename EA, typename EB, typename EC, typename Pre>
__device__ __forceinline__ f32x16 tile_mma_p(StreamP& st, const PiecesP& x, int lane, int role, EA ea, EB eb, EC ec, Pre pre,
                                              Trace& tr) {
  const f32x4* chunk = st.cur_buf();
  f32x16 acc = tile_bias(chunk, lane);
  if (role) acc = (f32x16)(0.0f);
  const f16x8* a_ptr = reinterpret_cast<const f16x8*>(chunk + kHdrF4) + lane + role * 128;
  // MFMAs in blocks of KP_GROUP k-steps issued back to back, each followed by the vector work of those k-steps (epilogue
  // slices, the next fragments' LDS reads, an LDS-DMA piece).  With two waves per SIMD the matrix core is shared at MFMA
  // granularity: a partner can only use a gap of >= 32 contiguous cycles, so vector work sliced between single MFMAs (the
  // one-wave kernels' schedule) leaves the partner nothing; blocks make the two waves alternate -- one's vector block under
  // the other's MFMA block.
#ifndef KP_GROUP
#define KP_GROUP 1
#endif
  constexpr int G = KP_GROUP;
  f16x8 fh[G + 1 > 8 ? 8 : 2 * G], fm[G + 1 > 8 ? 8 : 2 * G];      // fragments of the current and the next block
#pragma unroll
  for (int i = 0; i < G; ++i) { fh[i] = a_ptr[(4 * i) * 64]; fm[i] = a_ptr[(4 * i + 1) * 64]; }
  __builtin_amdgcn_sched_barrier(0);
  pre();
  // The two waves of a SIMD start every tile together (barrier) and, running the same order, stay roughly in lockstep (KP_TRACE
  // stamps).  -DKP_FLIP makes role 1 run each block pair in the opposite order, vector block first, so that one wave's vector
  // block would lie under the other's MFMA block: measured SLOWER (0.364 vs 0.337 ms for the sdf-only evaluation of 131 072
  // points; the one-wave-per-SIMD kernel: 0.302).
#pragma unroll
  for (int b = 0; b < 8 / G; ++b) {
    const int cur = (b & 1) * G, nxt = ((b + 1) & 1) * G;
    auto mblock = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      KP_MARK(tr);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int s = b * G + i;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fm[cur + i], x.h[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur + i], x.m[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur + i], x.h[s], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto vblock = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      KP_MARK(tr);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int s = b * G + i;
        if (s + G < 8) { fh[nxt + i] = a_ptr[(4 * (s + G)) * 64]; fm[nxt + i] = a_ptr[(4 * (s + G) + 1) * 64]; }
        ea(s); eb(s); ec(s);
        if (N16NEXT > 0) st.template piece<N16NEXT>(s);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#ifdef KP_FLIP
    if (role == 0) { mblock(); vblock(); } else { vblock(); mblock(); }
#else
    mblock(); vblock();
#endif
  }
  if (N16NEXT > 0) st.template done<N16NEXT>();
  return acc;
}

// a full-K tile over KS k-steps from a Pieces2 operand (layer 0: both waves compute whole tiles and keep their half)
template <int KS, int N16NEXT>
__device__ __forceinline__ f32x16 tile_mma_full(StreamP& st, const Pieces2& x, int lane) {
  const f32x4* chunk = st.cur_buf();
  f32x16 acc = tile_bias(chunk, lane);
  const f16x8* a_ptr = reinterpret_cast<const f16x8*>(chunk + kHdrF4) + lane;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const f16x8 ah = a_ptr[(2 * s) * 64], am = a_ptr[(2 * s + 1) * 64];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, x.h[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, x.m[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, x.h[s], acc, 0, 0, 0);
  }
  if (N16NEXT > 0) st.template prefetch<N16NEXT>();
  return acc;
}

}  // namespace kp
}  // namespace mlp
}  // namespace svs
