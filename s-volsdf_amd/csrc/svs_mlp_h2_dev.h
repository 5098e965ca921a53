// fp16x2 building blocks of the fused MLP kernels: float32-class products on the 16-bit matrix cores.
//
// Every float32 operand a is split into two fp16 pieces a = hi + mid (hi = fp16(a), mid = fp16(a - hi): 22
// significand bits, absolute floor 2^-25 for |a| < 2^-3), and sum_k a_k b_k is evaluated as hi*hi + hi*mid + mid*hi
// accumulated in float32 by v_mfma_f32_32x32x16_f16; the dropped mid*mid term is below 2^-22 |a b|.  Three MFMAs
// of 32 cycles (K = 16) replace eight v_mfma_f32_32x32x2_f32 of 64 cycles (K = 2): 5.3x fewer matrix-core cycles.
// Measured on the SDF network (tools/bench_kernels.py): max |sdf - float64| 1.5e-6 vs 1.2e-6 for the float32 MFMA
// kernels, 2.8x their speed.  (Three bf16 pieces with six products reach the same accuracy at 1.55x; bf16 pieces with
// three products are 10x less accurate.)  Operands must stay below fp16's 65504: true for SDF / radiance activations
// and weights in scene units; the float32 kernels (precision = SVS_MMA_F32) have no such limit.
//
// Activations stay transposed in registers as in svs_mlp_dev.h: accumulator registers 8s..8s+7 of input tile tau are
// the B fragment of k-step 2*tau + s (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand"); the
// weights are packed in that K order (svs_pack.hip, fmt = 1): per k-step, 64 lanes x 16 B of the hi piece, then of
// the mid piece -- a chunk is as large as its float32 counterpart, so stream layouts do not depend on the precision.
#pragma once
#include "svs_mlp_dev.h"

namespace svs {
namespace mlp {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// one layer input as MFMA B fragments: 16 k-steps (K = 256) x {hi, mid} = 128 VGPRs; k-steps 16, 17 hold the extra
// input rows of the radiance networks' first layer (16 fg / 27 bg; unused elsewhere: they cost no registers)
struct Pieces2 {
  f16x8 h[18], m[18];
};

__device__ __forceinline__ void split8(const float* v, f16x8& h, f16x8& m) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 ah = (_Float16)v[j];
    h[j] = ah;
    m[j] = (_Float16)(v[j] - (float)ah);
  }
}

// LLVM sinks pure arithmetic past the scheduling barriers of the k-loop; pinning a value makes it (and what it
// depends on) stay in the slot it was written in.
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(f16x8& a, f16x8& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// accumulator tile t (16 registers) -> k-steps 2t, 2t+1 of p
__device__ __forceinline__ void split_tile(const f32x16& y, int t, Pieces2& p) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = y[8 * s + j];
    split8(v, p.h[2 * t + s], p.m[2 * t + s]);
  }
}

// layer-0 input: k-step s (0..KS-1), element j of lane half h is PE[16 s + 8 h + j] (zero beyond the PAD entries)
template <int KS, int PAD>
__device__ __forceinline__ void split_pe(const float* pev, int half, Pieces2& p) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q0 = 16 * s + j, q1 = 16 * s + 8 + j;
      const float a0 = q0 < PAD ? pev[q0 < PAD ? q0 : 0] : 0.0f;
      const float a1 = q1 < PAD ? pev[q1 < PAD ? q1 : 0] : 0.0f;
      v[j] = half ? a1 : a0;
    }
    split8(v, p.h[s], p.m[s]);
  }
}
__device__ __forceinline__ void split_pe(const PosEnc& pe, int half, Pieces2& p) { split_pe<3, 40>(pe.v, half, p); }

__device__ __forceinline__ f32x16 tile_bias(const f32x4* __restrict__ chunk, int lane) {
  f32x16 acc;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const f32x4 b = chunk[r4 * 64 + lane];
    acc[4 * r4 + 0] = b[0]; acc[4 * r4 + 1] = b[1]; acc[4 * r4 + 2] = b[2]; acc[4 * r4 + 3] = b[3];
  }
  return acc;
}

// Diagnostic builds (tools/ablate_fwd.sh): -DSVS_ABL=<mask> compiles parts of the forward kernels out to time the rest
// (results are then wrong by construction).  1: identity instead of softplus, 2: no operand split, 4: no MFMAs,
// 8: no chunk wait / barrier, 16: cycle stamps instead of results (sdf_only), 32: let the slimmer variants run two
// workgroups per CU (otherwise their LDS request is padded to keep one), 64: no weight fetch; reverse pass of sdf_full:
// 128: no h loads, 256: no gbuf stores, 512: no softplus' arithmetic; pass A (svs_mlp_bwd_h2.hip): 1024 no side-tile loads,
// 2048 no u stores; pass B: 4096 no side-tile loads, 8192 no abar stores, 16384 no second-order / sbar terms, 32768 no softplus'
// arithmetic, 65536 per-tile cycle stamps into sbar_out (tools/bench_kernels.py BK_STAMPS=1); the weight fetch (svs_mlp_dev.h):
// 131072 a quarter of the lanes per LDS-DMA instruction, 262144 every lane the same address.  Never defined in the product build.
#ifndef SVS_ABL
#define SVS_ABL 0
#endif
#if SVS_ABL & 4
#define SVS_ABL_MFMA(x)
#else
#define SVS_ABL_MFMA(x) x
#endif

struct NoEpi { __device__ __forceinline__ void operator()(int) const {} };
struct NoPre { __device__ __forceinline__ void operator()() const {} };

// One output tile: acc(32 rows x 32 points) = hdr(bias) + sum over KS k-steps of (mid*hi + hi*mid + hi*hi).
// ea(s) / eb(s) are two slices of the PREVIOUS tile's epilogue (activation, split, stores) issued behind the first /
// second MFMA of k-step s, so that their VALU work runs while the matrix core is busy.  The A fragments of k-step
// s+1 are read right after the first MFMA of k-step s: the wait hipcc places before their first use (always
// lgkmcnt(0)) then has two MFMAs of cover.
// dma(s): the slice of the next chunk's LDS-DMA issued behind the third MFMA of k-step s (Stream::prefetch_step).
// pre(): vector work issued between the tile's first LDS reads (bias block, first A fragments) and its first MFMA, where
// the wave otherwise only waits for those reads.
template <int KS, typename EpiA, typename EpiB, typename Dma = NoEpi, typename EpiC = NoEpi, typename Pre = NoPre>
__device__ __forceinline__ f32x16 tile_mma_h2(const f32x4* __restrict__ chunk, const Pieces2& x, int lane, EpiA ea, EpiB eb,
                                              int first_step = 0, Dma dma = Dma(), EpiC ec = EpiC(), Pre pre = Pre()) {
  f32x16 acc = tile_bias(chunk, lane);
  const f16x8* a_ptr = reinterpret_cast<const f16x8*>(chunk + kHdrF4) + lane;
  f16x8 ah = a_ptr[0], am = a_ptr[64];
  __builtin_amdgcn_sched_barrier(0);
  pre();
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f16x8 nh, nm;
    __builtin_amdgcn_sched_barrier(0);
    SVS_ABL_MFMA(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, x.h[first_step + s], acc, 0, 0, 0));
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < KS) { nh = a_ptr[(2 * s + 2) * 64]; nm = a_ptr[(2 * s + 3) * 64]; }
    ea(s);
    __builtin_amdgcn_sched_barrier(0);
    SVS_ABL_MFMA(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, x.m[first_step + s], acc, 0, 0, 0));
    __builtin_amdgcn_sched_barrier(0);
    eb(s);
    __builtin_amdgcn_sched_barrier(0);
    SVS_ABL_MFMA(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, x.h[first_step + s], acc, 0, 0, 0));
    __builtin_amdgcn_sched_barrier(0);
    ec(s);
    dma(s);
    if (s + 1 < KS) { ah = nh; am = nm; }
  }
  return acc;
}
template <int KS>
__device__ __forceinline__ f32x16 tile_mma_h2(const f32x4* __restrict__ chunk, const Pieces2& x, int lane) {
  return tile_mma_h2<KS>(chunk, x, lane, NoEpi(), NoEpi());
}

// tile_mma_h2 on the current chunk with the prefetch of the next chunk (N16 float4) spread over its k-steps
template <int KS, int N16, typename S, typename EpiA, typename EpiB, typename EpiC = NoEpi, typename Pre = NoPre>
__device__ __forceinline__ f32x16 tile_mma_h2_pf(S& st, const Pieces2& x, int lane, EpiA ea, EpiB eb, EpiC ec = EpiC(),
                                                 Pre pre = Pre()) {
  const f32x16 acc = tile_mma_h2<KS>(st.cur_buf(), x, lane, ea, eb, 0, [&](int s) { st.template prefetch_step<N16, KS>(s); }, ec,
                                     pre);
  st.template prefetch_done<N16>();
  return acc;
}

// softplus100 in slices (see tile_mma_h2 and TrunkEpi): exp2 | max, log2 | the final fma
struct SoftplusA { float mx, lg; };
__device__ __forceinline__ float softplus100_b(const SoftplusA& r) {
  return r.mx + (0.69314718055994531f / 100.0f) * r.lg;
}

}  // namespace mlp
}  // namespace svs
