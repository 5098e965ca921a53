// 16-point wave tiles, two waves per SIMD: the SDF network's forward evaluation (ImplicitNetwork.get_sdf_vals,
// volsdf/model/network.py:125-131) on v_mfma_f32_16x16x32_f16 with the fp16x2 operand split of svs_mlp_h2_dev.h.
//
// Why a second tiling.  The 32-point kernels (svs_mlp_h2.hip) keep a wave's 256 x 32 activations as MFMA B fragments in
// registers -- 128 VGPRs for the layer input, 128 for the pieces being produced -- which forces one wave per SIMD, and at
// one wave per SIMD every softplus / split / store instruction is issued between that wave's own MFMAs, in order: the matrix
// core idles whenever a gap's vector work exceeds ~24 cycles (0.47 of the MFMA peak forward-only, 0.36
// with stores).  A 16-point wave needs 64 + 64 VGPRs, so two waves share a SIMD and one wave's epilogue issues under the
// other's MFMAs.  The price is the A operand: the same 36-KiB weight tile now feeds 16 points per wave instead of 32, i.e.
// twice the LDS read traffic per point -- 8 waves x 2 ds_read_b128 per 3 MFMAs = 64 of every 96 LDS cycles at the LDS's 256
// B/clk, which is why the 16x16x32 shape is affordable at all (MI355X_MICROARCH.md, LDS).
//
// Layout.  A workgroup = 8 waves = 128 points (the grid, the gate and the LDS weight ring are those of the 32-point kernel).
// MFMA roles: A = weights (16 output rows x 32 inputs), B = activations (32 inputs x 16 points), C/D lane (g = lane >> 4,
// n = lane & 15) holds rows 4g..4g+3 of point n.  Chunks are packed in the kFmtF16x2W encoding (svs_mlp_layout.h), whose K
// order makes the four accumulator registers of output tiles 2s and 2s+1 the B fragment of the next layer's k-step s: no
// cross-lane traffic between layers.  One chunk = 32 output rows = two sub-tiles = exactly one k-step of the next layer.
#include "svs_mlp_h2_dev.h"
#include "svs_mlp_host.h"
#include "svs_mlp_args.h"
#include <cstdlib>

namespace svs {
namespace mlp {
namespace w16 {

// diagnostic switches (never defined in the product build): W16_DEPTH = k-steps of A fragments requested ahead (default 2),
// W16_NOSTAGGER, W16_NOEPI (identity instead of softplus: times the MFMA + LDS skeleton; results wrong)
#ifdef W16_NOEPI
#define W16_ACT(x) (x)
#else
#define W16_ACT(x) softplus100(x)
#endif
#ifndef W16_DEPTH
#define W16_DEPTH 2
#endif

// NW = waves per workgroup: 8 (128 points, two waves per SIMD from ONE workgroup) or 4 (64 points, one wave per SIMD; two
// workgroups share a CU, 72 KiB of LDS each, so the two waves of a SIMD belong to different workgroups and do not meet at
// each other's barriers -- at twice the L2 -> LDS weight traffic per point)
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct PiecesW { f16x8 h[8], m[8]; };      // one layer input: 8 k-steps of 32 rows, both fp16 pieces = 64 VGPRs

// ---- weight ring: two chunk buffers, the whole workgroup copies a chunk by LDS-DMA (wave w moves pieces w, w + 8, ...)
template <int NW>
struct StreamW {
  static constexpr int kThreadsW = NW * 64;
  const f32x4* g;
  f32x4* buf;
  int cur;
  int wb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  __device__ __forceinline__ const f32x4* cur_buf() const { return buf + cur * kChunkF4; }
  template <int N16>
  __device__ __forceinline__ void prefetch() {
    f32x4* dst = buf + (cur ^ 1) * kChunkF4;
    const unsigned lane_bytes = (threadIdx.x & 63u) * 16u;
#pragma unroll
    for (int i = 0; i < (N16 + kThreadsW - 1) / kThreadsW; ++i) {
      const int idx = i * kThreadsW + wb;
      if ((i + 1) * kThreadsW <= N16 || idx < N16) {
        const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(dst + idx);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(lane_bytes), "s"(g + idx), "s"(lds_base) : "memory");
      }
    }
    g += N16;
  }
  // the same copy in pieces: piece i of the round-robin (a wave-instruction of 1 KiB per wave), issued between MFMA groups
  template <int N16>
  __device__ __forceinline__ void prefetch_piece(int i) {
    f32x4* dst = buf + (cur ^ 1) * kChunkF4;
    const unsigned lane_bytes = (threadIdx.x & 63u) * 16u;
    const int idx = i * kThreadsW + wb;
    if (i < (N16 + kThreadsW - 1) / kThreadsW && ((i + 1) * kThreadsW <= N16 || idx < N16)) {
      const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(dst + idx);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :: "v"(lane_bytes), "s"(g + idx), "s"(lds_base) : "memory");
    }
  }
  template <int N16>
  __device__ __forceinline__ void prefetch_done() { g += N16; }
  __device__ __forceinline__ void advance() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cur ^= 1;
  }
};

__device__ __forceinline__ f32x4v mma16(const f16x8& a, const f16x8& b, const f32x4v& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// One chunk = two 16-row sub-tiles x KS k-steps, as a flat sequence of 2 KS steps of three MFMAs.  The A fragments of
// step i + 2 are requested before the MFMAs of step i (a ring of three fragment pairs: hipcc left to itself reads each
// fragment right in front of its MFMA and exposes the LDS latency 2 KS times per chunk); piece i of the next chunk's
// LDS-DMA goes behind step i; epi(u, acc) -- the activation of a finished sub-tile -- is called behind its last step.
// NU: sub-tiles to run (the head uses one).
template <int KS, int N16NEXT, int NU, int NW, typename Epi>
__device__ __forceinline__ void chunk_mma(StreamW<NW>& st, const PiecesW& x, int lane, Epi epi) {
  constexpr int kThreadsW = NW * 64;
  const f32x4* chunk = st.cur_buf();
  const f16x8* a_ptr = reinterpret_cast<const f16x8*>(chunk + kHdrF4) + lane;
  constexpr int N = NU * KS, D = W16_DEPTH, RING = D + 1;
  f16x8 ah[RING], am[RING];
  f32x4 bias[2];
#pragma unroll
  for (int i = 0; i < D && i < N; ++i) { ah[i] = a_ptr[(2 * i) * 64]; am[i] = a_ptr[(2 * i + 1) * 64]; }
  bias[0] = chunk[lane];
  if (NU > 1) bias[1] = chunk[64 + lane];
  f32x4v acc;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int u = i / KS, s = i % KS;
    __builtin_amdgcn_sched_barrier(0);
    if (i + D < N) { ah[(i + D) % RING] = a_ptr[(2 * (i + D)) * 64]; am[(i + D) % RING] = a_ptr[(2 * (i + D) + 1) * 64]; }
    if (s == 0) { acc[0] = bias[u][0]; acc[1] = bias[u][1]; acc[2] = bias[u][2]; acc[3] = bias[u][3]; }
    __builtin_amdgcn_sched_barrier(0);
    acc = mma16(am[i % RING], x.h[s], acc);
    acc = mma16(ah[i % RING], x.m[s], acc);
    acc = mma16(ah[i % RING], x.h[s], acc);
    __builtin_amdgcn_sched_barrier(0);
    if (N16NEXT > 0) st.template prefetch_piece<N16NEXT>(i);
    if (s == KS - 1) epi(u, acc);
  }
  if (N16NEXT > 0) {
#pragma unroll
    for (int i = N; i < (N16NEXT + kThreadsW - 1) / kThreadsW; ++i) st.template prefetch_piece<N16NEXT>(i);   // (short chunks)
    st.template prefetch_done<N16NEXT>();
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void sdf_only_w16_kernel(SdfOnlyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kWavesW = NW;
  if (a.gate && a.gate[(size_t)(blockIdx.x * (NW * 16) / a.gate_points) * a.gate_stride] == 0) return;
  StreamW<NW> st;
  st.g = a.stream;
  st.buf = reinterpret_cast<f32x4*>(smem);
  st.cur = 1;
  const int lane = threadIdx.x & 63, g = lane >> 4, wave = threadIdx.x >> 6;
  const int p = (blockIdx.x * kWavesW + wave) * 16 + (lane & 15);

  st.template prefetch<kChunk0WF4>();
  float x0, x1, x2;
  load_point(a.src, p, x0, x1, x2);
  const float r2 = x0 * x0 + x1 * x1 + x2 * x2;
  PosEnc pe;
  pe.compute(x0, x1, x2);

  PiecesW x, xn;
  // layer-0 operand: k-step s, element j of lane group g = PE[32 s + 8 g + j] (zero beyond the 39 entries)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float sel = 0.0f;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        const int q = 32 * s + 8 * gg + j;
        const float c = q < kPeDim ? pe.v[q < kPeDim ? q : 0] : 0.0f;
        sel = g == gg ? c : sel;
      }
      v[j] = sel;
    }
    split8(v, x.h[s], x.m[s]);
  }
  st.advance();

  // The two waves of a SIMD (w and w + 4) run the same program between the same barriers; in lockstep they would want the
  // matrix core together and leave it idle together.  Waves 4..7 therefore run half a chunk late: they finish a chunk's
  // second sub-tile (activation + split) at the START of the next chunk, while waves 0..3 do it at the end of their own
  // (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  `late` is wave-uniform: scalar branches.
#ifdef W16_NOSTAGGER
  const bool late = false;
#else
  const bool late = NW == 8 && __builtin_amdgcn_readfirstlane(wave) >= 4;
#endif
  float v[8];
  f32x4v held = {0.0f, 0.0f, 0.0f, 0.0f};     // a late wave's second sub-tile, carried across the barrier
  // ---- layer 0: 39 (64) -> 256
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    auto finish = [&](int tt, const f32x4v& acc1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 + r] = W16_ACT(acc1[r]);
      split8(v, xn.h[tt], xn.m[tt]);
    };
    if (late && t > 0) finish(t - 1, held);
    auto epi = [&](int u, const f32x4v& acc) {
      if (u == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = W16_ACT(acc[r]);
      } else if (late) held = acc;
      else finish(t, acc);
    };
    if (t < 7) chunk_mma<2, kChunk0WF4, 2, NW>(st, x, lane, epi); else chunk_mma<2, kChunkF4, 2, NW>(st, x, lane, epi);
    st.advance();
  }
  // ---- layers 1..7 (layer 3 emits 217 rows; rows 217..255 of its output are the PE splice, network.py:80-81)
  for (int l = 1; l < 8; ++l) {
    const bool l3 = l == 3, prev_l3 = l == 4;
    // a late wave still owes the previous layer's last sub-tile (k-step 7 of this layer's input; after layer 3, k-step 6)
    auto finish_into = [&](PiecesW& dst, int tt, const f32x4v& acc1, bool splice) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float o = W16_ACT(acc1[r]);
        if (splice) {
          // layer 3, chunk 6, second sub-tile: rows 208 + 4 g + r, of which >= 217 carry PE[32 + row - 217]
#pragma unroll
          for (int gg = 2; gg < 4; ++gg) {
            const int row = 208 + 4 * gg + r;
            if (row >= 217) o = g == gg ? pe.v[32 + row - 217] : o;
          }
        }
        v[4 + r] = o;
      }
      split8(v, dst.h[tt], dst.m[tt]);
    };
    if (late) {
      if (prev_l3) finish_into(xn, 6, held, true); else finish_into(xn, 7, held, false);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) { x.h[s] = xn.h[s]; x.m[s] = xn.m[s]; }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t == 7 && l3) break;
      if (late && t > 0) finish_into(xn, t - 1, held, false);
      auto epi = [&](int u, const f32x4v& acc) {
        if (u == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = W16_ACT(acc[r]);
        } else if (late) held = acc;
        else finish_into(xn, t, acc, t == 6 && l3);
      };
      chunk_mma<8, kChunkF4, 2, NW>(st, x, lane, epi);
      st.advance();
    }
    if (l3) {
      // output tiles 14, 15 = PE[0..31]: k-step 7, element j of lane group g = PE[16 (j >> 2) + 4 g + (j & 3)]
      float w8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float sel = 0.0f;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) sel = g == gg ? pe.v[16 * (j >> 2) + 4 * gg + (j & 3)] : sel;
        w8[j] = sel;
      }
      split8(w8, xn.h[7], xn.m[7]);
    }
  }
  if (late) {
    // layer 7's last sub-tile
#pragma unroll
    for (int r = 0; r < 4; ++r) v[4 + r] = W16_ACT(held[r]);
    split8(v, xn.h[7], xn.m[7]);
  }
  // ---- head: rows 0..15 of lin8 (one sub-tile); row 0 = sdf = register 0 of lane group 0
  float sdf = 0.0f;
  chunk_mma<8, 0, 1, NW>(st, xn, lane, [&](int, const f32x4v& acc) { sdf = acc[0]; });
  if (a.sphere_radius > 0.0f && p < a.clamp_n) {
    const float nrm = __builtin_sqrtf(r2);
    sdf = __builtin_fminf(sdf, a.sphere_scale * (a.sphere_radius - nrm));
  }
  if (g == 0 && p < a.src.P) a.sdf[p] = sdf;
}

}  // namespace w16
}  // namespace mlp
}  // namespace svs

using namespace svs;
using namespace svs::mlp;

extern "C" {

// svs_sdf_vals with the 16-point-wave kernel: `stream` = svs_pack_stream which = 9 (fp16x2 only); every other argument as in
// svs_sdf_vals (include/svolsdf_hip.h)
int svs_sdf_vals16(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                   int S, int n_rays, const float* stream, float sphere_radius, float sphere_scale, int clamp_n, float* sdf,
                   const int* gate, int gate_points, int gate_stride, void* hip_stream) {
  SdfOnlyArgs a;
  if (int rc = fill_src(a.src, points, n_points, cam, cam_stride, dirs, z, S, n_rays, "svs_sdf_vals16")) return rc;
  if (!stream || !sdf) { set_error("svs_sdf_vals16: null stream/sdf"); return SVS_EINVAL; }
  a.stream = reinterpret_cast<const f32x4*>(stream); a.sdf = sdf;
  a.sphere_radius = sphere_radius; a.sphere_scale = sphere_scale; a.gate = gate;
  a.gate_points = gate_points > 0 ? gate_points : 0x7fffff80; a.gate_stride = gate_stride;
  if (gate && a.gate_points % kWgPts) { set_error("svs_sdf_vals16: gate_points must be a multiple of %d", kWgPts); return SVS_EINVAL; }
  a.clamp_n = clamp_n < 0 ? a.src.P : clamp_n;
  static int once = set_lds(w16::sdf_only_w16_kernel<8>, kLdsBytes, "svs_sdf_vals16") | set_lds(w16::sdf_only_w16_kernel<4>, kLdsBytes, "svs_sdf_vals16");
  if (once) return once;
  static const int nw = (getenv("SVS_W16_WAVES") && atoi(getenv("SVS_W16_WAVES")) == 4) ? 4 : 8;
  if (nw == 4) {
    if (gate && a.gate_points % 64) { set_error("svs_sdf_vals16: gate_points must be a multiple of 64"); return SVS_EINVAL; }
    w16::sdf_only_w16_kernel<4><<<(a.src.P + 63) / 64, 256, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  } else {
    w16::sdf_only_w16_kernel<8><<<(a.src.P + kWgPts - 1) / kWgPts, 512, kLdsBytes, (hipStream_t)hip_stream>>>(a);
  }
  return check_launch("svs_sdf_vals16");
}

}  // extern "C"
