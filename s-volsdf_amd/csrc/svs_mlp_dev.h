// Device-side building blocks shared by the fused MLP kernels (svs_mlp.hip: forward / input-gradient pass,
// svs_mlp_bwd.hip: training backward).  See svs_mlp.hip for the design notes.
#pragma once
#include "svs_common.h"
#include "svs_mlp_layout.h"

#ifndef SVS_DMA_ASM
#define SVS_DMA_ASM 1
#endif

namespace svs {
namespace mlp {

// ------------------------------------------------------------------------------------------------------
// weight stream
// ------------------------------------------------------------------------------------------------------
// LDS-DMA copy of N16 float4 (N16 % 64 == 0) from global to LDS, issued by the whole workgroup: wave w moves the
// 1 KiB pieces w, w + 4, ...  The wave's base is made provably uniform (readfirstlane) so that hipcc addresses a piece as
// scalar base + 32-bit lane offset and sets M0 from an SGPR; with a per-lane 64-bit pointer every piece cost five extra
// vector instructions.
__device__ __forceinline__ int wave_base_f4() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u)); }
template <int N16>
__device__ __forceinline__ void chunk_issue_piece(const f32x4* __restrict__ g, f32x4* lds, int i, int wave_base) {
  static_assert(N16 % 64 == 0, "chunk must be a whole number of wave-instructions");
  const int idx = i * kThreads + wave_base;  // wave-uniform (wave_base_f4(), read once per kernel: Stream::wb)
  const unsigned lane_bytes = (threadIdx.x & 63u) * 16u;
  if ((i + 1) * kThreads <= N16 || idx < N16) {   // i is a constant after unrolling: whole rounds carry no branch
#if SVS_DMA_ASM
    // Inline assembly, so that hipcc does not know the ring is written by vector-memory instructions.  The ring is
    // ordered by hand (Stream::advance / advance_keep<N>: counted vmcnt + barrier, inline assembly as well, hence
    // invisible to the waitcnt pass), and a DMA the pass can see makes it put `s_waitcnt vmcnt(0)` in front of the first
    // LDS read of (every other) tile -- which drained exactly the loads and stores advance_keep<N> had left in flight:
    // the reverse pass of sdf_full sat out the HBM latency of the h tile it had just requested, 32 k of its 180 k cycles.
    // The pass now under-counts the operations in flight when it waits for an ordinary load, i.e. it waits for too much
    // rather than too little: consumers therefore touch a side tile once at the TOP of the tile that uses it, before
    // the tile's first piece is issued, where the count is exact (reverse_layer_h2).
    // Address = SGPR pair + 32-bit lane offset: no per-piece vector address arithmetic.  M0 (LDS base of the piece) is a
    // reserved register that hipcc re-materialises before each use of its own; one wait state before the DMA reads it.
    // (Listing "m0" as a clobber is not possible: hipcc rejects reserved registers on the clobber list, -Winline-asm
    // "may lead to undefined behaviour".  No other M0 user exists in the translation units that contain this asm: no
    // s_movrel / sendmsg / builtin LDS-DMA.)
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(lds + idx);
#if defined(SVS_ABL) && (SVS_ABL & 131072)      // diagnostic: a quarter of the lanes only (same instruction count, a quarter of the data)
    asm volatile("s_mov_b64 s[96:97], exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                 "s_mov_b64 exec, s[96:97]" :: "v"(lane_bytes), "s"(g + idx), "s"(lds_base) : "memory", "s96", "s97");
#elif defined(SVS_ABL) && (SVS_ABL & 262144)    // diagnostic: every lane fetches the SAME 16 bytes (same LDS write, one line of L2 traffic)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(0u), "s"(g + idx), "s"(lds_base) : "memory");
#else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_bytes), "s"(g + idx), "s"(lds_base) : "memory");
#endif
#else
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g + idx) + lane_bytes),
        (__attribute__((address_space(3))) void*)(lds + idx), 16, 0, 0);
#endif
  }
}
template <int N16>
__device__ __forceinline__ void chunk_issue(const f32x4* __restrict__ g, f32x4* lds, int wave_base) {
#pragma unroll
  for (int i = 0; i < (N16 + kThreads - 1) / kThreads; ++i) chunk_issue_piece<N16>(g, lds, i, wave_base);
}
template <int N16>
constexpr int chunk_pieces() { return (N16 + kThreads - 1) / kThreads; }

// BUF: float4 per LDS buffer (two of them).
template <int BUF>
struct StreamT {
  const f32x4* g;  // next chunk to fetch
  f32x4* buf;      // LDS: two buffers of BUF float4
  int cur;         // buffer holding the chunk being consumed
  int wb = wave_base_f4();   // the wave's float4 offset within a round of pieces, in an SGPR for the whole kernel

  __device__ __forceinline__ const f32x4* cur_buf() const { return buf + cur * BUF; }
  template <int N16>
  __device__ __forceinline__ void prefetch() {
#if !(defined(SVS_ABL) && (SVS_ABL & 64))
    chunk_issue<N16>(g, buf + (cur ^ 1) * BUF, wb);
#endif
    g += N16;
  }
  // prefetch() in pieces: ceil(pieces / KS) of them at each of the first steps of a KS-step tile, then prefetch_done().
  // An LDS-DMA instruction costs its wave 60+ cycles of issue; spread over the first MFMA gaps of the tile the pieces
  // run under the matrix core instead of in front of it, and still land long before the tile ends.
  template <int N16, int KS>
  __device__ __forceinline__ void prefetch_step(int s) {
#if !(defined(SVS_ABL) && (SVS_ABL & 64))
    constexpr int np = chunk_pieces<N16>(), per = (np + KS - 1) / KS;
#pragma unroll
    for (int i = per * s; i < per * (s + 1) && i < np; ++i) chunk_issue_piece<N16>(g, buf + (cur ^ 1) * BUF, i, wb);
#endif
  }
  template <int N16>
  __device__ __forceinline__ void prefetch_done() { g += N16; }
  // the chunk fetched by prefetch() becomes current: own loads landed, then everyone's
  __device__ __forceinline__ void advance() {
#if !(defined(SVS_ABL) && (SVS_ABL & 8))
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) lgkmcnt(0) expcnt(0)
    __syncthreads();
#endif
    cur ^= 1;
  }
  // The same, but the N youngest vector-memory operations of the wave -- float32 activation stores that the epilogue
  // issued AFTER prefetch() -- stay in flight: loads, stores and LDS-DMA retire from vmcnt in issue order, so
  // "all but the N youngest" covers the chunk.  Raw s_barrier: __syncthreads() carries a fence that hipcc lowers to
  // vmcnt(0), which made every tile wait for its stores to reach memory (12 % of sdf_full).  N must not exceed the number
  // of vector-memory instructions the wave really issued after prefetch().
  template <int N>
  __device__ __forceinline__ void advance_keep() {
#if !(defined(SVS_ABL) && (SVS_ABL & 8))
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
    cur ^= 1;
  }
};
typedef StreamT<kChunkF4> Stream;

// ------------------------------------------------------------------------------------------------------
// element-wise pieces
// ------------------------------------------------------------------------------------------------------
// softplus(beta=100) = max(a,0) + log1p(exp(-|100 a|))/100 through the hardware exp2/log2: the absolute
// error is < 1e-9 (the reference switches to the identity above 100a > 20, where the two differ by 2e-11).
__device__ __forceinline__ float softplus100(float a) {
  const float e = __builtin_amdgcn_exp2f(__builtin_fabsf(a) * (-100.0f * 1.44269504088896341f));
  return __builtin_fmaxf(a, 0.0f) + (0.69314718055994531f / 100.0f) * __builtin_amdgcn_logf(1.0f + e);
}
// d softplus100 / da as a function of h = softplus100(a):  sigmoid(100 a) = 1 - exp(-100 h)
__device__ __forceinline__ float dsoftplus_from_h(float h) {
  return 1.0f - __builtin_amdgcn_exp2f(h * (-100.0f * 1.44269504088896341f));
}

// Positional encoding of one 3-D point, all 39 entries + a zero pad, in the reference's order
// [x, sin(2^0 x), cos(2^0 x), ..., sin(2^5 x), cos(2^5 x)] (embedder.py:10-36).
struct PosEnc {
  float v[40];
  __device__ __forceinline__ void compute(float x0, float x1, float x2) {
    v[0] = x0; v[1] = x1; v[2] = x2;
    const float xs[3] = {x0, x1, x2};
#pragma unroll
    for (int f = 0; f < 6; ++f) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float s, co;
        sincosf(xs[c] * (float)(1 << f), &s, &co);
        v[3 + 6 * f + c] = s;
        v[6 + 6 * f + c] = co;
      }
    }
    v[39] = 0.0f;
  }
};

// Sample positions of one launch: first the ray samples cam + z * dir (n_ray = R*S of them, may be 0),
// then n_pts explicit points (may be 0).  P = n_ray + n_pts.
struct PointSrc {
  const float* pts;   // (n_pts,3) explicit points
  const float* cam;   // ray part: camera centre(s)
  const float* dirs;  // ray part: (R,3)
  const float* z;     // ray part: (R,S)
  int cam_stride;     // 0: one camera for all rays, 3: per ray
  int S;              // samples per ray
  int n_ray;          // R * S
  int P;              // total number of points
};

__device__ __forceinline__ void load_point(const PointSrc& src, int p, float& x0, float& x1, float& x2) {
  if (p >= src.P) p = src.P - 1;
  if (p >= src.n_ray) {
    const int q = p - src.n_ray;
    x0 = src.pts[3 * q + 0]; x1 = src.pts[3 * q + 1]; x2 = src.pts[3 * q + 2];
  } else {
    const int r = p / src.S;
    const float zz = src.z[p];
    const float* o = src.cam + (size_t)r * src.cam_stride;
    // cam_loc + z * ray_dir as two rounded float32 ops (network.py:226, ray_sampler.py:84)
    x0 = __fadd_rn(o[0], __fmul_rn(zz, src.dirs[3 * r + 0]));
    x1 = __fadd_rn(o[1], __fmul_rn(zz, src.dirs[3 * r + 1]));
    x2 = __fadd_rn(o[2], __fmul_rn(zz, src.dirs[3 * r + 2]));
  }
}

// ------------------------------------------------------------------------------------------------------
// one output tile:  acc(32 x 32 points) = hdr(bias) + sum_k A[:, k] x[k, :]
// ------------------------------------------------------------------------------------------------------
template <int KS>  // number of k-steps (2 input rows each); x holds KS/16 tiles (last may be partial)
__device__ __forceinline__ f32x16 tile_mma(const f32x4* __restrict__ chunk, const f32x16* x, int lane) {
  f32x16 acc;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const f32x4 b = chunk[r4 * 64 + lane];
    acc[4 * r4 + 0] = b[0]; acc[4 * r4 + 1] = b[1]; acc[4 * r4 + 2] = b[2]; acc[4 * r4 + 3] = b[3];
  }
  const f32x4* a_ptr = chunk + kHdrF4 + lane;
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
    const f32x4 a = a_ptr[s4 * 64];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = 4 * s4 + j;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], x[s / 16][s % 16], acc, 0, 0, 0);
    }
  }
  return acc;
}

// layer 0: the B operand of k-step s is PE row 2s (lanes 0-31) / 2s+1 (lanes 32-63)
__device__ __forceinline__ f32x16 tile_mma_pe(const f32x4* __restrict__ chunk, const PosEnc& pe, int lane, int half) {
  f32x16 acc;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const f32x4 b = chunk[r4 * 64 + lane];
    acc[4 * r4 + 0] = b[0]; acc[4 * r4 + 1] = b[1]; acc[4 * r4 + 2] = b[2]; acc[4 * r4 + 3] = b[3];
  }
  const f32x4* a_ptr = chunk + kHdrF4 + lane;
#pragma unroll
  for (int s4 = 0; s4 < 5; ++s4) {
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

// PE rows spliced into the layer-4 input (skip connection, network.py:80-81): accumulator rows 217..223
// carry PE[32..38] and rows 224..255 carry PE[0..31]; the 1/sqrt(2) is folded into the packed W4.
__device__ __forceinline__ void splice_skip(f32x16* y, const PosEnc& pe, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row0 = rho(r), row1 = rho(r) + 4;  // local rows for half 0 / 1
    y[7][r] = half ? pe.v[row1] : pe.v[row0];
    const int l0 = row0 - 25, l1 = row1 - 25;     // tile 6: local rows 25..31 -> PE[32..38]
    if (l0 >= 0 || l1 >= 0) {
      const float v0 = l0 >= 0 ? pe.v[32 + (l0 >= 0 ? l0 : 0)] : y[6][r];
      const float v1 = l1 >= 0 ? pe.v[32 + (l1 >= 0 ? l1 : 0)] : y[6][r];
      y[6][r] = half ? v1 : v0;
    }
  }
}

// A wave's 256 x 32 activation block in global memory: float4 index (i/4)*64 + lane holds accumulator
// registers i..i+3 (i = 16*tile + r) of that lane -- every wave-instruction moves 1 KiB contiguously.
__device__ __forceinline__ void store_tile_regs(float* __restrict__ dst, const f32x16* x, int lane) {
  f32x4* d = reinterpret_cast<f32x4*>(dst) + lane;
#pragma unroll
  for (int i4 = 0; i4 < 32; ++i4) {
    f32x4 v;
    v[0] = x[i4 / 4][4 * (i4 % 4) + 0]; v[1] = x[i4 / 4][4 * (i4 % 4) + 1];
    v[2] = x[i4 / 4][4 * (i4 % 4) + 2]; v[3] = x[i4 / 4][4 * (i4 % 4) + 3];
    d[i4 * 64] = v;
  }
}
__device__ __forceinline__ void load_tile_regs(const float* __restrict__ src, f32x16* x, int lane) {
  const f32x4* d = reinterpret_cast<const f32x4*>(src) + lane;
#pragma unroll
  for (int i4 = 0; i4 < 32; ++i4) {
    const f32x4 v = d[i4 * 64];
    x[i4 / 4][4 * (i4 % 4) + 0] = v[0]; x[i4 / 4][4 * (i4 % 4) + 1] = v[1];
    x[i4 / 4][4 * (i4 % 4) + 2] = v[2]; x[i4 / 4][4 * (i4 % 4) + 3] = v[3];
  }
}

// Multi-block activation buffers (hbuf, gbuf, ubuf, a2buf, abuf) are laid out [block][wave tile]: the blocks that all
// workgroups of a launch touch at the same time (same layer) are contiguous, and so are the blocks one weight-gradient
// job reads.  [wave tile][block] put concurrent accesses 256 KiB apart: 9 % less HBM throughput in a streaming
// micro-benchmark (tools/micro/layout_bw.hip: 5.0 vs 5.5 TB/s).  Stride between a tile's consecutive blocks:
__device__ __forceinline__ size_t block_stride() { return (size_t)gridDim.x * kWaves * kBlockF; }

__device__ __forceinline__ f32x16 load_tile(const float* __restrict__ block, int t, int lane) {
  const f32x4* d = reinterpret_cast<const f32x4*>(block) + lane;
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 f = SVS_STREAM_LOAD(d + (4 * t + q) * 64);
    v[4 * q] = f[0]; v[4 * q + 1] = f[1]; v[4 * q + 2] = f[2]; v[4 * q + 3] = f[3];
  }
  return v;
}
// quarter q (4 accumulator registers) of load_tile, for callers that spread the four loads over their MFMA gaps
__device__ __forceinline__ void load_tile_quarter(const float* __restrict__ block, int t, int lane, int q, f32x16& v) {
  const f32x4 f = SVS_STREAM_LOAD(reinterpret_cast<const f32x4*>(block) + lane + (4 * t + q) * 64);
  v[4 * q] = f[0]; v[4 * q + 1] = f[1]; v[4 * q + 2] = f[2]; v[4 * q + 3] = f[3];
}
__device__ __forceinline__ void store_tile(float* __restrict__ block, int t, int lane, const f32x16& v) {
  f32x4* d = reinterpret_cast<f32x4*>(block) + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 f; f[0] = v[4 * q]; f[1] = v[4 * q + 1]; f[2] = v[4 * q + 2]; f[3] = v[4 * q + 3];
    SVS_STREAM_STORE(f, d + (4 * t + q) * 64);   // streamed once: keep the weight stream in L2
  }
}

// zero the accumulator rows >= 217 of tile 6 (local rows 25..31): the PE splice, not network outputs
__device__ __forceinline__ void zero_splice_rows_tile6(f32x16& v, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool z0 = rho(r) >= 25, z1 = rho(r) + 4 >= 25;
    if (z0 || z1) { if (half ? z1 : z0) v[r] = 0.0f; }
  }
}

// Forward through layers 0..7; on return x holds h_8 (the input of lin8).  When HBUF, every layer's
// activations (h_1..h_8) are also stored to this wave's scratch tile (reverse pass / training backward).
template <bool HBUF>
__device__ __forceinline__ void forward_trunk(Stream& st, f32x16* x, f32x16* y, const PosEnc& pe, int lane, int half,
                                              float* __restrict__ hbuf) {
  // Stores of a finished tile are issued at the top of the NEXT tile (before its weight prefetch), so that the
  // vmcnt(0) of Stream::advance() never waits on a store that was issued a few cycles earlier.
  // ---- layer 0 : 39(40) -> 256
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (HBUF && t > 0) store_tile(hbuf, t - 1, lane, x[t - 1]);
    if (t < 7) st.prefetch<kChunk0F4>(); else st.prefetch<kChunkF4>();
    const f32x16 acc = tile_mma_pe(st.cur_buf(), pe, lane, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) x[t][r] = softplus100(acc[r]);
    st.advance();
  }
  if (HBUF) store_tile(hbuf, 7, lane, x[7]);
  // ---- layers 1..7 : 256 -> 256 (layer 3 emits 217 rows + the skip splice)
  for (int l = 1; l < 8; ++l) {
    float* hb = hbuf + (size_t)l * block_stride();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t == 7 && l == 3) break;  // lin3 has 217 outputs = 7 tiles; tile 7 is the PE splice
      if (HBUF && t > 0) store_tile(hb, t - 1, lane, y[t - 1]);
      st.prefetch<kChunkF4>();
      const f32x16 acc = tile_mma<128>(st.cur_buf(), x, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) y[t][r] = softplus100(acc[r]);
      st.advance();
    }
    if (l == 3) {
      splice_skip(y, pe, half);
      if (HBUF) { store_tile(hb, 6, lane, y[6]); store_tile(hb, 7, lane, y[7]); }   // tile 5 went out in the loop
    } else if (HBUF) {
      store_tile(hb, 7, lane, y[7]);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = y[t];
  }
}

// sdf = b8[0] + W8[0,:] . h8 from the VEC chunk (both halves end up with the full sum)
__device__ __forceinline__ float sdf_head(const f32x4* __restrict__ chunk, const f32x16* x, int lane) {
  float acc = 0.0f;
  const f32x4* w_ptr = chunk + kHdrF4 + lane;
#pragma unroll
  for (int s4 = 0; s4 < 32; ++s4) {
    const f32x4 w = w_ptr[s4 * 64];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = 4 * s4 + j;
      acc = __builtin_fmaf(w[j], x[s / 16][s % 16], acc);
    }
  }
  acc += __shfl_xor(acc, 32);
  return acc + chunk[lane][0];
}

}  // namespace mlp
}  // namespace svs
