// conv0 of the cost-regularisation U-Net (3x3x3, stride 1, padding 1, C -> 8, folded BN + ReLU; reference:
// models/CasMVSNet.py:107-131 Conv3d block, :441-472 CostRegNet) on the fp16 matrix cores, fed by the "split volume"
// the fused warp + variance kernel writes (svs_costvol.hip): channel-last fp16 hi / mid pieces with a zero border, so
// that this kernel does no conversion and no bounds test -- a depth slice of the input window is copied into LDS by
// LDS-DMA (global_load_lds_dwordx4) while the MFMAs of the previous slice run.
//
// What bounds the layer is the MFMA pipe at its power-limited clock (matrix-core busy x clock
// is ~1.05-1.15 GHz-equivalent), so the point of the design is to waste as few MFMA rows as possible: with 8 output
// channels a 16-row tile would be half empty.  Here the 16 rows of v_mfma_f32_16x16x32_f16 are 8 channels x 2
// neighbouring x positions (2n, 2n+1) and a column is that PAIR of positions; K runs over the 4 input columns
// 2n-1 .. 2n+2 the pair touches (x 3 kh x 3 kd x C channels), the weight matrix holding w[kw = t] in the rows of the
// even position and w[kw = t-1] in the rows of the odd one (zero where kw falls outside 0..2).  27 of 36 products
// are useful instead of 27 of 54.  fp16x2 split operands as everywhere else (hi*hi + hi*mid + mid*hi, float32
// accumulation in three separate chains: no dependent back-to-back MFMAs, and the small terms sum among themselves).
//
// Split volume (C = 8 G channels): [D+2][Hp][2 pieces][G][Wp] units of 16 B = 8 consecutive channels of one voxel and
// piece; voxel (z,y,x) lives at padded coordinates (z+1, y+1, x+1); Hp = 4 ceil(H/4) + 2, Wp = 32 ceil(W/32) + 4;
// everything outside the D x H x W interior is zero (svs_split_volume_dims, svs_split_volume_pack).
//
// A workgroup owns a 4 (y) x 32 (x) output window and marches along z with a four-slot ring of input slices (6 rows x
// 35 columns x C channels x 2 pieces, 26 KiB at C = 32); a wave computes one row: 16 column pairs = one tile.  The
// weight fragments stay in registers for the whole kernel.  At C = 32 that is 288 registers, which leaves one wave
// per SIMD with part of the weights parked in AGPRs (a copy in front of every MFMA that uses them: 0.245 ms); instead
// two waves share a row, each with half of the k-steps and half of the weights, and the upper half's partial sums
// cross through LDS once per slice (0.21 ms, matrix cores 52 % busy at 1.87 GHz: the power-limited ceiling).  LDS
// layout of a slice: [row][piece][g][35] units -- the copy is contiguous on both sides, and the odd row-segment
// length puts the two channel groups (or x taps) a ds_read_b128 lane group mixes on opposite 16-byte slot parities:
// conflict-free (SQ_LDS_BANK_CONFLICT = 0).
#include "svs_common.h"
#include "svs_split_volume.h"
#include <type_traits>

namespace svs {
namespace convpair {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kTY = 4, kTX = 32, kHY = kTY + 2;
constexpr int kRS = 35;                            // units per row segment: x0-1 .. x0+33 (one more than needed: odd)

using splitvol::padded_h;
using splitvol::padded_w;

struct Args {
  const uint4* in;      // split volume
  const uint4* wfrag;   // [KS][2 pieces][64 lanes] A fragments
  const float* bias;    // [Cout] or nullptr
  float* out;           // (Cout, D, H, W) float32
  int Cout, D, H, W, relu;
  int z_per_wg;
};

#ifndef CONVP_PD
#define CONVP_PD 3
#endif

// KSPLIT = 2: eight waves, waves w and w + 4 share output row w and each takes half of the k-steps (and keeps only
// that half of the weights: 144 registers instead of 288, so that two waves fit on a SIMD and nothing lives in AGPRs);
// the upper half's partial sums cross through LDS at the end of the step.
template <int CIN, int KSPLIT>
__global__ __launch_bounds__(256 * KSPLIT, 1) void conv3d_pair_kernel(Args a) {
  constexpr int G = CIN / 8;
  constexpr int KS = 9 * G;                       // k-steps of 32 = 4 (x tap, channel group) combinations
  constexpr int KSW = (KS + KSPLIT - 1) / KSPLIT; // k-steps of a wave
  constexpr int UNITS = kHY * 2 * G * kRS;        // 16-byte units per slice
  constexpr int SLICE = UNITS * 16;
  constexpr int NW = 4 * KSPLIT;                  // waves
  constexpr int NDMA = (UNITS + 64 * NW - 1) / (64 * NW);   // copy instructions per wave and slice
  constexpr int PD = CONVP_PD;                    // k-steps the B fragments are read ahead of their MFMAs
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4v* xchg = reinterpret_cast<f32x4v*>(smem + 4 * SLICE);     // [step parity][row][lane] partial sums (KSPLIT = 2)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = wave & 3, half = wave >> 2;
  const int n = lane & 15, kg = lane >> 4;
  // Workgroups are dealt to the 8 XCDs round-robin in launch order; neighbouring windows share halo rows and columns, so
  // XCD k takes a contiguous range of the (z chunk, y, x) window list and finds its neighbours' halos in its own L2.
  const unsigned wg = xcd_chunked(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int bx = wg % gridDim.x, by = (wg / gridDim.x) % gridDim.y, bz = wg / (gridDim.x * gridDim.y);
  const int x0 = bx * kTX, y0 = by * kTY;
  const int z_begin = bz * a.z_per_wg;
  const int z_end = z_begin + a.z_per_wg < a.D ? z_begin + a.z_per_wg : a.D;
  const int Hp = padded_h(a.H), Wp = padded_w(a.W);
  const size_t HW = (size_t)a.H * a.W, DHW = (size_t)a.D * HW;
  const size_t slice_units = (size_t)Hp * 2 * G * Wp;   // units per padded z slice of the volume

  // ---- weights: the A fragments of this wave's k-steps
  f16x8 wh[KSW], wm[KSW];
#pragma unroll
  for (int i = 0; i < KSW; ++i) {
    const int s = half * KSW + i;
    wh[i] = s < KS ? __builtin_bit_cast(f16x8, a.wfrag[(2 * s) * 64 + lane]) : (f16x8)(_Float16)0;
    wm[i] = s < KS ? __builtin_bit_cast(f16x8, a.wfrag[(2 * s + 1) * 64 + lane]) : (f16x8)(_Float16)0;
  }

  // ---- slice copy: wave w moves the 1-KiB pieces w, w + NW, ...; unit u of the slice = (row segment u / 35, column u % 35)
  unsigned goff[NDMA];
#pragma unroll
  for (int j = 0; j < NDMA; ++j) {
    const int u = (wave + NW * j) * 64 + lane;
    const int rs = u / kRS, x = u - rs * kRS;
    const int ly = rs / (2 * G), pg = rs - ly * (2 * G);
    goff[j] = (unsigned)((((y0 + ly) * 2 * G + pg) * Wp + x0 + x) * 16);
  }
  auto copy_slice = [&](int z, int slot) {       // volume slice z (unpadded index, -1 .. D) -> ring slot
    const uint4* gz = a.in + (size_t)(z + 1) * slice_units;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      const int piece = wave + NW * j;
      if ((j + 1) * 64 * NW <= UNITS || piece * 64 + lane < UNITS) {
        const unsigned lds_base =
            (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(smem + slot * SLICE + piece * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(goff[j]), "s"(gz), "s"(lds_base) : "memory");
      }
    }
  };

  // ---- B fragments: lane (n, kg) of k-step s reads 8 channels of input column 2n + t of row `row` + kh, slice kd
  constexpr int TL = G == 4 ? 0 : (G == 2 ? 1 : 2);            // log2 of the x taps one k-step spans
  const int t_l = kg >> (2 - TL), g_l = kg & (G - 1);
  const int lane_off = ((row * 2 * G + g_l) * kRS + 2 * n + t_l) * 16;

  const int yo = y0 + row;
  const int pos = kg >> 1, co0 = 4 * (kg & 1);
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = (a.bias && co0 + r < a.Cout) ? a.bias[co0 + r] : 0.0f;
  const int xo = x0 + 2 * n + pos;
  const bool live = half == 0 && yo < a.H && xo < a.W;
  float* orow = a.out + (size_t)co0 * DHW + (size_t)yo * a.W + xo;
  f32x4v res = (f32x4v)(0.0f);       // this wave's sums of the previous step
  auto finish = [&](int z) {         // output slice z from the sums of the step that has just been closed by a barrier
    if (!live) return;
    f32x4v v = res;
    if (KSPLIT == 2) v += xchg[((z & 1) * 4 + row) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float o = v[r] + bias[r];
      if (a.relu) o = __builtin_fmaxf(o, 0.0f);
      if (co0 + r < a.Cout) orow[(size_t)r * DHW + (size_t)z * HW] = o;
    }
  };

  // Ring: slice zz lives in slot (zz - z_begin + 1) & 3.  Step z reads slices z-1, z, z+1; the copy of slice z+2 is
  // issued at its top (into the slot slice z-2 left) and has the whole step to land; the stores of step z-1 follow it.
  copy_slice(z_begin - 1, 0);
  copy_slice(z_begin, 1);
  copy_slice(z_begin + 1, 2);
  for (int z = z_begin; z < z_end; ++z) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                 // everyone's part of slice z+1 has landed; nobody reads slice z-2 any more
    const int r0 = (z - z_begin) & 3;
#ifndef CONVP_NODMA
    if (z + 2 <= z_end) copy_slice(z + 2, (r0 + 3) & 3);
#endif
    if (z > z_begin) finish(z - 1);
    const unsigned char* base[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) base[kd] = smem + ((r0 + kd) & 3) * SLICE + lane_off;
    f32x4v acc[3];
    acc[0] = (f32x4v)(0.0f); acc[1] = (f32x4v)(0.0f); acc[2] = (f32x4v)(0.0f);
    auto ksteps = [&](auto half_c) {
      constexpr int S0 = decltype(half_c)::value * KSW;
      constexpr int N = KS - S0 < KSW ? KS - S0 : KSW;
      f16x8 bh[N], bm[N];
      auto frag = [&](int i, int piece) -> f16x8 {
        const int s = S0 + i;
        const int row9 = s / G, kd = row9 / 3, kh = row9 % 3;
        const int t_s = (s % G) * (4 / G);
#ifdef CONVP_NOLDS
        if (i >= PD) return piece ? bm[i - PD] : bh[i - PD];
#endif
        return *reinterpret_cast<const f16x8*>(base[kd] + (((kh * 2 + piece) * G) * kRS + t_s) * 16);
      };
#pragma unroll
      for (int i = 0; i < PD && i < N; ++i) { bh[i] = frag(i, 0); bm[i] = frag(i, 1); }
      // hipcc would otherwise sink every fragment read to just in front of its first MFMA (shortest live range) and
      // the wave would sit out the LDS latency at every k-step: the barriers pin the read-ahead distance
#pragma unroll
      for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        if (i + PD < N) { bh[i + PD] = frag(i + PD, 0); bm[i + PD] = frag(i + PD, 1); }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[i], bh[i], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], bm[i], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], bh[i], acc[2], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (KSPLIT == 1 || half == 0) ksteps(std::integral_constant<int, 0>{});
    else ksteps(std::integral_constant<int, KSPLIT - 1>{});
    // accumulator: row 4 kg + r = (position kg >> 1, channel 4 (kg & 1) + r), column n
    res = (acc[0] + acc[1]) + acc[2];
    if (KSPLIT == 2 && half == 1) xchg[((z & 1) * 4 + row) * 64 + lane] = res;
  }
  if (z_end > z_begin) {
    if (KSPLIT == 2) __syncthreads();
    finish(z_end - 1);
  }
}

template <int CIN, int KSPLIT>
int launch(const Args& a, hipStream_t s) {
  constexpr int lds = 4 * kHY * 2 * (CIN / 8) * kRS * 16 + (KSPLIT == 2 ? 2 * 4 * 64 * 16 : 0);
  static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_pair_kernel<CIN, KSPLIT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { set_error("svs_conv3d_pair: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  dim3 grid((a.W + kTX - 1) / kTX, (a.H + kTY - 1) / kTY, (a.D + a.z_per_wg - 1) / a.z_per_wg);
  conv3d_pair_kernel<CIN, KSPLIT><<<grid, 256 * KSPLIT, lds, s>>>(a);
  return check_launch("svs_conv3d_pair");
}

// ---- conv2 of the U-Net (2b -> 2b channels at half resolution, CasMVSNet.py:446,462) from a split volume: the same slice
// ring and LDS-DMA copy as above, but the 16 MFMA rows are 16 output channels (no pairing needed) and a column is one x
// position; a wave computes one output row as two 16-column tiles over shared weight fragments.  K = (kd, kh, kw, g) in
// that order, 4 combinations per k-step; 27 G / 4 k-steps (13.5 at C = 16: the second half of the last one is zero).
// Its input comes from conv1, whose epilogue writes the split form directly (svs_conv3d_s2c8 with split_out).
struct RowsArgs {
  const uint4* in;      // split volume of (Cin, D, H, W)
  const uint4* wfrag;   // [KS][2 pieces][64 lanes] A fragments
  const float* bias;
  float* out;           // (Cout, D, H, W) float32
  int Cout, D, H, W, relu;
  int z_per_wg;
};

template <int CIN>
__global__ __launch_bounds__(256, 2) void conv3d_rows_kernel(RowsArgs a) {
  constexpr int G = CIN / 8;
  constexpr int KS = (27 * G + 3) / 4;
  constexpr int UNITS = kHY * 2 * G * kRS;
  constexpr int SLICE = UNITS * 16;
  constexpr int NDMA = (UNITS + 255) / 256;
  constexpr int PD = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kg = lane >> 4;
  const int x0 = blockIdx.x * kTX, y0 = blockIdx.y * kTY;
  const int z_begin = blockIdx.z * a.z_per_wg;
  const int z_end = z_begin + a.z_per_wg < a.D ? z_begin + a.z_per_wg : a.D;
  const int Hp = padded_h(a.H), Wp = padded_w(a.W);
  const size_t HW = (size_t)a.H * a.W, DHW = (size_t)a.D * HW;
  const size_t slice_units = (size_t)Hp * 2 * G * Wp;

  f16x8 wh[KS], wm[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    wh[s] = __builtin_bit_cast(f16x8, a.wfrag[(2 * s) * 64 + lane]);
    wm[s] = __builtin_bit_cast(f16x8, a.wfrag[(2 * s + 1) * 64 + lane]);
  }
  unsigned goff[NDMA];
#pragma unroll
  for (int j = 0; j < NDMA; ++j) {
    const int u = (wave + 4 * j) * 64 + lane;
    const int rs = u / kRS, x = u - rs * kRS;
    const int ly = rs / (2 * G), pg = rs - ly * (2 * G);
    goff[j] = (unsigned)((((y0 + ly) * 2 * G + pg) * Wp + x0 + x) * 16);
  }
  auto copy_slice = [&](int z, int slot) {
    const uint4* gz = a.in + (size_t)(z + 1) * slice_units;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      const int piece = wave + 4 * j;
      if ((j + 1) * 256 <= UNITS || piece * 64 + lane < UNITS) {
        const unsigned lds_base =
            (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(smem + slot * SLICE + piece * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(goff[j]), "s"(gz), "s"(lds_base) : "memory");
      }
    }
  };
  // B fragment of k-step s, lane group kg: combination c = 4 s + kg = (tap c / G, channel group c % G); tap > 26: zero
  // weights (any valid address will do).  Per k-step: the slice (kd) and the byte offset of (row + kh, piece 0, g, n + kw).
  int bkd[KS], boff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int c = 4 * s + kg;
    int tap = c / G;
    const int g = c % G;
    if (tap > 26) tap = 26;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    bkd[s] = kd;
    boff[s] = ((((wave + kh) * 2) * G + g) * kRS + n + kw) * 16;
  }
  const int yo = y0 + wave;
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = (a.bias && 4 * kg + r < a.Cout) ? a.bias[4 * kg + r] : 0.0f;
  f32x4v res[2];
  res[0] = (f32x4v)(0.0f); res[1] = (f32x4v)(0.0f);
  auto finish = [&](int z) {
    if (yo >= a.H) return;
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
      const int xo = x0 + 16 * xt + n;
      if (xo >= a.W) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * kg + r;
        if (co >= a.Cout) continue;
        float o = res[xt][r] + bias[r];
        if (a.relu) o = __builtin_fmaxf(o, 0.0f);
        a.out[(size_t)co * DHW + (size_t)z * HW + (size_t)yo * a.W + xo] = o;
      }
    }
  };
  copy_slice(z_begin - 1, 0);
  copy_slice(z_begin, 1);
  copy_slice(z_begin + 1, 2);
  for (int z = z_begin; z < z_end; ++z) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int r0 = (z - z_begin) & 3;
    if (z + 2 <= z_end) copy_slice(z + 2, (r0 + 3) & 3);
    if (z > z_begin) finish(z - 1);
    int sb[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) sb[kd] = ((r0 + kd) & 3) * SLICE;
    constexpr int PIECE = G * kRS * 16;            // piece 1 (mid) of a row segment set follows piece 0 (hi)
    f32x4v acc[2][3];
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) { acc[xt][0] = (f32x4v)(0.0f); acc[xt][1] = (f32x4v)(0.0f); acc[xt][2] = (f32x4v)(0.0f); }
    f16x8 bh[KS][2], bm[KS][2];
    auto frag = [&](int s) {
      const unsigned char* p = smem + sb[bkd[s]] + boff[s];
#pragma unroll
      for (int xt = 0; xt < 2; ++xt) {
        bh[s][xt] = *reinterpret_cast<const f16x8*>(p + xt * 256);
        bm[s][xt] = *reinterpret_cast<const f16x8*>(p + PIECE + xt * 256);
      }
    };
#pragma unroll
    for (int s = 0; s < PD && s < KS; ++s) frag(s);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      if (s + PD < KS) frag(s + PD);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int xt = 0; xt < 2; ++xt) {
        acc[xt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[s], bh[s][xt], acc[xt][0], 0, 0, 0);
        acc[xt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bm[s][xt], acc[xt][1], 0, 0, 0);
        acc[xt][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh[s][xt], acc[xt][2], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) res[xt] = (acc[xt][0] + acc[xt][1]) + acc[xt][2];
  }
  if (z_end > z_begin) finish(z_end - 1);
}

// float32 channel-first volume -> split volume (interior only; the border is the caller's zero fill).  One thread per unit.
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ in, uint4* __restrict__ out, int C,
                                                         int D, int H, int W) {
  const int G = C / 8, Hp = padded_h(H), Wp = padded_w(W);
  const size_t total = (size_t)D * H * G * W;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const int g = (int)((i / W) % G);
  const int y = (int)((i / ((size_t)W * G)) % H);
  const int z = (int)(i / ((size_t)W * G * H));
  const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
  f16x8 h, m;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = in[(size_t)(8 * g + j) * DHW + (size_t)z * HW + (size_t)y * W + x];
    const _Float16 hh = (_Float16)v;
    h[j] = hh;
    m[j] = (_Float16)(v - (float)hh);
  }
  const size_t u = splitvol::unit(z, y, 0, g, x, G, Hp, Wp);
  out[u] = __builtin_bit_cast(uint4, h);
  out[u + (size_t)G * Wp] = __builtin_bit_cast(uint4, m);
}

}  // namespace convpair
}  // namespace svs

using namespace svs;

extern "C" {

// dims[0..1] = Hp, Wp of the split volume of a (C, D, H, W) volume; returns its size in bytes
size_t svs_split_volume_dims(int C, int D, int H, int W, int* dims) {
  const int Hp = convpair::padded_h(H), Wp = convpair::padded_w(W);
  if (dims) { dims[0] = Hp; dims[1] = Wp; }
  return (size_t)(D + 2) * Hp * 2 * (C / 8) * Wp * 16;
}

// in (C,D,H,W) float32 -> split (svs_split_volume_dims bytes, zero-filled by the caller before its FIRST use: only the
// interior is written).  The fused producer is svs_warp_variance_split; this one serves tests and other callers.
int svs_split_volume_pack(const float* in, void* split, int C, int D, int H, int W, void* hip_stream) {
  if (!in || !split || (C != 8 && C != 16 && C != 32) || D < 1 || H < 1 || W < 1) {
    set_error("svs_split_volume_pack: bad argument (C in {8,16,32})"); return SVS_EINVAL;
  }
  const size_t total = (size_t)D * H * (C / 8) * W;
  convpair::split_pack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)hip_stream>>>(
      in, reinterpret_cast<uint4*>(split), C, D, H, W);
  return check_launch("svs_split_volume_pack");
}

// bytes of the packed A fragments of a layer: [9 Cin / 8][2][64][16 B]
size_t svs_conv3d_pair_wfrag_bytes(int Cin) { return (size_t)(9 * Cin / 8) * 2 * 64 * 16; }

// out (Cout,D,H,W) = [relu](conv3d(split volume of (Cin,D,H,W), 3x3x3, stride 1, padding 1) + bias); Cin in {8,16,32},
// Cout <= 8.  wfrag: fp16 hi / mid A fragments, fragment [k-step s][piece][lane] = for row m = lane & 15 (output
// channel m & 7 at x position parity m >> 3) and j = 0..7 the weight of k = 32 s + 8 (lane >> 4) + j, where
// k = (((kd*3 + kh)*4 + t)*G + g)*8 + c8 (G = Cin/8) multiplies input column 2n-1+t, channel 8g + c8: the folded
// weight of tap (kd, kh, kw = t - (m >> 3)), zero if kw is outside 0..2 or m & 7 >= Cout.
int svs_conv3d_pair(const void* split, const void* wfrag, const float* bias, float* out, int Cin, int Cout, int D, int H,
                    int W, int relu, void* hip_stream) {
  if (!split || !wfrag || !out || Cout < 1 || Cout > 8 || D < 1 || H < 1 || W < 1 || (Cin != 8 && Cin != 16 && Cin != 32)) {
    set_error("svs_conv3d_pair: bad argument (Cin in {8,16,32}, Cout <= 8)"); return SVS_EINVAL;
  }
  convpair::Args a{reinterpret_cast<const uint4*>(split), reinterpret_cast<const uint4*>(wfrag), bias, out, Cout, D, H, W,
                   relu, 0};
  // z split: a whole number of rounds on the 256 CUs, each workgroup paying a two-slice prologue
  const int xy = ((W + convpair::kTX - 1) / convpair::kTX) * ((H + convpair::kTY - 1) / convpair::kTY);
  int best = 1;
  double best_cost = 1e30;
  for (int zs = 1; zs <= D && zs <= 64; ++zs) {
    const int zp = (D + zs - 1) / zs;
    const long long wgs = (long long)xy * ((D + zp - 1) / zp);
    const double rounds = (double)((wgs + 255) / 256);
    const double cost = rounds * (zp + 2);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = zs; }
  }
  a.z_per_wg = (D + best - 1) / best;
  hipStream_t s = (hipStream_t)hip_stream;
#ifndef CONVP_SPLIT32
#define CONVP_SPLIT32 2
#endif
  if (Cin == 8) return convpair::launch<8, 1>(a, s);
  if (Cin == 16) return convpair::launch<16, 1>(a, s);
  return convpair::launch<32, CONVP_SPLIT32>(a, s);
}

// bytes of the packed A fragments of svs_conv3d_rows: [ceil(27 Cin / 32)][2][64][16 B]
size_t svs_conv3d_rows_wfrag_bytes(int Cin) { return (size_t)((27 * (Cin / 8) + 3) / 4) * 2 * 64 * 16; }

// out (Cout,D,H,W) = [relu](conv3d(split volume of (Cin,D,H,W), 3x3x3, stride 1, padding 1) + bias); Cin = 16, Cout <= 16
// (conv2 of CostRegNet).  wfrag: fragment [k-step s][piece][lane]: row lane & 15 = output channel, k = 32 s + 8 (lane >> 4)
// + c8 with combination 4 s + (lane >> 4) = tap * (Cin/8) + g (tap = (kd*3+kh)*3+kw): the folded weight of that tap and
// input channel 8 g + c8; zero for tap > 26 and channels >= Cout.
int svs_conv3d_rows(const void* split, const void* wfrag, const float* bias, float* out, int Cin, int Cout, int D, int H,
                    int W, int relu, void* hip_stream) {
  if (!split || !wfrag || !out || Cout < 1 || Cout > 16 || D < 1 || H < 1 || W < 1 || Cin != 16) {
    set_error("svs_conv3d_rows: bad argument (Cin = 16, Cout <= 16)"); return SVS_EINVAL;
  }
  convpair::RowsArgs a{reinterpret_cast<const uint4*>(split), reinterpret_cast<const uint4*>(wfrag), bias, out, Cout, D, H, W,
                       relu, 0};
  const int xy = ((W + convpair::kTX - 1) / convpair::kTX) * ((H + convpair::kTY - 1) / convpair::kTY);
  int best = 1;
  double best_cost = 1e30;
  for (int zs = 1; zs <= D && zs <= 64; ++zs) {
    const int zp = (D + zs - 1) / zs;
    const long long wgs = (long long)xy * ((D + zp - 1) / zp);
    const double rounds = (double)((wgs + 511) / 512);          // two workgroups per CU
    const double cost = rounds * (zp + 2);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = zs; }
  }
  a.z_per_wg = (D + best - 1) / best;
  constexpr int lds = 4 * convpair::kHY * 2 * 2 * convpair::kRS * 16;
  static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(convpair::conv3d_rows_kernel<16>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { set_error("svs_conv3d_rows: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  dim3 grid((W + convpair::kTX - 1) / convpair::kTX, (H + convpair::kTY - 1) / convpair::kTY, (D + a.z_per_wg - 1) / a.z_per_wg);
  convpair::conv3d_rows_kernel<16><<<grid, 256, lds, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_conv3d_rows");
}

}  // extern "C"
