// CasMVSNet cost-volume build for gfx950: fused homography warp + variance, 3-D U-Net convolutions with folded
// BatchNorm, and the softmax / depth-regression / confidence tail.
//
// Reference: models/CasMVSNet.py:280-315 (homo_warping), :601-663 (DepthNet.forward), :441-472 (CostRegNet),
// :107-186 (Conv3d / Deconv3d blocks), :519-595 and :733-751 (depth hypotheses).
//
// Warp + variance is HBM-bound: the reference materialises one (C,D,H,W) warped volume per source view and keeps
// three volumes live; here every output voxel is produced once -- sum and sum of squares stay in registers and
// only the variance is written (algorithmic bytes: write C*D*H*W*4 + read D*H*W*4 + V*C*H*W*4).
#include "svs_common.h"
#include <cstdlib>
#include "svs_split_volume.h"

namespace svs {
namespace costvol {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---- (C,H,W) -> (H,W,C): a source view's feature map in channel-last order, so that one bilinear corner is one
// contiguous C-vector -------------------------------------------------------------------------------------------
__global__ void chw_to_hwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)C * HW) return;
  const int c = (int)(idx % C);
  const size_t p = idx / C;
  out[idx] = in[(size_t)c * HW + p];
}

constexpr int kMaxSrc = 4;
struct WarpArgs {
  const float* ref;              // (C,H,W) reference-view feature
  const float* src_hwc[kMaxSrc]; // (H,W,C) source-view features
  float rot[kMaxSrc][9];         // src_proj @ inv(ref_proj), rows
  float trans[kMaxSrc][3];
  const float* depth_values;     // (D,H,W)
  float* variance;               // (C,D,H,W)
  int n_src, D, H, W;
  int raw_warp;                  // 1: write the warped volume of source 0 instead of the variance (homo_warping alone)
  uint4* split;                  // not null: the variance goes here as a split volume (svs_split_volume.h), not to `variance`
};

// One workgroup: kWarpPasses * (256 / (C/4)) consecutive x of one image row, `dz` depth planes.
//  * C/4 adjacent lanes own one voxel, 4 channels each: a bilinear corner ((H,W,C) source layout) is ONE contiguous
//    C-vector, so a wave's load instruction touches 64/(C/4) whole lines instead of 64 partial ones;
//  * corner weights and offsets are computed once per (voxel, source) by a separate phase and handed over in LDS
//    (otherwise every lane of a voxel repeats the projection and its divisions: the kernel was VALU-bound on that);
//  * the blend is packed fused multiply-adds (v_pk_fma_f32);
//  * the variance tile goes through LDS (aliased with the corner tables) so that every channel plane is written as
//    contiguous runs of x.
// What bounds it: the vector L1 delivers 64 B/clk/CU and every voxel pulls n_src * 4 corners * C * 4 B through it
// (4.0 GB at stage 1 = 0.12 ms) -- more than the HBM write of the volume (0.53 GB).
constexpr int kWarpPasses = 5;      // 5 * 32 = 160 = stage-1 width at C = 32 (and 320 / C = 16, 640 / C = 8)

template <int C, int NS>
__global__ __launch_bounds__(256, NS <= 2 ? 4 : 2) void warp_variance_kernel(WarpArgs a, int dz_planes) {
  constexpr int LPV = C / 4;                       // lanes per voxel
  constexpr int VPP = 256 / LPV;                   // voxels per pass
  constexpr int TW = kWarpPasses * VPP;            // tile width in x
  constexpr int S = TW + 4;                        // LDS row stride (floats): rows stay 16-byte aligned
  constexpr int kTileF = C * S, kTapF = NS * TW * 8;
  // the variance tile and the corner tables are never live together
  __shared__ __attribute__((aligned(16))) float lds[kTileF > kTapF ? kTileF : kTapF];
  float* tile = lds;
  f32x4* tapw = reinterpret_cast<f32x4*>(lds);               // per (source, voxel): the 4 corner weights (0: outside)
  i32x4* tapo = reinterpret_cast<i32x4*>(lds + NS * TW * 4); //                       and offsets into the (H,W,C) map
  const int tid = threadIdx.x;
  const int cg = tid % LPV, vl = tid / LPV;
  const int H = a.H, W = a.W, y = blockIdx.y;
  const int xt = blockIdx.x * TW;
  const size_t HW = (size_t)H * W;
  const float inv_nv = 1.0f / (float)(NS + 1);

  // reference-view feature of the tile, kept across the depth planes
  f32x4 ref[kWarpPasses];
#pragma unroll
  for (int p = 0; p < kWarpPasses; ++p) {
    const int x = xt + p * VPP + vl;
#pragma unroll
    for (int j = 0; j < 4; ++j) ref[p][j] = (x < W && !a.raw_warp) ? a.ref[(size_t)(4 * cg + j) * HW + (size_t)y * W + x] : 0.0f;
  }
  for (int dz = 0; dz < dz_planes; ++dz) {
    const int d = blockIdx.z * dz_planes + dz;
    if (d >= a.D) break;
    f32x4 r[kWarpPasses];
    // ---- phase A: corner weights and offsets, one (voxel, source) per thread --------------------------------------------
    __syncthreads();
    for (int i = tid; i < NS * TW; i += 256) {
      const int v = i / TW, vx = i - v * TW;
      const int x = xt + vx;
      f32x4 w4 = {0.0f, 0.0f, 0.0f, 0.0f};
      i32x4 o4 = {0, 0, 0, 0};
      if (x < W) {
        const float depth = a.depth_values[((size_t)d * H + y) * W + x];
        const float fx = (float)x, fy = (float)y;
        const float* R = a.rot[v];
        // rot @ [x,y,1] * depth + trans   (CasMVSNet.py:300-303)
        const float qx = ((R[0] * fx + R[1] * fy) + R[2]) * depth + a.trans[v][0];
        const float qy = ((R[3] * fx + R[4] * fy) + R[5]) * depth + a.trans[v][1];
        const float qz = ((R[6] * fx + R[7] * fy) + R[8]) * depth + a.trans[v][2];
        // correctly rounded divisions, the reference's operation order: at |coordinate| ~ 300 px one ulp of the
        // quotient already moves a sample by 3e-5 px (once per voxel and source: not what bounds the kernel)
        const float px = qx / qz, py = qy / qz;
        // normalised with the (W-1)/2 formula, sampled with align_corners=False (:305-312)
        const float gx = px / ((float)(W - 1) / 2.0f) - 1.0f, gy = py / ((float)(H - 1) / 2.0f) - 1.0f;
        const float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
        const float x0 = __builtin_floorf(ix), y0 = __builtin_floorf(iy);
        const float tx = ix - x0, ty = iy - y0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xx = x0 + (float)(k & 1), yy = y0 + (float)(k >> 1);
          // zeros padding: a corner outside contributes nothing (NaN coordinates compare false)
          if (xx >= 0.0f && xx <= (float)(W - 1) && yy >= 0.0f && yy <= (float)(H - 1)) {
            w4[k] = ((k & 1) ? tx : 1.0f - tx) * ((k >> 1) ? ty : 1.0f - ty);
            o4[k] = ((int)yy * W + (int)xx) * (C * 4);      // byte offset
          }
        }
      }
      tapw[i] = w4;
      tapo[i] = o4;
    }
    __syncthreads();
    // ---- phase B: gather the corners of all sources, 4 channels per lane ------------------------------------------------
#pragma unroll
    for (int p = 0; p < kWarpPasses; ++p) {
      const int vx = p * VPP + vl;
      f32x4 f[NS][4], w4[NS];
#pragma unroll
      for (int v = 0; v < NS; ++v) {
        w4[v] = tapw[v * TW + vx];
        const i32x4 o4 = tapo[v * TW + vx];
        const char* __restrict__ src = reinterpret_cast<const char*>(a.src_hwc[v]);   // uniform base + 32-bit offset
#pragma unroll
        for (int k = 0; k < 4; ++k) f[v][k] = *reinterpret_cast<const f32x4*>(src + ((unsigned)o4[k] + 16u * cg));
      }
      f32x4 sum = ref[p], sq = ref[p] * ref[p];
#pragma unroll
      for (int v = 0; v < NS; ++v) {
        // a corner outside has weight 0 and adds +-0: the same bits as skipping it
        f32x4 warped = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          warped = __builtin_elementwise_fma(f32x4{w4[v][k], w4[v][k], w4[v][k], w4[v][k]}, f[v][k], warped);
        sum += warped; sq = __builtin_elementwise_fma(warped, warped, sq);
        if (a.raw_warp) sum = warped;
      }
      if (a.raw_warp) r[p] = sum;
      else { const f32x4 m = sum * inv_nv; r[p] = sq * inv_nv - m * m; }
    }
    if (a.split) {
      // ---- phase C': fp16 hi / mid pieces straight from the registers: the two lanes that hold the 8 channels of a
      // unit write its halves, a wave's store covers 64 / LPV consecutive voxels of every channel group
      const int Hp = splitvol::padded_h(H), Wp = splitvol::padded_w(W);
#pragma unroll
      for (int p = 0; p < kWarpPasses; ++p) {
        const int x = xt + p * VPP + vl;
        if (x >= W) continue;
        f16x4 h, m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const _Float16 hh = (_Float16)r[p][j];
          h[j] = hh;
          m[j] = (_Float16)(r[p][j] - (float)hh);
        }
        uint2* u = reinterpret_cast<uint2*>(a.split + splitvol::unit(d, y, 0, cg >> 1, x, C / 8, Hp, Wp)) + (cg & 1);
        u[0] = __builtin_bit_cast(uint2, h);
        u[(size_t)(C / 8) * Wp * 2] = __builtin_bit_cast(uint2, m);
      }
      continue;
    }
    // ---- phase C: variance -> LDS tile -> contiguous runs of every channel plane ----------------------------------------
    __syncthreads();
#pragma unroll
    for (int p = 0; p < kWarpPasses; ++p) {
      const int vx = p * VPP + vl;
      // column swizzle (bit 3 by bit 1 of cg): with the 16-byte-aligned stride the 64 lanes of a wave would otherwise
      // fall on 16 banks; whole groups of 4 columns move together, so rows still read back as float4
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[(4 * cg + j) * S + (vx ^ ((cg & 2) << 2))] = r[p][j];
    }
    __syncthreads();
    const size_t cs = (size_t)a.D * HW;
    const size_t row = ((size_t)d * H + y) * W + xt;
    if ((W & 3) == 0) {
      for (int i = tid; i < C * (TW / 4); i += 256) {
        const int c = i / (TW / 4), v4 = (i - c * (TW / 4)) * 4;
        const int vx = v4 ^ (((c >> 2) & 2) << 2);
        if (xt + vx < W) *reinterpret_cast<f32x4*>(a.variance + c * cs + row + vx) = *reinterpret_cast<const f32x4*>(tile + c * S + v4);
      }
    } else {
      for (int i = tid; i < C * TW; i += 256) {
        const int c = i / TW, vx = i - c * TW;
        if (xt + vx < W) a.variance[c * cs + row + vx] = tile[c * S + (vx ^ (((c >> 2) & 2) << 2))];
      }
    }
  }
}

// ---- the split-volume producer with corner reuse along the depth axis ---------------------------------------------------
// Neighbouring depth planes move a voxel's sampling position in the source view by a fraction of a pixel (at config 3: the four
// bilinear corners of 3 voxels out of 4 are those of the plane before, at every stage; on DTU geometry the step is ~0.4 px at
// stage 1 and less afterwards).  The texture path is what bounds the kernel above (TA 73 % busy), so this variant keeps a
// voxel's corners in registers while it walks kWarpDz planes and gathers again only where the integer position has changed:
// the depth loop is the INNER loop (passes over x outside), the corner tables of all planes of the workgroup are computed up
// front.  Output: split volume only (direct stores; the float32 form needs a plane's whole tile at once and stays on the
// kernel above).
#ifndef SVS_WARP_DZ
#define SVS_WARP_DZ 4
#endif
// Diagnostic builds (tools/dev/ablate_warp.sh; results are then wrong by construction): -DSVS_WARP_ABL=<mask>, 1: no stores of
// the volume, 2: no gathers (a voxel's corners are never re-fetched), 4: no projection arithmetic (constant corner tables).
#ifndef SVS_WARP_ABL
#define SVS_WARP_ABL 0
#endif
constexpr int kWarpDz = SVS_WARP_DZ;

// depth: the hypothesis of voxel (d, y, x), loaded by the caller (all of a thread's loads are requested before the first
// projection: a load per loop iteration in front of its ~100 dependent instructions was a latency chain, round 5)
template <int C>
__device__ __forceinline__ void warp_taps(const WarpArgs& a, int v, int x, int y, int d, float depth, f32x4& w4, i32x4& o4) {
  const int H = a.H, W = a.W;
  w4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  o4 = i32x4{0, 0, 0, 0};
  if (x >= W || d >= a.D) return;
#if SVS_WARP_ABL & 16
  depth = 500.0f + (float)d;
#endif
  const float fx = (float)x, fy = (float)y;
  const float* R = a.rot[v];
  // rot @ [x,y,1] * depth + trans   (CasMVSNet.py:300-303); same operations, same order as warp_variance_kernel
  const float qx = ((R[0] * fx + R[1] * fy) + R[2]) * depth + a.trans[v][0];
  const float qy = ((R[3] * fx + R[4] * fy) + R[5]) * depth + a.trans[v][1];
  const float qz = ((R[6] * fx + R[7] * fy) + R[8]) * depth + a.trans[v][2];
#if SVS_WARP_ABL & 8
  const float rz = __builtin_amdgcn_rcpf(qz);
  const float px = qx * rz, py = qy * rz;
  const float gx = px * __builtin_amdgcn_rcpf((float)(W - 1) / 2.0f) - 1.0f, gy = py * __builtin_amdgcn_rcpf((float)(H - 1) / 2.0f) - 1.0f;
#else
  const float px = qx / qz, py = qy / qz;
  const float gx = px / ((float)(W - 1) / 2.0f) - 1.0f, gy = py / ((float)(H - 1) / 2.0f) - 1.0f;
#endif
  const float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
  const float x0 = __builtin_floorf(ix), y0 = __builtin_floorf(iy);
  const float tx = ix - x0, ty = iy - y0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float xx = x0 + (float)(k & 1), yy = y0 + (float)(k >> 1);
    if (xx >= 0.0f && xx <= (float)(W - 1) && yy >= 0.0f && yy <= (float)(H - 1)) {
      w4[k] = ((k & 1) ? tx : 1.0f - tx) * ((k >> 1) ? ty : 1.0f - ty);
      o4[k] = ((int)yy * W + (int)xx) * (C * 4);      // byte offset
    }
  }
}

// passes over x per workgroup: the corner tables (kWarpDz x NS x TW x 32 B) stay at 40 KiB for two sources (four workgroups
// per CU): TW = 160 voxels at C = 32, 128 at C = 16 and 8
template <int C> constexpr int reuse_passes() { return C == 32 ? 5 : (C == 16 ? 2 : 1); }

template <int C, int NS>
__global__ __launch_bounds__(256, NS <= 2 ? 4 : 2) void warp_variance_reuse_kernel(WarpArgs a) {
  constexpr int kPasses = reuse_passes<C>();
  constexpr int LPV = C / 4, VPP = 256 / LPV, TW = kPasses * VPP;
  __shared__ __attribute__((aligned(16))) f32x4 tapw[kWarpDz][NS][TW];
  __shared__ __attribute__((aligned(16))) i32x4 tapo[kWarpDz][NS][TW];
  const int tid = threadIdx.x;
  const int cg = tid % LPV, vl = tid / LPV;
  const int H = a.H, W = a.W, y = blockIdx.y;
  const int xt = blockIdx.x * TW, d0 = blockIdx.z * kWarpDz;
  const size_t HW = (size_t)H * W;
  const float inv_nv = 1.0f / (float)(NS + 1);
  const int Hp = splitvol::padded_h(H), Wp = splitvol::padded_w(W);
  // ---- corner weights and offsets of all planes, one (plane, source, voxel) per thread and round; the depth hypotheses of
  // all rounds first
  constexpr int kRounds = (kWarpDz * NS * TW + 255) / 256;
  float dep[kRounds];
#pragma unroll
  for (int k = 0; k < kRounds; ++k) {
    const int i = tid + 256 * k;
    const int dz = i / (NS * TW), r = i - dz * (NS * TW);
    const int vx = r - (r / TW) * TW;
    const int x = xt + vx, d = d0 + dz;
    dep[k] = (i < kWarpDz * NS * TW && x < W && d < a.D) ? a.depth_values[((size_t)d * H + y) * W + x] : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < kRounds; ++k) {
    const int i = tid + 256 * k;
    if (i >= kWarpDz * NS * TW) break;
    const int dz = i / (NS * TW), r = i - dz * (NS * TW);
    const int v = r / TW, vx = r - v * TW;
    f32x4 w4; i32x4 o4;
#if SVS_WARP_ABL & 4
    w4 = f32x4{0.25f, 0.25f, 0.25f, 0.25f}; o4 = i32x4{0, C * 4, C * 4 * W, C * 4 * (W + 1)};
#else
    warp_taps<C>(a, v, xt + vx, y, d0 + dz, dep[k], w4, o4);
#endif
    tapw[dz][v][vx] = w4;
    tapo[dz][v][vx] = o4;
  }
  __syncthreads();
  // Loads and stores retire from vmcnt in issue order: a wait for a plane's gathers also waited for the previous plane's
  // stores to be acknowledged -- every plane paid a write latency (ablation, round 5: the stores cost 0.085 of the kernel's
  // 0.205 ms although nothing reads them).  The plane loop is therefore skewed by one plane: a plane's two stores are issued
  // BEHIND the table reads and (conditional) gathers of the next plane -- or of the next pass's first plane, which are
  // unconditional: a new voxel -- so that the wait in front of the next blend leaves exactly those stores in flight.
  f32x4 f[NS][4];
  i32x4 held[NS];
  f32x4 w4s[NS];
  f32x4 ref;
  auto first_gathers = [&](int pass) {
    const int vxn = pass * VPP + vl;
#pragma unroll
    for (int j = 0; j < 4; ++j) ref[j] = xt + vxn < W ? a.ref[(size_t)(4 * cg + j) * HW + (size_t)y * W + xt + vxn] : 0.0f;
#pragma unroll
    for (int v = 0; v < NS; ++v) {
      w4s[v] = tapw[0][v][vxn];
      const i32x4 o4 = tapo[0][v][vxn];
      const char* __restrict__ src = reinterpret_cast<const char*>(a.src_hwc[v]);
#pragma unroll
      for (int k = 0; k < 4; ++k) f[v][k] = *reinterpret_cast<const f32x4*>(src + ((unsigned)o4[k] + 16u * cg));
      held[v] = o4;
    }
  };
  first_gathers(0);
#pragma unroll 1
  for (int p = 0; p < kPasses; ++p) {
    const int vx = p * VPP + vl, x = xt + vx;
#pragma unroll
    for (int dz = 0; dz < kWarpDz; ++dz) {
      const int d = d0 + dz;
      f32x4 sum = ref, sq = ref * ref;
#pragma unroll
      for (int v = 0; v < NS; ++v) {
        f32x4 warped = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          warped = __builtin_elementwise_fma(f32x4{w4s[v][k], w4s[v][k], w4s[v][k], w4s[v][k]}, f[v][k], warped);
        sum += warped; sq = __builtin_elementwise_fma(warped, warped, sq);
      }
      const f32x4 m = sum * inv_nv;
      const f32x4 res = sq * inv_nv - m * m;
      f16x4 h, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const _Float16 hh = (_Float16)res[j];
        h[j] = hh;
        lo[j] = (_Float16)(res[j] - (float)hh);
      }
      // the next plane's corner weights and (conditional) gathers: all sources requested before this plane's stores
      if (dz + 1 < kWarpDz) {
#pragma unroll
        for (int v = 0; v < NS; ++v) {
          w4s[v] = tapw[dz + 1][v][vx];
          const i32x4 o4 = tapo[dz + 1][v][vx];
          const char* __restrict__ src = reinterpret_cast<const char*>(a.src_hwc[v]);
          // (a voxel's LPV lanes take the same branch; a corner outside the image has offset 0 and weight 0)
#if SVS_WARP_ABL & 32          // diagnostic: every plane gathers again (no reuse along depth; static load / store counts)
#pragma unroll
          for (int k = 0; k < 4; ++k) f[v][k] = *reinterpret_cast<const f32x4*>(src + ((unsigned)o4[k] + 16u * cg));
#else
          if (!(SVS_WARP_ABL & 2) &&
              (o4[0] != held[v][0] || o4[1] != held[v][1] || o4[2] != held[v][2] || o4[3] != held[v][3])) {
#pragma unroll
            for (int k = 0; k < 4; ++k) f[v][k] = *reinterpret_cast<const f32x4*>(src + ((unsigned)o4[k] + 16u * cg));
            held[v] = o4;
          }
#endif
        }
      } else if (p + 1 < kPasses) {
        first_gathers(p + 1);
      }
      if (x < W && d < a.D && !((SVS_WARP_ABL & 1) && res[0] != 1.2345e-30f)) {
        uint2* u = reinterpret_cast<uint2*>(a.split + splitvol::unit(d, y, 0, cg >> 1, x, C / 8, Hp, Wp)) + (cg & 1);
        u[0] = __builtin_bit_cast(uint2, h);
        u[(size_t)(C / 8) * Wp * 2] = __builtin_bit_cast(uint2, lo);
      }
    }
  }
}

// ---- round 6: the producer re-written around its instruction count ---------------------------------------------------------
// Per (voxel, plane) round 5's kernel executes ~1540 lane-instructions -- 2 x 250 for the two corner-table entries and 8 lanes x
// 130 in the blend loop, of which ~50 are the blend, the variance and the fp16 split; the rest is per-lane overhead (table
// reads, the moved-or-not test, 16 register copies around the conditional gathers, address arithmetic, waits).  This kernel:
//   * the conditional re-gather is ONE inline-asm statement per source: exec is narrowed INSIDE it and the corner registers are
//     tied operands of straight-line code (no copy of the old corners, no vmcnt(0) in front of one); all vector-memory operations
//     of the blend loop are inline assembly, the loop is skewed by one plane (a plane's two stores are issued BEHIND the next
//     plane's gathers; `s_waitcnt vmcnt(2)` in front of the next blend waits for the gathers only -- the count is static because
//     an active wave issues exactly two stores per stored plane, and waves without a voxel inside the image leave the pass);
//   * corner tables source-major (a wave's entries belong to one source: its camera block comes from SGPRs), the four corners
//     without branches, the four IEEE divisions as the compiler's own sequence without its scaling / fix-up instructions where
//     no operand needs them, the refined reciprocal shared (bit-identical quotients);
//   * CPL channels per lane: 4 (default: round 5's shape, 16 waves per CU) or 8 (one 16-byte unit per lane, half the per-lane
//     overhead, but 176-190 VGPRs: one 5-wave workgroup per CU -- measured slower, SVS_WARP_KERNEL=8).
// ~1060 lane-instructions per (voxel, plane) at CPL = 4.  Measured (NOTES/r06.md): -4 % / -11 % / -6 % at the three stage
// shapes against round 5's kernel -- a third fewer instructions buy a twentieth of the time, so the kernel is NOT bound by
// instruction issue (this round's hypothesis), nor by where its gathers hit (XCD experiment below).
// Same arithmetic per channel in the same order as warp_variance_kernel: the values are the float32 kernel's bit for bit
// (tests/test_gpu_costvol.py::test_warp_variance_split_volume).
// IEEE float32 quotients n0 / d and n1 / d as the compiler's own division sequence computes them (v_div_scale, v_rcp, the
// two Newton steps on the reciprocal and the quotient, v_div_fmas, v_div_fixup: LLVM's AMDGPU f32 fdiv lowering), with the
// scaling and fix-up instructions left out and the refined reciprocal shared by both quotients: when neither operand is
// scaled (|d| and |n| well inside the normal range: v_div_scale returns them unchanged, v_div_fmas is a plain fma,
// v_div_fixup passes the quotient through) these ARE that sequence's operations, operation by operation, so the quotients
// are the correctly rounded ones bit for bit.  Outside that range the ordinary division runs.  (A projection divides two
// coordinates by one depth and the two results by two constants: four divisions of ~12 instructions per table entry.)
__device__ __forceinline__ void div2_ieee(float n0, float n1, float d, float& q0, float& q1) {
  const float ad = __builtin_fabsf(d);
  if (ad > 0x1p-40f && ad < 0x1p40f && __builtin_fabsf(n0) < 0x1p40f && __builtin_fabsf(n1) < 0x1p40f) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float m0 = n0 * r, m1 = n1 * r;
    m0 = __builtin_fmaf(__builtin_fmaf(-d, m0, n0), r, m0);
    m1 = __builtin_fmaf(__builtin_fmaf(-d, m1, n1), r, m1);
    q0 = __builtin_fmaf(__builtin_fmaf(-d, m0, n0), r, m0);
    q1 = __builtin_fmaf(__builtin_fmaf(-d, m1, n1), r, m1);
  } else {
    q0 = n0 / d; q1 = n1 / d;
  }
}
// n / d for a constant d whose refined reciprocal r = refine(rcp(d)) the caller formed once (same sequence)
__device__ __forceinline__ float refined_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float div_const_ieee(float n, float d, float r) {
  if (__builtin_fabsf(n) < 0x1p40f) {
    float m = n * r;
    m = __builtin_fmaf(__builtin_fmaf(-d, m, n), r, m);
    return __builtin_fmaf(__builtin_fmaf(-d, m, n), r, m);
  }
  return n / d;
}

// warp_taps (above) for the eight-channel producer: the same operations in the same order, with the four divisions as
// above (hw = (W - 1) / 2, hh = (H - 1) / 2 and their refined reciprocals are formed once per thread) and the four corners
// without branches (compare, select): an entry costs ~130 instructions instead of ~245.
template <int C>
__device__ __forceinline__ void warp_taps8(const WarpArgs& a, int v, int x, int y, int d, float depth, float hw, float rhw,
                                           float hh, float rhh, f32x4& w4, i32x4& o4) {
  const int H = a.H, W = a.W;
  const float fx = (float)x, fy = (float)y;
  const float* R = a.rot[v];
  const float qx = ((R[0] * fx + R[1] * fy) + R[2]) * depth + a.trans[v][0];
  const float qy = ((R[3] * fx + R[4] * fy) + R[5]) * depth + a.trans[v][1];
  const float qz = ((R[6] * fx + R[7] * fy) + R[8]) * depth + a.trans[v][2];
  float px, py;
  div2_ieee(qx, qy, qz, px, py);
  const float gx = div_const_ieee(px, hw, rhw) - 1.0f, gy = div_const_ieee(py, hh, rhh) - 1.0f;
  const float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
  const float x0 = __builtin_floorf(ix), y0 = __builtin_floorf(iy);
  const float tx = ix - x0, ty = iy - y0;
  const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
  const float wmax = (float)(W - 1), hmax = (float)(H - 1);
  const bool live = x < W && d < a.D;
  // zeros padding: a corner outside contributes nothing (NaN coordinates compare false)
  const bool vx0 = live && x0 >= 0.0f && x0 <= wmax, vx1 = live && x1 >= 0.0f && x1 <= wmax;
  const bool vy0 = y0 >= 0.0f && y0 <= hmax, vy1 = y1 >= 0.0f && y1 <= hmax;
  const float ux = 1.0f - tx, uy = 1.0f - ty;
  w4[0] = (vx0 && vy0) ? ux * uy : 0.0f;
  w4[1] = (vx1 && vy0) ? tx * uy : 0.0f;
  w4[2] = (vx0 && vy1) ? ux * ty : 0.0f;
  w4[3] = (vx1 && vy1) ? tx * ty : 0.0f;
  const int xi0 = (int)x0, xi1 = (int)x1, r0 = (int)y0 * W, r1 = (int)y1 * W;
  o4[0] = (vx0 && vy0) ? (r0 + xi0) * (C * 4) : 0;
  o4[1] = (vx1 && vy0) ? (r0 + xi1) * (C * 4) : 0;
  o4[2] = (vx0 && vy1) ? (r1 + xi0) * (C * 4) : 0;
  o4[3] = (vx1 && vy1) ? (r1 + xi1) * (C * 4) : 0;
}

// Geometry of the producer for CPL channels per lane: T threads, P passes over x per workgroup (TW = P * T / (C / CPL) voxels).
//   CPL = 4 (default): 256 threads, TW = 160 / 128 / 128 at C = 32 / 16 / 8 -- round 5's shape: 16 waves per CU;
//   CPL = 8: 320 / 320 / 128 threads, TW = 160 / 160 / 128 -- half the per-lane overhead, but 176-190 VGPRs (64 of them
//            corners): one 5-wave workgroup per CU instead of four 4-wave ones, and the kernel turns latency-bound (measured
//            0.267 against 0.226 ms at stage 1, NOTES/r06.md): kept selectable (SVS_WARP_KERNEL=8), not the default.
template <int C, int CPL> struct WarpCfg {};
template <> struct WarpCfg<32, 4> { static constexpr int T = 256, P = 5; };
template <> struct WarpCfg<16, 4> { static constexpr int T = 256, P = 2; };
template <> struct WarpCfg<8, 4>  { static constexpr int T = 256, P = 1; };
template <> struct WarpCfg<32, 8> { static constexpr int T = 320, P = 2; };
template <> struct WarpCfg<16, 8> { static constexpr int T = 320, P = 1; };
template <> struct WarpCfg<8, 8>  { static constexpr int T = 128, P = 1; };

__device__ __forceinline__ void gload128(f32x4& v, unsigned off, const void* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void gload128_16(f32x4& v, unsigned off, const void* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(v) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void gload32(float& v, unsigned off, const void* base) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void gstore128(unsigned off, const i32x4& v, void* base) {
  asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(off), "v"(v), "s"(base) : "memory");
}
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gstore64(unsigned off, const i32x2& v, void* base) {
#if defined(SVS_WARP_NT_STORES) && SVS_WARP_NT_STORES == 1        // experiments: streaming / system-coherent store hints
  asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(off), "v"(v), "s"(base) : "memory");
#elif defined(SVS_WARP_NT_STORES) && SVS_WARP_NT_STORES == 2
  asm volatile("global_store_dwordx2 %0, %1, %2 sc0 sc1" :: "v"(off), "v"(v), "s"(base) : "memory");
#elif defined(SVS_WARP_NT_STORES) && SVS_WARP_NT_STORES == 3
  asm volatile("global_store_dwordx2 %0, %1, %2 sc1 nt" :: "v"(off), "v"(v), "s"(base) : "memory");
#else
  asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(off), "v"(v), "s"(base) : "memory");
#endif
}
// The conditional re-gather of one source's four corners (2 x 16 bytes each) for the lanes of `mask`: exec is narrowed INSIDE
// the statement and the corner registers are tied operands of straight-line code.  (As `if (moved) f = load(...)` in C++ the
// merge of old and new corners becomes a copy of all eight registers in front of the branch -- 72 v_mov per plane -- and, with
// loads the compiler can see, a vmcnt(0) in front of each copy.)  No lane moved: the loads are skipped, nothing is counted.
__device__ __forceinline__ void regather(f32x4 (&f)[4][2], const i32x4& o4, unsigned lane_off, const void* base,
                                         unsigned long long mask) {
  unsigned long long saved;
  const unsigned o0 = (unsigned)o4[0] + lane_off, o1 = (unsigned)o4[1] + lane_off, o2 = (unsigned)o4[2] + lane_off,
                 o3 = (unsigned)o4[3] + lane_off;
  asm volatile(
      "s_and_saveexec_b64 %[sv], %[mask]\n\t"
      "s_cbranch_execz 1f\n\t"
      "global_load_dwordx4 %[a0], %[o0], %[base]\n\t"
      "global_load_dwordx4 %[a1], %[o0], %[base] offset:16\n\t"
      "global_load_dwordx4 %[b0], %[o1], %[base]\n\t"
      "global_load_dwordx4 %[b1], %[o1], %[base] offset:16\n\t"
      "global_load_dwordx4 %[c0], %[o2], %[base]\n\t"
      "global_load_dwordx4 %[c1], %[o2], %[base] offset:16\n\t"
      "global_load_dwordx4 %[d0], %[o3], %[base]\n\t"
      "global_load_dwordx4 %[d1], %[o3], %[base] offset:16\n"
      "1:\n\t"
      "s_mov_b64 exec, %[sv]"
      : [a0] "+v"(f[0][0]), [a1] "+v"(f[0][1]), [b0] "+v"(f[1][0]), [b1] "+v"(f[1][1]), [c0] "+v"(f[2][0]), [c1] "+v"(f[2][1]),
        [d0] "+v"(f[3][0]), [d1] "+v"(f[3][1]), [sv] "=&s"(saved)
      : [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [base] "s"(base), [mask] "s"(mask)
      : "memory");
}

__device__ __forceinline__ void regather(f32x4 (&f)[4][1], const i32x4& o4, unsigned lane_off, const void* base,
                                         unsigned long long mask) {
  unsigned long long saved;
  const unsigned o0 = (unsigned)o4[0] + lane_off, o1 = (unsigned)o4[1] + lane_off, o2 = (unsigned)o4[2] + lane_off,
                 o3 = (unsigned)o4[3] + lane_off;
  asm volatile(
      "s_and_saveexec_b64 %[sv], %[mask]\n\t"
      "s_cbranch_execz 1f\n\t"
      "global_load_dwordx4 %[a0], %[o0], %[base]\n\t"
      "global_load_dwordx4 %[b0], %[o1], %[base]\n\t"
      "global_load_dwordx4 %[c0], %[o2], %[base]\n\t"
      "global_load_dwordx4 %[d0], %[o3], %[base]\n"
      "1:\n\t"
      "s_mov_b64 exec, %[sv]"
      : [a0] "+v"(f[0][0]), [b0] "+v"(f[1][0]), [c0] "+v"(f[2][0]), [d0] "+v"(f[3][0]), [sv] "=&s"(saved)
      : [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [base] "s"(base), [mask] "s"(mask)
      : "memory");
}

// `s_waitcnt vmcnt(N)` + a scheduling barrier: hipcc otherwise hoists register-only instructions above an asm wait
// (the loads are invisible to its own waitcnt pass; naming the loaded registers as operands of the wait instead makes the
// register allocator copy every one of them around it: 80 v_mov per plane)
template <int N>
__device__ __forceinline__ void wait_loads() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int C, int NS, int CPL>
__global__ __launch_bounds__((WarpCfg<C, CPL>::T), (CPL == 4 ? (NS <= 2 ? 4 : 2) : 1)) void warp_variance_reuse2_kernel(WarpArgs a) {
  constexpr int T = WarpCfg<C, CPL>::T, P = WarpCfg<C, CPL>::P;
  constexpr int LPV = C / CPL, VPP = T / LPV, TW = P * VPP, G = C / 8, Q = CPL / 4;
  constexpr int kSlice = kWarpDz * TW;              // table entries of one source
  static_assert(kSlice % 64 == 0, "a wave's table entries belong to one source");
  __shared__ __attribute__((aligned(16))) f32x4 tapw[kWarpDz][NS][TW];
  __shared__ __attribute__((aligned(16))) i32x4 tapo[kWarpDz][NS][TW];
  const int tid = threadIdx.x;
  const int H = a.H, W = a.W;
  // (Experiment, -DSVS_WARP_XCD_ROWS: workgroups are dealt to the 8 XCDs round-robin in launch order, so every XCD sees every
  // row and gathers from the WHOLE source feature maps; re-dealing the launch index (xcd_chunked) row-slowest gives XCD k
  // the rows [k H/8, (k+1) H/8).  Measured on one box, three alternations: 0.216-0.222 against 0.213-0.220 ms at stage 1,
  // 0.085 / 0.085 at stage 2, 0.057 against 0.061 at stage 3: the gathers are not what the kernel waits for.  Off.)
#ifndef SVS_WARP_XCD_ROWS
  const int y = blockIdx.y, xb = blockIdx.x, zb = blockIdx.z;
#else
  const unsigned gx = gridDim.x, gz = gridDim.z;
  const unsigned lin = xcd_chunked(blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z), gx * gridDim.y * gz);
  const int y = (int)(lin / (gx * gz));
  const unsigned rem = lin - (unsigned)y * (gx * gz);
  const int zb = (int)(rem / gx), xb = (int)(rem - (unsigned)zb * gx);
#endif
  const int xt = xb * TW, d0 = zb * kWarpDz;
  const float inv_nv = 1.0f / (float)(NS + 1);
  // ---- corner tables, one (source, plane, voxel) per thread and round, source-major: a wave's entries of a round belong to
  // ONE source (its rot / trans rows are scalars); the depth hypotheses of all rounds are requested first
  {
    constexpr int kRounds = (NS * kSlice + T - 1) / T;
    float dep[kRounds];
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int i = tid + T * k, v = i / kSlice, r = i - v * kSlice, dz = r / TW, vx = r - dz * TW;
      const int x = xt + vx, d = d0 + dz;
      dep[k] = (i < NS * kSlice && x < W && d < a.D) ? a.depth_values[((size_t)d * H + y) * W + x] : 0.0f;
    }
    const float hw = (float)(W - 1) / 2.0f, hh = (float)(H - 1) / 2.0f;
    const float rhw = refined_rcp(hw), rhh = refined_rcp(hh);
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int i = tid + T * k;
      if (i >= NS * kSlice) break;
      const int v = __builtin_amdgcn_readfirstlane(i / kSlice);
      const int r = i - v * kSlice, dz = r / TW, vx = r - dz * TW;
      f32x4 w4; i32x4 o4;
#ifdef SVS_WARP8_PLAIN_TAPS
      warp_taps<C>(a, v, xt + vx, y, d0 + dz, dep[k], w4, o4);
#else
      warp_taps8<C>(a, v, xt + vx, y, d0 + dz, dep[k], hw, rhw, hh, rhh, w4, o4);
#endif
#if SVS_WARP_ABL & 4            // diagnostic: constant corner tables (the projection arithmetic is dead code)
      w4 = f32x4{0.25f, 0.25f, 0.25f, 0.25f}; o4 = i32x4{0, C * 4, C * 4 * W, C * 4 * (W + 1)};
#endif
      tapw[dz][v][vx] = w4;
      tapo[dz][v][vx] = o4;
    }
  }
  __syncthreads();
  const int g = tid % LPV, vl = tid / LPV;
  const int Hp = splitvol::padded_h(H), Wp = splitvol::padded_w(W);
  const unsigned plane_b = (unsigned)((size_t)Hp * 2 * G * Wp * 16);     // bytes between two planes of the split volume
  const unsigned mid_b = (unsigned)(G * Wp * 16);                        // from a unit's hi piece to its mid piece
  const unsigned HW4 = (unsigned)(H * W) * 4u;
  const unsigned lane_off = (unsigned)(4 * CPL * g);                     // this lane's channels inside a corner's C-vector
  f32x4 f[NS][4][Q];
  float ref[CPL];
  i32x4 held[NS];
  f32x4 w4s[NS];
#pragma unroll
  for (int v = 0; v < NS; ++v)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < Q; ++q) f[v][k][q] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < CPL; ++j) ref[j] = 0.0f;
  // first plane of a pass: a new voxel, everything is fetched
  auto first_loads = [&](int vxn) {
    const int xc = xt + vxn < W ? xt + vxn : W - 1;                      // (lanes outside the image fetch a valid address)
    unsigned ro = ((unsigned)(CPL * g) * (unsigned)(H * W) + (unsigned)(y * W + xc)) * 4u;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { gload32(ref[j], ro, a.ref); ro += HW4; }
#pragma unroll
    for (int v = 0; v < NS; ++v) {
      w4s[v] = tapw[0][v][vxn];
      const i32x4 o4 = tapo[0][v][vxn];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gload128(f[v][k][0], (unsigned)o4[k] + lane_off, a.src_hwc[v]);
        if (Q == 2) gload128_16(f[v][k][Q - 1], (unsigned)o4[k] + lane_off, a.src_hwc[v]);
      }
      held[v] = o4;
    }
  };
  // Straight-line code from here on (P and kWarpDz are unrolled; the only run-time branches are uniform and merge no
  // registers): planes beyond D are computed like the others (their table entries have weight 0 and offset 0) and not stored.
  int prev_stored = 0;                   // uniform: did the plane before this one issue its two stores?
  first_loads(vl);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int vx = p * VPP + vl, x = xt + vx;
    // waves without a voxel inside the image leave here (x grows with p: they were not active before either)
    if (__builtin_amdgcn_readfirstlane(xt + p * VPP + (tid & ~63) / LPV) >= W) break;
    // CPL = 8: the lane owns unit g; CPL = 4: half (g & 1) of unit g >> 1 (`so` = the unit's hi piece)
    unsigned so = (unsigned)(splitvol::unit(d0, y, 0, CPL == 8 ? g : g >> 1, x < W ? x : W - 1, G, Hp, Wp) * 16);
#pragma unroll
    for (int dz = 0; dz < kWarpDz; ++dz) {
      const int d = d0 + dz;
#if defined(SVS_WARP_STORE128)
      constexpr int kStores = CPL == 8 ? 2 : 1;
#else
      constexpr int kStores = 2;                         // vector-memory stores an active wave issues per stored plane
#endif
      if (prev_stored) wait_loads<kStores>(); else wait_loads<0>();
      f32x4 res[Q];
#pragma unroll
      for (int hh = 0; hh < Q; ++hh) {
        const f32x4 rf = {ref[4 * hh], ref[4 * hh + 1], ref[4 * hh + 2], ref[4 * hh + 3]};
        f32x4 sum = rf, sq = rf * rf;
#pragma unroll
        for (int v = 0; v < NS; ++v) {
          f32x4 warped = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            warped = __builtin_elementwise_fma(f32x4{w4s[v][k], w4s[v][k], w4s[v][k], w4s[v][k]}, f[v][k][hh], warped);
          sum += warped; sq = __builtin_elementwise_fma(warped, warped, sq);
        }
        const f32x4 m = sum * inv_nv;
        res[hh] = sq * inv_nv - m * m;
      }
      f16x4 h[Q], lo[Q];
#pragma unroll
      for (int hh = 0; hh < Q; ++hh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const _Float16 q = (_Float16)res[hh][j];
          h[hh][j] = q;
          lo[hh][j] = (_Float16)(res[hh][j] - (float)q);
        }
      // the next plane's corner weights and (masked) gathers -- or the next pass's first plane -- in front of this plane's
      // stores
      if (dz + 1 < kWarpDz) {
#pragma unroll
        for (int v = 0; v < NS; ++v) {
          w4s[v] = tapw[dz + 1][v][vx];
          const i32x4 o4 = tapo[dz + 1][v][vx];
          // (a voxel's LPV lanes decide alike; a corner outside the image has offset 0 and weight 0)
          const bool moved = !(SVS_WARP_ABL & 2) &&       // (diagnostic 2: a voxel's corners are never fetched again)
                             (o4[0] != held[v][0] || o4[1] != held[v][1] || o4[2] != held[v][2] || o4[3] != held[v][3]);
          regather(f[v], o4, lane_off, a.src_hwc[v], __builtin_amdgcn_ballot_w64(moved));
          held[v] = o4;
        }
      } else if (p + 1 < P) {
        first_loads((p + 1) * VPP + vl);
      }
#if SVS_WARP_ABL & 1            // diagnostic: no stores of the volume (the guard keeps the values alive)
      const int stored = d < a.D && res[0][0] == 1.2345e-30f;
#else
      const int stored = d < a.D;                                        // uniform
#endif
      if (stored && x < W) {
        if (CPL == 8) {
          gstore128(so, __builtin_bit_cast(i32x4, __builtin_shufflevector(h[0], h[Q - 1], 0, 1, 2, 3, 4, 5, 6, 7)), a.split);
          gstore128(so + mid_b, __builtin_bit_cast(i32x4, __builtin_shufflevector(lo[0], lo[Q - 1], 0, 1, 2, 3, 4, 5, 6, 7)), a.split);
        } else {
#ifndef SVS_WARP_STORE128
          gstore64(so + 8u * (g & 1), __builtin_bit_cast(i32x2, h[0]), a.split);
          gstore64(so + 8u * (g & 1) + mid_b, __builtin_bit_cast(i32x2, lo[0]), a.split);
#else
          // (experiment, -DSVS_WARP_STORE128: the two lanes of a unit swap halves (DPP quad_perm [1,0,3,2]), the even lane stores
          // the whole hi unit, the odd lane the whole mid unit -- ONE 16-byte store per lane and plane instead of two 8-byte
          // ones.  Measured on one box, three alternations: 0.228-0.240 against 0.225-0.239 ms at stage 1, 0.090 / 0.088,
          // 0.063 / 0.062: the number of store instructions is not what the kernel waits for either.)
          const i32x2 hv = __builtin_bit_cast(i32x2, h[0]), lv = __builtin_bit_cast(i32x2, lo[0]);
          const i32x2 ph = {__builtin_amdgcn_mov_dpp(hv[0], 0xB1, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(hv[1], 0xB1, 0xF, 0xF, true)};
          const i32x2 pl = {__builtin_amdgcn_mov_dpp(lv[0], 0xB1, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lv[1], 0xB1, 0xF, 0xF, true)};
          const i32x4 u = (g & 1) ? i32x4{pl[0], pl[1], lv[0], lv[1]} : i32x4{hv[0], hv[1], ph[0], ph[1]};
          gstore128(so + ((g & 1) ? mid_b : 0u), u, a.split);
#endif
        }
      }
      prev_stored = stored;
      so += plane_b;
    }
  }
}

template <int C, int CPL>
static bool launch_warp_reuse2(const WarpArgs& a, hipStream_t s) {
  // 32-bit byte offsets into the split volume and the feature maps
  const size_t vol = (size_t)(a.D + 2) * splitvol::padded_h(a.H) * 2 * (C / 8) * splitvol::padded_w(a.W) * 16;
  if (vol >= (1ull << 32) || (size_t)a.H * a.W * C * 4 >= (1ull << 31)) return false;
  constexpr int T = WarpCfg<C, CPL>::T, tw = WarpCfg<C, CPL>::P * (T / (C / CPL));
  dim3 grid((a.W + tw - 1) / tw, a.H, (a.D + kWarpDz - 1) / kWarpDz), block(T);
  if (CPL == 8) {
    if (a.n_src == 1) warp_variance_reuse2_kernel<C, 1, CPL><<<grid, block, 0, s>>>(a);
    else if (a.n_src == 2) warp_variance_reuse2_kernel<C, 2, CPL><<<grid, block, 0, s>>>(a);
    else return false;
    return true;
  }
  switch (a.n_src) {
    case 1: warp_variance_reuse2_kernel<C, 1, 4><<<grid, block, 0, s>>>(a); break;
    case 2: warp_variance_reuse2_kernel<C, 2, 4><<<grid, block, 0, s>>>(a); break;
    case 3: warp_variance_reuse2_kernel<C, 3, 4><<<grid, block, 0, s>>>(a); break;
    default: warp_variance_reuse2_kernel<C, 4, 4><<<grid, block, 0, s>>>(a); break;
  }
  return true;
}
template <int CPL>
static bool launch_warp_reuse2_any(int C, const WarpArgs& a, hipStream_t s) {
  return C == 8 ? launch_warp_reuse2<8, CPL>(a, s) : (C == 16 ? launch_warp_reuse2<16, CPL>(a, s) : launch_warp_reuse2<32, CPL>(a, s));
}

template <int C>
static void launch_warp_reuse(const WarpArgs& a, hipStream_t s) {
  const int tw = reuse_passes<C>() * (256 / (C / 4));
  dim3 grid((a.W + tw - 1) / tw, a.H, (a.D + kWarpDz - 1) / kWarpDz), block(256);
  switch (a.n_src) {
    case 1: warp_variance_reuse_kernel<C, 1><<<grid, block, 0, s>>>(a); break;
    case 2: warp_variance_reuse_kernel<C, 2><<<grid, block, 0, s>>>(a); break;
    case 3: warp_variance_reuse_kernel<C, 3><<<grid, block, 0, s>>>(a); break;
    default: warp_variance_reuse_kernel<C, 4><<<grid, block, 0, s>>>(a); break;
  }
}

template <int C>
static void launch_warp(const WarpArgs& a, hipStream_t s) {
  const int tw = kWarpPasses * (256 / (C / 4));
  const int dz = a.D >= 64 ? 4 : 1;                // few planes: one per workgroup, so that the launch fills the chip
  dim3 grid((a.W + tw - 1) / tw, a.H, (a.D + dz - 1) / dz), block(256);
  switch (a.n_src) {
    case 1: warp_variance_kernel<C, 1><<<grid, block, 0, s>>>(a, dz); break;
    case 2: warp_variance_kernel<C, 2><<<grid, block, 0, s>>>(a, dz); break;
    case 3: warp_variance_kernel<C, 3><<<grid, block, 0, s>>>(a, dz); break;
    default: warp_variance_kernel<C, 4><<<grid, block, 0, s>>>(a, dz); break;
  }
}

// ---- 3x3x3 convolution, padding 1, stride 1/2, folded BN (scale in the weights, shift as bias), optional ReLU ----
// weights: [Cin][27][Cout] (tap = (kd*3+kh)*3+kw).  One thread: VX consecutive x outputs x CT output channels.
constexpr int kConvThreads = 256;
constexpr int kCinChunk = 8;

struct ConvArgs {
  const float* in;     // (Cin, Di, Hi, Wi)
  const float* w;      // [Cin][27][Cout]
  const float* bias;   // [Cout] or nullptr
  const float* skip;   // (Cout, Do, Ho, Wo) added AFTER the ReLU (CasMVSNet.py:468-470) or nullptr
  float* out;          // (Cout, Do, Ho, Wo)
  int Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, stride, relu;
};

template <int CT, int VX>
__global__ __launch_bounds__(kConvThreads) void conv3d_kernel(ConvArgs a) {
  __shared__ float wl[kCinChunk * 27 * CT];
  const int co0 = blockIdx.y * CT;
  const int wg = (a.Wo + VX - 1) / VX;
  const long long total = (long long)a.Do * a.Ho * wg;
  const long long t = (long long)blockIdx.x * kConvThreads + threadIdx.x;
  const bool live = t < total;
  const long long tt = live ? t : 0;
  const int xo0 = (int)(tt % wg) * VX;
  const int yo = (int)((tt / wg) % a.Ho), zo = (int)(tt / ((long long)wg * a.Ho));
  float acc[CT][VX];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int v = 0; v < VX; ++v) acc[c][v] = 0.0f;
  const int s = a.stride;
  constexpr int NX = (VX - 1) * 2 + 3;   // input columns touched (stride <= 2)
  for (int ci0 = 0; ci0 < a.Cin; ci0 += kCinChunk) {
    const int nci = a.Cin - ci0 < kCinChunk ? a.Cin - ci0 : kCinChunk;
    __syncthreads();
    for (int i = threadIdx.x; i < nci * 27 * CT; i += kConvThreads) {
      const int c = i % CT, rest = i / CT;
      wl[i] = (co0 + c < a.Cout) ? a.w[((size_t)(ci0 * 27 + rest)) * a.Cout + co0 + c] : 0.0f;
    }
    __syncthreads();
    if (!live) continue;
    for (int ci = 0; ci < nci; ++ci) {
      const float* inc = a.in + (size_t)(ci0 + ci) * a.Di * a.Hi * a.Wi;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int zi = zo * s + kd - 1;
        if (zi < 0 || zi >= a.Di) continue;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int yi = yo * s + kh - 1;
          if (yi < 0 || yi >= a.Hi) continue;
          const float* row = inc + ((size_t)zi * a.Hi + yi) * a.Wi;
          float xin[NX];
          const int xi0 = xo0 * s - 1;
#pragma unroll
          for (int j = 0; j < NX; ++j) {
            const int xi = xi0 + j;
            xin[j] = (j < (VX - 1) * s + 3 && xi >= 0 && xi < a.Wi) ? row[xi] : 0.0f;
          }
          const float* wrow = wl + (ci * 27 + (kd * 3 + kh) * 3) * CT;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
              const float wv = wrow[kw * CT + c];
#pragma unroll
              for (int v = 0; v < VX; ++v) {
                const float xv = s == 1 ? xin[v + kw] : xin[2 * v + kw];
                acc[c][v] = __builtin_fmaf(wv, xv, acc[c][v]);
              }
            }
          }
        }
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if (co0 + c >= a.Cout) break;
    const float b = a.bias ? a.bias[co0 + c] : 0.0f;
#pragma unroll
    for (int v = 0; v < VX; ++v) {
      const int xo = xo0 + v;
      if (xo >= a.Wo) break;
      const size_t o = (((size_t)(co0 + c) * a.Do + zo) * a.Ho + yo) * a.Wo + xo;
      float r = acc[c][v] + b;
      if (a.relu) r = __builtin_fmaxf(r, 0.0f);
      if (a.skip) r += a.skip[o];
      a.out[o] = r;
    }
  }
}

// ---- ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1): out[o] += in[(o+1-t)/2] * w[ci][co][t] for the taps
// t whose (o+1-t) is even and in range.  weights: [Cin][27][Cout] (same layout, tap order of the torch kernel).
template <int CT>
__global__ __launch_bounds__(kConvThreads) void deconv3d_kernel(ConvArgs a) {
  __shared__ float wl[kCinChunk * 27 * CT];
  const int co0 = blockIdx.y * CT;
  const long long total = (long long)a.Do * a.Ho * a.Wo;
  const long long t = (long long)blockIdx.x * kConvThreads + threadIdx.x;
  const bool live = t < total;
  const long long tt = live ? t : 0;
  const int xo = (int)(tt % a.Wo), yo = (int)((tt / a.Wo) % a.Ho), zo = (int)(tt / ((long long)a.Wo * a.Ho));
  float acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = 0.0f;
  // valid taps per dimension: o even -> t=1 (i=o/2); o odd -> t=0 (i=(o+1)/2) and t=2 (i=(o-1)/2)
  int tz[2], iz[2], nz = 0, ty_[2], iy_[2], ny = 0, tx_[2], ix_[2], nx = 0;
  auto taps = [](int o, int n_in, int* tt_, int* ii_, int& n) {
    n = 0;
    for (int k = 0; k < 3; ++k) {
      const int num = o + 1 - k;
      if (num >= 0 && (num & 1) == 0 && (num >> 1) < n_in) { tt_[n] = k; ii_[n] = num >> 1; ++n; }
    }
  };
  taps(zo, a.Di, tz, iz, nz); taps(yo, a.Hi, ty_, iy_, ny); taps(xo, a.Wi, tx_, ix_, nx);
  for (int ci0 = 0; ci0 < a.Cin; ci0 += kCinChunk) {
    const int nci = a.Cin - ci0 < kCinChunk ? a.Cin - ci0 : kCinChunk;
    __syncthreads();
    for (int i = threadIdx.x; i < nci * 27 * CT; i += kConvThreads) {
      const int c = i % CT, rest = i / CT;
      wl[i] = (co0 + c < a.Cout) ? a.w[((size_t)(ci0 * 27 + rest)) * a.Cout + co0 + c] : 0.0f;
    }
    __syncthreads();
    if (!live) continue;
    for (int ci = 0; ci < nci; ++ci) {
      const float* inc = a.in + (size_t)(ci0 + ci) * a.Di * a.Hi * a.Wi;
      for (int p = 0; p < nz; ++p)
        for (int q = 0; q < ny; ++q)
          for (int r = 0; r < nx; ++r) {
            const float xv = inc[((size_t)iz[p] * a.Hi + iy_[q]) * a.Wi + ix_[r]];
            const float* wrow = wl + (ci * 27 + (tz[p] * 3 + ty_[q]) * 3 + tx_[r]) * CT;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = __builtin_fmaf(wrow[c], xv, acc[c]);
          }
    }
  }
  if (!live) return;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if (co0 + c >= a.Cout) break;
    const size_t o = (((size_t)(co0 + c) * a.Do + zo) * a.Ho + yo) * a.Wo + xo;
    float r = acc[c] + (a.bias ? a.bias[co0 + c] : 0.0f);
    if (a.relu) r = __builtin_fmaxf(r, 0.0f);
    if (a.skip) r += a.skip[o];
    a.out[o] = r;
  }
}

// ---- 3x3x3 convolution to ONE output channel (the U-Net's final `prob` layer, CasMVSNet.py:458,471), float32 on
// the vector ALUs.  With one output channel a matrix-core tile is 1/16 full (0.097 ms at stage 1 on
// conv3d_mfma_kernel, staging and conversion included); 216 float32 FMAs per voxel are 25 us of VALU time for the
// whole volume.  A thread owns 4 x-positions of one row over a run of z: every loaded input value feeds up to 9 FMAs
// (3 kw x 3 kd), rows of neighbouring threads overlap in the L1.  Weights are wave-uniform (scalar operands of
// v_pk_fma_f32).  No LDS, no barrier.
struct C1Args {
  const float* in;     // (Cin, D, H, W)
  const float* w;      // [Cin][27] (Cout = 1)
  const float* bias;   // [1] or nullptr
  const float* skip;   // (D, H, W) or nullptr
  float* out;          // (D, H, W)
  int Cin, D, H, W, relu;
};
#ifndef SVS_C1_VZ
#define SVS_C1_VZ 4
#endif
constexpr int kC1VX = 4, kC1VZ = SVS_C1_VZ, kC1TX = 8, kC1TY = 32;

// A thread marches along z over kC1VZ output slices of its 4 x-positions with three accumulator sets in flight: input
// slice zi adds its kd = 2 / 1 / 0 taps to the outputs zi-1 / zi / zi+1, the oldest set is then complete.  Loads are
// unconditional (clamped addresses, the value replaced by 0 outside the volume): no divergent branches around them.
//
// conv3d_c1_flat_kernel (W a multiple of 4): the lanes of a wave are 64 CONSECUTIVE 4-wide strips of the flattened
// (y, x/4) plane, so a row of the window is one float4 load per lane (1 KiB contiguous per wave) and the two edge
// columns x0-1, x0+4 come from the neighbouring lanes (only lanes 0 and 63 load theirs).  The first version loaded
// the edge columns per lane: 27 load instructions per (channel, slice), 18 of them touching 8 lines for 256 useful
// bytes -- the vector L1 bound the kernel at 0.12 ms.
__global__ __launch_bounds__(256, 4) void conv3d_c1_flat_kernel(const float* __restrict__ in, const float* __restrict__ wgt,
                                                                 const float* __restrict__ bias, const float* __restrict__ skip,
                                                                 float* __restrict__ out, int Cin, int D, int H, int W, int relu) {
  // (separate __restrict__ parameters: with the stores of finished slices inside the loop, hipcc reads weights it cannot
  // prove unaliased through the vector memory path and waits for ALL outstanding loads -- the prefetched rows -- each time)
  struct { const float* in; const float* w; const float* bias; const float* skip; float* out; int Cin, relu; } a{in, wgt, bias, skip, out, Cin, relu};
  constexpr int VX = kC1VX, VZ = kC1VZ;
  const int S = W / VX;                                     // strips per row
  const int sidx = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = sidx < H * S;
  const int sc = live ? sidx : H * S - 1;
  const int y = sc / S, col = sc - y * S, x0 = col * VX, z0 = blockIdx.y * VZ;
  const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
  const float b = a.bias ? a.bias[0] : 0.0f;
  int yoff[3];
  bool yok[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int yi = y + kh - 1;
    yok[kh] = yi >= 0 && yi < H;
    yoff[kh] = (yi < 0 ? 0 : (yi >= H ? H - 1 : yi)) * W + x0;
  }
  const bool lok = col > 0, rok = col < S - 1;
  // lanes 0 and 63 have their neighbour in another wave: one more load per row fetches lane 0's left and lane 63's
  // right column (two addresses per wave-instruction: the lower half of the wave reads the one, the upper half the
  // other).  Unconditional, so that hipcc can count the loads in flight and wait with vmcnt(N) on the prefetch ring.
  int offL[3], offR[3];
  {
    const int first = blockIdx.x * 256 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    const int last = first + 63 < H * S ? first + 63 : H * S - 1;
    const int yL = first / S, cL = first - yL * S, yR = last / S, cR = last - yR * S;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int yl = yL + kh - 1, yr = yR + kh - 1;
      offL[kh] = (yl < 0 ? 0 : (yl >= H ? H - 1 : yl)) * W + (cL > 0 ? cL * VX - 1 : 0);
      offR[kh] = (yr < 0 ? 0 : (yr >= H ? H - 1 : yr)) * W + (cR < S - 1 ? cR * VX + VX : 0);
    }
  }
  int offE[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) offE[kh] = lane < 32 ? offL[kh] : offR[kh];
  float a0[VX], a1[VX], a2[VX];
#pragma unroll
  for (int v = 0; v < VX; ++v) a0[v] = a1[v] = a2[v] = 0.0f;
  const int nz = D - z0 < VZ ? D - z0 : VZ;
  // the window rows of one (slice, channel): fetched one iteration ahead of their FMAs (a thread's iterations are a
  // serial chain and there are only ~2-4 waves per SIMD to hide a load behind)
  struct Rows { f32x4 m[3]; float e[3]; };
  auto fetch = [&](int zz, int c, Rows& R) {
    const int zi = z0 + zz - 1;
    const float* __restrict__ inc = a.in + (size_t)(zi < 0 ? 0 : (zi >= D ? D - 1 : zi)) * HW + (size_t)c * DHW;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const float* row = inc + yoff[kh];
      R.m[kh] = *reinterpret_cast<const f32x4*>(row);
      R.e[kh] = inc[offE[kh]];
    }
  };
  // ring of kPF row sets, the loop unrolled by kPF so that every set has a fixed register home (a `cur = nxt` copy
  // would wait for the prefetched loads at the end of every iteration); Cin is a multiple of kPF (launcher)
  constexpr int kPF = 4;
  Rows ring[kPF];
#pragma unroll
  for (int j = 0; j < kPF - 1; ++j) fetch(j / a.Cin, j % a.Cin, ring[j]);     // Cin >= kPF: slice 0
  for (int zz = 0; zz < nz + 2; ++zz) {
    const int zi = z0 + zz - 1;
    const bool zok = zi >= 0 && zi < D;
    const bool use2 = zz >= 2, use1 = zz >= 1 && zz <= nz, use0 = zz < nz;
    for (int c0 = 0; c0 < a.Cin; c0 += kPF) {
#pragma unroll
      for (int j = 0; j < kPF; ++j) {
        const int c = c0 + j;
        {
          const int cn = c + kPF - 1;
          const bool wrap = cn >= a.Cin;
          fetch(wrap ? zz + 1 : zz, wrap ? cn - a.Cin : cn, ring[(j + kPF - 1) % kPF]);   // (past the end: clamped, unused)
        }
        const Rows& cur = ring[j];
        const float* __restrict__ wc = a.w + c * 27;
        float r[3][VX + 2];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const bool ok = zok && yok[kh];
          const f32x4 m = cur.m[kh];
          // neighbour lanes' columns by DPP wave shifts (lane 0 / 63 keep their loaded edge column).  Inline assembly:
          // with __builtin_amdgcn_update_dpp hipcc 7.2 fed m[0] to BOTH shifts.  s_nop: a DPP source written by the
          // preceding vector instruction needs two wait states, which the compiler does not insert for inline assembly.
          float l = cur.e[kh], rr = cur.e[kh];
          asm("s_nop 1\n\tv_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %1, %3 wave_shl:1 row_mask:0xf bank_mask:0xf"
                       : "+v"(l), "+v"(rr) : "v"(m[3]), "v"(m[0]));
          r[kh][0] = (ok && lok) ? l : 0.0f;
          r[kh][VX + 1] = (ok && rok) ? rr : 0.0f;
#pragma unroll
          for (int v = 0; v < VX; ++v) r[kh][1 + v] = ok ? m[v] : 0.0f;
        }
        auto taps = [&](int kd, float (&acc)[VX]) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const float wv = wc[(kd * 3 + kh) * 3 + kw];
#pragma unroll
              for (int v = 0; v < VX; ++v) acc[v] = __builtin_fmaf(wv, r[kh][v + kw], acc[v]);
            }
        };
        if (use2) taps(2, a0);
        if (use1) taps(1, a1);
        if (use0) taps(0, a2);
      }
    }
    if (use2 && live) {
      const size_t o = (size_t)(zi - 1) * HW + (size_t)y * W + x0;
      f32x4 res;
#pragma unroll
      for (int v = 0; v < VX; ++v) {
        float t = a0[v] + b;
        if (a.relu) t = __builtin_fmaxf(t, 0.0f);
        if (a.skip) t += a.skip[o + v];
        res[v] = t;
      }
      *reinterpret_cast<f32x4*>(a.out + o) = res;
    }
#pragma unroll
    for (int v = 0; v < VX; ++v) { a0[v] = a1[v]; a1[v] = a2[v]; a2[v] = 0.0f; }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256, 4) void conv3d_c1_kernel(C1Args a) {
  constexpr int VX = kC1VX, VZ = kC1VZ;
  const int tx = threadIdx.x % kC1TX, ty = threadIdx.x / kC1TX;
  const int x0 = (blockIdx.x * kC1TX + tx) * VX, y = blockIdx.y * kC1TY + ty, z0 = blockIdx.z * VZ;
  const int D = a.D, H = a.H, W = a.W;
  if (x0 >= W || y >= H) return;
  const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
  const float b = a.bias ? a.bias[0] : 0.0f;
  // per-row constants: clamped row offsets and validity of the three kh rows, clamped edge columns
  int yoff[3];
  bool yok[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int yi = y + kh - 1;
    yok[kh] = yi >= 0 && yi < H;
    yoff[kh] = (yi < 0 ? 0 : (yi >= H ? H - 1 : yi)) * W;
  }
  const bool lok = x0 > 0, rok = x0 + VX < W;
  const int xl = lok ? x0 - 1 : 0, xr = rok ? x0 + VX : W - 1;
  float a0[VX], a1[VX], a2[VX];
#pragma unroll
  for (int v = 0; v < VX; ++v) a0[v] = a1[v] = a2[v] = 0.0f;
  const int nz = D - z0 < VZ ? D - z0 : VZ;          // output slices of this thread
  for (int zz = 0; zz < nz + 2; ++zz) {
    const int zi = z0 + zz - 1;
    const bool zok = zi >= 0 && zi < D;
    const float* __restrict__ slice = a.in + (size_t)(zi < 0 ? 0 : (zi >= D ? D - 1 : zi)) * HW;
    const bool use2 = zz >= 2, use1 = zz >= 1 && zz <= nz, use0 = zz < nz;
#pragma unroll 2
    for (int c = 0; c < a.Cin; ++c) {
      const float* __restrict__ wc = a.w + c * 27;
      const float* __restrict__ inc = slice + (size_t)c * DHW;
      float r[3][VX + 2];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float* row = inc + yoff[kh];
        const bool ok = zok && yok[kh];
        const float l = row[xl], rr = row[xr];
        r[kh][0] = (ok && lok) ? l : 0.0f;
        r[kh][VX + 1] = (ok && rok) ? rr : 0.0f;
        if (VEC) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(row + x0);
#pragma unroll
          for (int v = 0; v < VX; ++v) r[kh][1 + v] = ok ? m[v] : 0.0f;
        } else {
#pragma unroll
          for (int v = 0; v < VX; ++v) {
            const float t = row[x0 + v < W ? x0 + v : W - 1];
            r[kh][1 + v] = (ok && x0 + v < W) ? t : 0.0f;
          }
        }
      }
      auto taps = [&](int kd, float (&acc)[VX]) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float wv = wc[(kd * 3 + kh) * 3 + kw];
#pragma unroll
            for (int v = 0; v < VX; ++v) acc[v] = __builtin_fmaf(wv, r[kh][v + kw], acc[v]);
          }
      };
      if (use2) taps(2, a0);
      if (use1) taps(1, a1);
      if (use0) taps(0, a2);
    }
    if (use2) {
      const size_t o = (size_t)(zi - 1) * HW + (size_t)y * W + x0;
      float res[VX];
#pragma unroll
      for (int v = 0; v < VX; ++v) {
        float t = a0[v] + b;
        if (a.relu) t = __builtin_fmaxf(t, 0.0f);
        if (a.skip && x0 + v < W) t += a.skip[o + v];
        res[v] = t;
      }
      if (VEC) *reinterpret_cast<f32x4*>(a.out + o) = f32x4{res[0], res[1], res[2], res[3]};
      else {
#pragma unroll
        for (int v = 0; v < VX; ++v) if (x0 + v < W) a.out[o + v] = res[v];
      }
    }
#pragma unroll
    for (int v = 0; v < VX; ++v) { a0[v] = a1[v]; a1[v] = a2[v]; a2[v] = 0.0f; }
  }
}

// ---- softmax over D, depth regression, photometric confidence (CasMVSNet.py:648-663) ----------------------------
// 256 threads = 256/DS consecutive pixels x DS interleaved depth slices (slice sl owns d = sl, sl + DS, ...): at stage 1
// there are only 128 x 160 pixels, one thread per pixel would leave most of the chip idle behind 3 x 192 serial loads.
// Partial maxima / sums meet in LDS and are combined in slice order (deterministic).
template <int DS>
__global__ __launch_bounds__(256) void prob_depth_conf_kernel(const float* __restrict__ reg, const float* __restrict__ depth_values,
                                                              int D, int HW, float* __restrict__ prob, float* __restrict__ depth,
                                                              float* __restrict__ conf, int* __restrict__ index) {
  constexpr int PX = 256 / DS;
  __shared__ float red[3][DS][PX];
  const int lp = threadIdx.x % PX, sl = threadIdx.x / PX;
  const int p = blockIdx.x * PX + lp;
  const bool live = p < HW;
  // A thread's share of the depth axis (<= kSoftN values: D = 192 at DS = 8) stays in registers across the three passes
  // (maximum, sum of exponentials, probabilities): `reg` is read once.  Deeper volumes fall back to re-reading it.
  constexpr int kSoftN = 24;
  const bool cached = D <= kSoftN * DS;
  float v[kSoftN];
  float m = -__builtin_inff();
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < kSoftN; ++i) {
        const int d = sl + i * DS;
        v[i] = d < D ? reg[(size_t)d * HW + p] : -__builtin_inff();
        m = __builtin_fmaxf(m, v[i]);
      }
    } else {
      for (int d = sl; d < D; d += DS) m = __builtin_fmaxf(m, reg[(size_t)d * HW + p]);
    }
  }
  if (DS > 1) {
    red[0][sl][lp] = m;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DS; ++k) m = __builtin_fmaxf(m, red[0][k][lp]);
  }
  float s = 0.0f;
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < kSoftN; ++i)
        if (sl + i * DS < D) { v[i] = __expf(v[i] - m); s += v[i]; }
    } else {
      for (int d = sl; d < D; d += DS) s += __expf(reg[(size_t)d * HW + p] - m);
    }
  }
  if (DS > 1) {
    red[1][sl][lp] = s;
    __syncthreads();
    s = 0.0f;
#pragma unroll
    for (int k = 0; k < DS; ++k) s += red[1][k][lp];
  }
  const float inv = 1.0f / s;
  float dep = 0.0f, idxf = 0.0f;
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < kSoftN; ++i) {
        const int d = sl + i * DS;
        if (d < D) {
          const float pr = v[i] * inv;
          prob[(size_t)d * HW + p] = pr;
          dep += pr * depth_values[(size_t)d * HW + p];
          idxf += pr * (float)d;
        }
      }
    } else {
      for (int d = sl; d < D; d += DS) {
        const float pr = __expf(reg[(size_t)d * HW + p] - m) * inv;
        prob[(size_t)d * HW + p] = pr;
        dep += pr * depth_values[(size_t)d * HW + p];
        idxf += pr * (float)d;
      }
    }
  }
  if (DS > 1) {
    __syncthreads();                          // red[0] is read by every slice above
    red[0][sl][lp] = dep; red[2][sl][lp] = idxf;
    __syncthreads();
    dep = 0.0f; idxf = 0.0f;
#pragma unroll
    for (int k = 0; k < DS; ++k) { dep += red[0][k][lp]; idxf += red[2][k][lp]; }
  }
  if (!live || sl != 0) return;
  int idx = (int)idxf;                       // .long() truncation
  idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
  float c = 0.0f;                            // p[idx-1] + p[idx] + p[idx+1] + p[idx+2], zero padded
  for (int k = idx - 1; k <= idx + 2; ++k)
    if (k >= 0 && k < D) c += __expf(reg[(size_t)k * HW + p] - m) * inv;
  depth[p] = dep;
  conf[p] = c;
  if (index) index[p] = idx;
}

// ---- depth hypotheses (CasMVSNet.py:519-595, 733-751) -------------------------------------------------------------
// stage 1: D planes from dmin..dmax (or inverse-depth), identical for every pixel.
// stage >= 2: previous depth bilinearly resized to the image (align_corners=False), +-(D/2)*pixel_interval around it,
// then the trilinear resize of :747-749 to (D, H/s, W/s), which for the integer scales used is evaluated directly.
struct HypoArgs {
  const float* prev_depth;   // (Hp,Wp) previous stage depth or nullptr
  int Hp, Wp, H_img, W_img, D, Hs, Ws;   // Hs = H_img/scale
  float dmin, dmax, pix_interval;
  int inverse;
  float* out;                // (D,Hs,Ws)
};

__device__ __forceinline__ void lin_src(int o, int n_in, int n_out, int& i0, int& i1, float& t) {
  const float scale = (float)n_in / (float)n_out;
  float src = ((float)o + 0.5f) * scale - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src; if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
  t = src - (float)i0;
}

__device__ __forceinline__ float upsampled_prev(const HypoArgs& a, int yi, int xi) {
  int y0, y1, x0, x1; float ty, tx;
  lin_src(yi, a.Hp, a.H_img, y0, y1, ty);
  lin_src(xi, a.Wp, a.W_img, x0, x1, tx);
  const float* p = a.prev_depth;
  const float top = (1.0f - tx) * p[y0 * a.Wp + x0] + tx * p[y0 * a.Wp + x1];
  const float bot = (1.0f - tx) * p[y1 * a.Wp + x0] + tx * p[y1 * a.Wp + x1];
  return (1.0f - ty) * top + ty * bot;
}

__global__ void depth_hypotheses_kernel(HypoArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (!a.prev_depth) {
    if (idx >= a.D * a.Hs * a.Ws) return;
    const int d = idx / (a.Ws * a.Hs);
    float v;
    if (a.inverse) {
      const float step = 1.0f / (float)(a.D - 1);
      const float t = d < a.D / 2 ? __builtin_fmaf(step, (float)d, 0.0f) : __builtin_fmaf(-step, (float)(a.D - 1 - d), 1.0f);
      v = 1.0f / (1.0f / a.dmin * (1.0f - t) + 1.0f / a.dmax * t);
    } else {
      v = a.dmin + (float)d * ((a.dmax - a.dmin) / (float)(a.D - 1));
    }
    a.out[idx] = v;
    return;
  }
  // trilinear resize (D,H_img,W_img) -> (D,Hs,Ws): depth axis is an identity, the spatial axes interpolate.  One thread per
  // pixel: the four corner hypotheses' start and step do not depend on the plane, so they are formed once and the thread
  // walks the D planes (per plane the expressions of the per-voxel form, value for value).
  if (idx >= a.Hs * a.Ws) return;
  const int x = idx % a.Ws, y = idx / a.Ws;
  int y0, y1, x0, x1; float ty, tx;
  lin_src(y, a.H_img, a.Hs, y0, y1, ty);
  lin_src(x, a.W_img, a.Ws, x0, x1, tx);
  float cmin[4], step[4];
  const int yi[4] = {y0, y0, y1, y1}, xi[4] = {x0, x1, x0, x1};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float cur = upsampled_prev(a, yi[c], xi[c]);
    const float half = (float)a.D / 2.0f * a.pix_interval;
    const float lo = cur - half, hi = cur + half;
    cmin[c] = lo;
    step[c] = (hi - lo) / (float)(a.D - 1);
  }
  const size_t plane = (size_t)a.Hs * a.Ws;
  for (int d = 0; d < a.D; ++d) {
    float h[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) h[c] = cmin[c] + (float)d * step[c];
    const float top = (1.0f - tx) * h[0] + tx * h[1];
    const float bot = (1.0f - tx) * h[2] + tx * h[3];
    a.out[d * plane + idx] = (1.0f - ty) * top + ty * bot;
  }
}

}  // namespace costvol
}  // namespace svs

using namespace svs;
using namespace svs::costvol;

extern "C" {

int svs_chw_to_hwc(const float* in, float* out, int C, int H, int W, void* hip_stream) {
  if (!in || !out || C < 1 || H < 1 || W < 1) { set_error("svs_chw_to_hwc: bad argument"); return SVS_EINVAL; }
  const size_t n = (size_t)C * H * W;
  chw_to_hwc_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)hip_stream>>>(in, out, C, H * W);
  return check_launch("svs_chw_to_hwc");
}

static int warp_variance_any(const float* ref_feature, const float* const* src_features_hwc, const float* rot_trans,
                             int n_src, int C, int D, int H, int W, const float* depth_values, float* variance,
                             void* split, int raw_warp, void* hip_stream) {
  if (!ref_feature || !src_features_hwc || !rot_trans || !depth_values || (!variance && !split)) {
    set_error("svs_warp_variance: null argument"); return SVS_EINVAL;
  }
  if (n_src < 1 || n_src > kMaxSrc || D < 1 || H < 2 || W < 2) { set_error("svs_warp_variance: bad sizes"); return SVS_ESHAPE; }
  WarpArgs a;
  a.ref = ref_feature; a.depth_values = depth_values; a.variance = variance; a.n_src = n_src; a.D = D; a.H = H; a.W = W;
  a.raw_warp = raw_warp; a.split = reinterpret_cast<uint4*>(split);
  for (int v = 0; v < n_src; ++v) {
    if (!src_features_hwc[v]) { set_error("svs_warp_variance: null source %d", v); return SVS_EINVAL; }
    a.src_hwc[v] = src_features_hwc[v];
    for (int k = 0; k < 9; ++k) a.rot[v][k] = rot_trans[12 * v + k];      // HOST array: 9 rot + 3 trans per source
    for (int k = 0; k < 3; ++k) a.trans[v][k] = rot_trans[12 * v + 9 + k];
  }
  if (C != 8 && C != 16 && C != 32) { set_error("svs_warp_variance: C must be 8, 16 or 32 (FeatureNet outputs)"); return SVS_ESHAPE; }
  hipStream_t s = (hipStream_t)hip_stream;
  static const char* no_reuse = getenv("SVS_WARP_REUSE_OFF");
  static const char* which = getenv("SVS_WARP_KERNEL");            // A/B: "8" | "5" (below)
  if (a.split && !raw_warp && !(no_reuse && no_reuse[0] == '1')) {
    // default: the round-6 producer at four channels per lane; "8": eight channels per lane; "5": round 5's kernel
    if (!(which && which[0] == '5')) {
      // (eight channels per lane only with up to two sources: 64 corner registers per source pair)
      const bool ok = (which && which[0] == '8' && n_src <= 2) ? launch_warp_reuse2_any<8>(C, a, s) : launch_warp_reuse2_any<4>(C, a, s);
      if (ok) return check_launch("svs_warp_variance_split");
    }
    if (C == 8) launch_warp_reuse<8>(a, s);
    else if (C == 16) launch_warp_reuse<16>(a, s);
    else launch_warp_reuse<32>(a, s);
    return check_launch("svs_warp_variance_split");
  }
  if (C == 8) launch_warp<8>(a, s);
  else if (C == 16) launch_warp<16>(a, s);
  else launch_warp<32>(a, s);
  return check_launch("svs_warp_variance");
}

int svs_warp_variance(const float* ref_feature, const float* const* src_features_hwc, const float* rot_trans, int n_src,
                      int C, int D, int H, int W, const float* depth_values, float* variance, int raw_warp,
                      void* hip_stream) {
  return warp_variance_any(ref_feature, src_features_hwc, rot_trans, n_src, C, D, H, W, depth_values, variance, nullptr,
                           raw_warp, hip_stream);
}

// the variance as a split volume (svs_split_volume_dims bytes, zero-filled by the caller before its first use): the
// form svs_conv3d_pair reads.  Same values as svs_warp_variance, each split into its fp16 hi and mid parts.
int svs_warp_variance_split(const float* ref_feature, const float* const* src_features_hwc, const float* rot_trans,
                            int n_src, int C, int D, int H, int W, const float* depth_values, void* split,
                            void* hip_stream) {
  return warp_variance_any(ref_feature, src_features_hwc, rot_trans, n_src, C, D, H, W, depth_values, nullptr, split, 0,
                           hip_stream);
}

// transposed != 0: ConvTranspose3d(k3,s2,p1,op1) (Do = 2*Di); else Conv3d(k3,p1,stride)
int svs_conv3d(const float* in, const float* weight, const float* bias, const float* skip, float* out, int Cin, int Cout,
               int Di, int Hi, int Wi, int stride, int transposed, int relu, void* hip_stream) {
  if (!in || !weight || !out || Cin < 1 || Cout < 1 || Di < 1 || Hi < 1 || Wi < 1 || (stride != 1 && stride != 2)) {
    set_error("svs_conv3d: bad argument"); return SVS_EINVAL;
  }
  ConvArgs a;
  a.in = in; a.w = weight; a.bias = bias; a.skip = skip; a.out = out; a.Cin = Cin; a.Cout = Cout;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.stride = stride; a.relu = relu;
  hipStream_t s = (hipStream_t)hip_stream;
  if (transposed) {
    a.Do = 2 * Di; a.Ho = 2 * Hi; a.Wo = 2 * Wi;
    const long long total = (long long)a.Do * a.Ho * a.Wo;
    dim3 grid((unsigned)((total + kConvThreads - 1) / kConvThreads), (Cout + 7) / 8);
    deconv3d_kernel<8><<<grid, kConvThreads, 0, s>>>(a);
  } else {
    a.Do = (Di - 1) / stride + 1; a.Ho = (Hi - 1) / stride + 1; a.Wo = (Wi - 1) / stride + 1;
    constexpr int VX = 4;
    const long long total = (long long)a.Do * a.Ho * ((a.Wo + VX - 1) / VX);
    const long long wgs16 = ((total + kConvThreads - 1) / kConvThreads) * ((Cout + 15) / 16);
    if (wgs16 < 768) {
      // small volume (the coarse U-Net levels): one output voxel x 8 channels per thread, so that the launch still
      // has a few hundred workgroups
      const long long total1 = (long long)a.Do * a.Ho * a.Wo;
      dim3 grid((unsigned)((total1 + kConvThreads - 1) / kConvThreads), (Cout + 7) / 8);
      conv3d_kernel<8, 1><<<grid, kConvThreads, 0, s>>>(a);
    } else if (Cout <= 8) {
      dim3 grid((unsigned)((total + kConvThreads - 1) / kConvThreads), (Cout + 7) / 8);
      conv3d_kernel<8, VX><<<grid, kConvThreads, 0, s>>>(a);
    } else {
      dim3 grid((unsigned)((total + kConvThreads - 1) / kConvThreads), (Cout + 15) / 16);
      conv3d_kernel<16, VX><<<grid, kConvThreads, 0, s>>>(a);
    }
  }
  return check_launch("svs_conv3d");
}

// 3x3x3, stride 1, padding 1 convolution to one output channel in float32 (fused multiply-adds): the U-Net's `prob`
// layer.  weight [Cin][27][1].
int svs_conv3d_c1(const float* in, const float* weight, const float* bias, const float* skip, float* out, int Cin, int D,
                  int H, int W, int relu, void* hip_stream) {
  if (!in || !weight || !out || Cin < 1 || D < 1 || H < 1 || W < 1) { set_error("svs_conv3d_c1: bad argument"); return SVS_EINVAL; }
  C1Args a{in, weight, bias, skip, out, Cin, D, H, W, relu};
  hipStream_t s = (hipStream_t)hip_stream;
  if (W % 4 == 0 && Cin % 4 == 0) {
    dim3 grid((H * (W / kC1VX) + 255) / 256, (D + kC1VZ - 1) / kC1VZ);
    conv3d_c1_flat_kernel<<<grid, 256, 0, s>>>(in, weight, bias, skip, out, Cin, D, H, W, relu);
  } else {
    dim3 grid((W + kC1VX * kC1TX - 1) / (kC1VX * kC1TX), (H + kC1TY - 1) / kC1TY, (D + kC1VZ - 1) / kC1VZ);
    conv3d_c1_kernel<false><<<grid, 256, 0, s>>>(a);
  }
  return check_launch("svs_conv3d_c1");
}

int svs_prob_depth_conf(const float* reg, const float* depth_values, int D, int H, int W, float* prob, float* depth,
                        float* conf, int* index, void* hip_stream) {
  if (!reg || !depth_values || !prob || !depth || !conf || D < 1 || H < 1 || W < 1) {
    set_error("svs_prob_depth_conf: bad argument"); return SVS_EINVAL;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  const int HW = H * W;
  if (D >= 64) prob_depth_conf_kernel<8><<<(HW + 31) / 32, 256, 0, s>>>(reg, depth_values, D, HW, prob, depth, conf, index);
  else if (D >= 16) prob_depth_conf_kernel<4><<<(HW + 63) / 64, 256, 0, s>>>(reg, depth_values, D, HW, prob, depth, conf, index);
  else prob_depth_conf_kernel<1><<<(HW + 255) / 256, 256, 0, s>>>(reg, depth_values, D, HW, prob, depth, conf, index);
  return check_launch("svs_prob_depth_conf");
}

int svs_depth_hypotheses(const float* prev_depth, int Hp, int Wp, int H_img, int W_img, int D, int scale, float dmin,
                         float dmax, float pix_interval, int inverse, float* out, void* hip_stream) {
  if (!out || D < 2 || scale < 1 || H_img % scale || W_img % scale || (prev_depth && (Hp < 1 || Wp < 1))) {
    set_error("svs_depth_hypotheses: bad argument"); return SVS_EINVAL;
  }
  HypoArgs a{prev_depth, Hp, Wp, H_img, W_img, D, H_img / scale, W_img / scale, dmin, dmax, pix_interval, inverse, out};
  const int n = (prev_depth ? 1 : D) * a.Hs * a.Ws;
  depth_hypotheses_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_depth_hypotheses");
}

}  // extern "C"
