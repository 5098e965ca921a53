// 3x3x3 convolution (stride 1, padding 1, folded BN, optional ReLU / skip) of the cost-regularisation U-Net on the
// fp16 matrix cores with two-piece split operands (fp16x2, see svs_mlp_h2_dev.h): the layers that carry most of the
// U-Net's work -- conv0 (C -> 8 at full resolution: 68 / 52 / 35 % of the MACs of stage 1 / 2 / 3), conv2 (16 -> 16)
// and the final prob layer (8 -> 1).  Reference: models/CasMVSNet.py:107-131 (Conv3d block), :441-472 (CostRegNet).
//
// Implicit GEMM per output row segment: D[cout][voxel] = sum_k W[cout][k] * P[k][voxel], k = (tap, cin) with cin
// fastest.  v_mfma_f32_16x16x32_f16: M = 16 output channels (Cout <= 16), N = 16 consecutive x, K = 32.  The weights
// (A operand) of ALL k-steps live in registers for the whole kernel (<= 216 VGPRs); the input patches (B operand) are
// 16-byte channel vectors read from a four-slot z ring in LDS.  A workgroup (4 waves) owns a 4 (y) x 32 (x) output
// window and marches along z: per step it converts one new input slice (6 x 34 voxels x Cin, float32 channel-first in
// global memory) into channel-last fp16 hi / mid pieces in LDS while the MFMAs of the previous step run.
#include "svs_common.h"

namespace svs {
namespace convmfma {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kTY = 4, kTX = 32;               // output window of a workgroup
constexpr int kHY = kTY + 2, kHX = kTX + 2;    // with halo

struct Args {
  const float* in;      // (Cin, D, H, W)
  const uint4* wfrag;   // [KS][2 pieces][64 lanes] 16-byte A fragments (packed by the host)
  const float* bias;    // [Cout] or nullptr
  const float* skip;    // (Cout, D, H, W) added after the ReLU, or nullptr
  float* out;           // (Cout, D, H, W)
  int Cout, D, H, W, relu;
  int z_per_wg;         // z extent of a workgroup
};

template <int CIN>
__global__ __launch_bounds__(256, 1) void conv3d_mfma_kernel(Args a) {
  constexpr int KS = (27 * CIN + 31) / 32;                 // k-steps of 32
  constexpr int PITCH = CIN == 8 ? 16 : (CIN == 16 ? 48 : 80);   // bytes per voxel and piece: odd multiple of 16
  constexpr int PIECE = kHY * kHX * PITCH;                 // one piece of one slice
  constexpr int SLICE = 2 * PIECE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int vox = lane & 15, kg = lane >> 4;
  const int x0 = blockIdx.x * kTX, y0 = blockIdx.y * kTY;
  const int z_begin = blockIdx.z * a.z_per_wg;
  const int z_end = z_begin + a.z_per_wg < a.D ? z_begin + a.z_per_wg : a.D;
  const size_t HW = (size_t)a.H * a.W, DHW = (size_t)a.D * HW;

  // ---- weights: all A fragments of this lane, for the whole kernel
  f16x8 wh[KS], wm[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    wh[s] = __builtin_bit_cast(f16x8, a.wfrag[(2 * s) * 64 + lane]);
    wm[s] = __builtin_bit_cast(f16x8, a.wfrag[(2 * s + 1) * 64 + lane]);
  }
  // ---- the B-fragment address of k-step s without the slice base: (tap, channel group) of this lane's 8 elements
  // kk = 32 s + 8 kg: tap = kk / CIN, ci0 = kk % CIN
  int boff[KS], bkd[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int kk = 32 * s + 8 * kg;
    int tap = kk / CIN;
    const int ci0 = kk % CIN;
    if (tap > 26) tap = 26;                       // padded k: the weights are zero there
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    bkd[s] = kd;
    boff[s] = ((wave + kh) * kHX + (vox + kw)) * PITCH + 2 * ci0;
  }

  // ---- slice staging: thread -> one (y, x) column of the halo window, all channels
  const bool loader = tid < kHY * kHX;
  const int ly = tid / kHX, lx = tid % kHX;
  const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
  const bool in_xy = loader && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
  const float* gcol = a.in + (in_xy ? (size_t)gy * a.W + gx : 0);
  const int lcol = (ly * kHX + lx) * PITCH;
  float stage[CIN];
  auto fetch = [&](int z) {
    const bool ok = in_xy && z >= 0 && z < a.D;
#pragma unroll
    for (int c = 0; c < CIN; ++c) stage[c] = ok ? gcol[(size_t)c * DHW + (size_t)z * HW] : 0.0f;
  };
  // one group of 8 channels of the staged column -> hi / mid pieces in ring slot `slot`
  auto commit_group = [&](int slot, int c8) {
    if (!loader) return;
    unsigned char* ph = smem + slot * SLICE + lcol + 16 * c8;
    f16x8 h, m;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = stage[8 * c8 + j];
      const _Float16 hh = (_Float16)v;
      h[j] = hh;
      m[j] = (_Float16)(v - (float)hh);
    }
    *reinterpret_cast<f16x8*>(ph) = h;
    *reinterpret_cast<f16x8*>(ph + PIECE) = m;
  };
  auto commit = [&](int slot) {
#pragma unroll
    for (int c8 = 0; c8 < CIN / 8; ++c8) commit_group(slot, c8);
  };

  // Four-slot z ring: slice zz lives in slot (zz - z_begin + 1) & 3.  Step z reads slices z-1, z, z+1 while the
  // staged slice z+2 is converted into the fourth slot between the MFMAs of its first k-steps; right after that
  // the loads of slice z+3 are issued and stay in flight until the next step.  One barrier per step.
  fetch(z_begin - 1); commit(0);
  fetch(z_begin); commit(1);
  fetch(z_begin + 1); commit(2);
  fetch(z_begin + 2);
  const int yo = y0 + wave;
  __syncthreads();
  for (int z = z_begin; z < z_end; ++z) {
    const int r0 = (z - z_begin) & 3;
    const int base[3] = {r0 * SLICE, ((r0 + 1) & 3) * SLICE, ((r0 + 2) & 3) * SLICE};
    const int slot_new = (r0 + 3) & 3;
    f32x4v acc[2];
    acc[0] = (f32x4v)(0.0f); acc[1] = (f32x4v)(0.0f);
    // two-stage pipeline: the fragments of k-step s+1 are read behind the first MFMA of k-step s (one wave per SIMD:
    // nobody else hides the LDS latency; hipcc waits with lgkmcnt(0) before a fragment's first use)
    f16x8 bh[2], bm[2];
    {
      const unsigned char* p = smem + base[bkd[0]] + boff[0];
#pragma unroll
      for (int xt = 0; xt < 2; ++xt) {
        bh[xt] = *reinterpret_cast<const f16x8*>(p + xt * 16 * PITCH);
        bm[xt] = *reinterpret_cast<const f16x8*>(p + PIECE + xt * 16 * PITCH);
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f16x8 nh[2], nm[2];
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[s], bh[0], acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#ifdef CONVX_NOLDS
      nh[0] = bh[0]; nh[1] = bh[1]; nm[0] = bm[0]; nm[1] = bm[1];
#else
      if (s + 1 < KS) {
        const unsigned char* p = smem + base[bkd[s + 1]] + boff[s + 1];
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
          nh[xt] = *reinterpret_cast<const f16x8*>(p + xt * 16 * PITCH);
          nm[xt] = *reinterpret_cast<const f16x8*>(p + PIECE + xt * 16 * PITCH);
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[s], bh[1], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bm[0], acc[0], 0, 0, 0);
#ifndef CONVX_NOFETCH
      if (s < CIN / 8) commit_group(slot_new, s);         // staged slice z+2, hidden behind the MFMAs
      if (s == CIN / 8) fetch(z + 3);                     // its registers are free again
#endif
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bm[1], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh[1], acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < KS) {
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) { bh[xt] = nh[xt]; bm[xt] = nm[xt]; }
      }
    }
    // accumulator: row = 4 kg + r (output channel), column = vox
    if (yo < a.H) {
#pragma unroll
      for (int xt = 0; xt < 2; ++xt) {
        const int xo = x0 + 16 * xt + vox;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 4 * kg + r;
          if (co < a.Cout && xo < a.W) {
            const size_t o = (size_t)co * DHW + (size_t)z * HW + (size_t)yo * a.W + xo;
            float v = acc[xt][r] + (a.bias ? a.bias[co] : 0.0f);
            if (a.relu) v = __builtin_fmaxf(v, 0.0f);
            if (a.skip) v += a.skip[o];
            a.out[o] = v;
          }
        }
      }
    }
    __syncthreads();                  // slice z+2 is complete and everyone is done reading slice z-1
  }
}

template <int CIN>
constexpr int lds_bytes() {
  return 4 * 2 * kHY * kHX * (CIN == 8 ? 16 : (CIN == 16 ? 48 : 80));
}

template <int CIN>
int launch(const Args& a, hipStream_t s) {
  constexpr int lds = lds_bytes<CIN>();
  static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_mfma_kernel<CIN>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { set_error("svs_conv3d_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  dim3 grid((a.W + kTX - 1) / kTX, (a.H + kTY - 1) / kTY, (a.D + a.z_per_wg - 1) / a.z_per_wg);
  conv3d_mfma_kernel<CIN><<<grid, 256, lds, s>>>(a);
  return check_launch("svs_conv3d_mfma");
}

}  // namespace convmfma
}  // namespace svs

using namespace svs;

extern "C" {

// bytes of the packed A fragments of a layer: [ceil(27 Cin / 32)][2][64][16 B]
size_t svs_conv3d_mfma_wfrag_bytes(int Cin) { return (size_t)((27 * Cin + 31) / 32) * 2 * 64 * 16; }

// out (Cout,D,H,W) = [relu](conv3d(in (Cin,D,H,W), 3x3x3, stride 1, padding 1) + bias) [+ skip]; Cin in {8,16,32},
// Cout <= 16.  wfrag: the folded weights as fp16 hi / mid A fragments of v_mfma_f32_16x16x32_f16: fragment
// [k-step s][piece][lane] holds, for output channel lane & 15 and j = 0..7, the weight of k = 32 s + 8 (lane >> 4) + j
// = tap * Cin + cin (tap = (kd*3+kh)*3+kw; zero for tap > 26 and for channels >= Cout).
int svs_conv3d_mfma(const float* in, const void* wfrag, const float* bias, const float* skip, float* out, int Cin,
                    int Cout, int D, int H, int W, int relu, void* hip_stream) {
  if (!in || !wfrag || !out || Cout < 1 || Cout > 16 || D < 1 || H < 1 || W < 1 || (Cin != 8 && Cin != 16 && Cin != 32)) {
    set_error("svs_conv3d_mfma: bad argument (Cin in {8,16,32}, Cout <= 16)"); return SVS_EINVAL;
  }
  convmfma::Args a{in, reinterpret_cast<const uint4*>(wfrag), bias, skip, out, Cout, D, H, W, relu, 0};
  // Split z so that the launch is a whole number of rounds on the 256 CUs (one workgroup per CU at a time) with at
  // least ~16 z steps per workgroup (each pays a three-slice prologue)
  const int xy = ((W + convmfma::kTX - 1) / convmfma::kTX) * ((H + convmfma::kTY - 1) / convmfma::kTY);
  int best = 1;
  double best_cost = 1e30;
  for (int zs = 1; zs <= D && zs <= 64; ++zs) {
    const int zp = (D + zs - 1) / zs;
    const long long wgs = (long long)xy * ((D + zp - 1) / zp);
    const double rounds = (double)((wgs + 255) / 256);
    const double cost = rounds * (zp + 3);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = zs; }
  }
  a.z_per_wg = (D + best - 1) / best;
  hipStream_t s = (hipStream_t)hip_stream;
  if (Cin == 8) return convmfma::launch<8>(a, s);
  if (Cin == 16) return convmfma::launch<16>(a, s);
  return convmfma::launch<32>(a, s);
}

}  // extern "C"
