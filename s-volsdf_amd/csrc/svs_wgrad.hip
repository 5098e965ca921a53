// Weight-gradient GEMM for the fused MLPs: dW[o][i] (+)= sum over points p of A[o,p] * B[i,p], with A and B
// stored as wave-tile activation blocks (svs_mlp.hip: 256 feature rows x 32 points per block, float4 index
// (i/4)*64+lane).  This is the "K = number of points" contraction of the training backward
// (d loss / d W_l = a_bar_l h_l^T + g_hat_l u_l^T).
//
// One workgroup (4 waves) owns a full 256 x (32*NTB) accumulator (wave w: output tiles 2w, 2w+1 x all input
// tiles = 32*NTB/... accumulator registers) and walks over its share of the point tiles (split-K over
// workgroups); a tile's A and B blocks are transposed through LDS ([row][33] pitch: conflict-free column
// reads) into the MFMA operand layout; partial sums are flushed with float atomics, two 128-byte row segments
// per wave-instruction (the full-rate shape, MI355X_MICROARCH.md "Global float atomics").
#include "svs_common.h"
#include "svs_mlp_layout.h"

namespace svs {
namespace wgrad {

constexpr int kPitch = 33;

struct Pair {
  const float* a;        // blocks [n_tiles][block_stride_a]: A rows (output features / gradients)
  const float* a_h;      // optional: A is multiplied by softplus'(.) = 1 - exp(-100 h) of this block (same layout)
  const float* b;        // blocks: B rows (input features)
  size_t stride_a, stride_h, stride_b;   // floats between consecutive point tiles
  int relu_mask_b;       // unused (reserved)
};

struct Args {
  Pair p[2];
  int n_pairs;
  int n_tiles;           // point tiles (32 points each)
  int n_valid_points;    // points beyond this index contribute nothing (ragged last tile)
  const float* b_extra;  // optional 9th B tile: [n_tiles][16*64] (16 rows x 32 points, both halves) or nullptr
  size_t stride_extra;
  float* dW;             // [256][ldw] accumulated with atomics (caller zeroes)
  int ldw;
  float* db;             // [256] row sums of A of pair 0 (bias gradient) or nullptr
};

__device__ __forceinline__ float dsoftplus_from_h(float h) {
  return 1.0f - __builtin_amdgcn_exp2f(h * (-100.0f * 1.44269504088896341f));
}

// global wave-tile block -> LDS [feature][kPitch] (feature = 32*tile + 8*(i4%4) + j + 4*half, column = point)
__device__ __forceinline__ void stage_block(const float* __restrict__ g, const float* __restrict__ gh, float* lds,
                                            int n_f4_rows /* 32 for 256 rows */, bool zero_tail, int n_live_pts) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, pt = lane & 31;
  for (int i4 = wave; i4 < n_f4_rows; i4 += 4) {
    f32x4 v = reinterpret_cast<const f32x4*>(g)[i4 * 64 + lane];
    if (gh) {
      const f32x4 h = reinterpret_cast<const f32x4*>(gh)[i4 * 64 + lane];
      v[0] *= dsoftplus_from_h(h[0]); v[1] *= dsoftplus_from_h(h[1]);
      v[2] *= dsoftplus_from_h(h[2]); v[3] *= dsoftplus_from_h(h[3]);
    }
    if (zero_tail && pt >= n_live_pts) v = (f32x4)(0.0f);
    const int feat = 32 * (i4 >> 2) + 8 * (i4 & 3) + 4 * half;
#pragma unroll
    for (int j = 0; j < 4; ++j) lds[(feat + j) * kPitch + pt] = v[j];
  }
}

template <int NTB>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(Args a) {
  extern __shared__ float smem[];
  float* ldsA = smem;                          // 256 x 33
  float* ldsB = smem + 256 * kPitch;           // (32*NTB) x 33
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, col = lane & 31;
  f32x16 acc[2][NTB];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int i = 0; i < NTB; ++i) acc[o][i] = (f32x16)(0.0f);
  float bias_acc = 0.0f;

  for (int t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    const int live = a.n_valid_points - t * 32;
    const bool ragged = live < 32;
    for (int pi = 0; pi < a.n_pairs; ++pi) {
      const Pair& p = a.p[pi];
      __syncthreads();
      stage_block(p.a + (size_t)t * p.stride_a, p.a_h ? p.a_h + (size_t)t * p.stride_h : nullptr, ldsA, 32, ragged, live);
      stage_block(p.b + (size_t)t * p.stride_b, nullptr, ldsB, 32, false, 32);
      if (NTB == 9) {
        // extra tile: 16 rows stored as 4 float4 rows per lane-half layout (registers 0..15 of one tile)
        if (a.b_extra && pi == 0) stage_block(a.b_extra + (size_t)t * a.stride_extra, nullptr, ldsB + 256 * kPitch, 4, false, 32);
        else for (int i = threadIdx.x; i < 32 * kPitch; i += 256) ldsB[256 * kPitch + i] = 0.0f;
      }
      __syncthreads();
      if (a.db && pi == 0) {
        float s = 0.0f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) s += ldsA[threadIdx.x * kPitch + q];
        bias_acc += s;
      }
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const int pt = 2 * s + half;
        const float a0 = ldsA[(32 * (2 * wave) + col) * kPitch + pt];
        const float a1 = ldsA[(32 * (2 * wave + 1) + col) * kPitch + pt];
#pragma unroll
        for (int i = 0; i < NTB; ++i) {
          const float b = ldsB[(32 * i + col) * kPitch + pt];
          acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][i], 0, 0, 0);
          acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][i], 0, 0, 0);
        }
      }
    }
  }
  // flush: C[row = rho(r) + 4*half][col]; two 128-byte row segments per wave-instruction
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int i = 0; i < NTB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (2 * wave + o) + rho(r) + 4 * half;
        const int c = 32 * i + col;
        if (c < a.ldw) atomicAdd(&a.dW[(size_t)row * a.ldw + c], acc[o][i][r]);
      }
  if (a.db) atomicAdd(&a.db[threadIdx.x], bias_acc);
}

}  // namespace wgrad
}  // namespace svs

using namespace svs;
using namespace svs::wgrad;

extern "C" {

// dW[256][ldw] += sum_p A(p) B(p)^T over n_pairs (<= 2) operand pairs; db[256] += row sums of A of pair 0.
// a/b/a_h: wave-tile blocks with the given strides (floats between point tiles); a_h optional (A *= softplus'(h)).
// b_extra: optional 16 extra B rows per tile (columns 256..271 of dW; ldw >= 272).
int svs_wgrad(const float* a0, const float* a0_h, const float* b0, long long sa0, long long sh0, long long sb0,
              const float* a1, const float* a1_h, const float* b1, long long sa1, long long sh1, long long sb1,
              const float* b_extra, long long s_extra, int n_points, float* dW, int ldw, float* db, void* hip_stream) {
  if (!a0 || !b0 || !dW || n_points <= 0 || ldw < 256 || ldw > 288 || (b_extra && ldw < 288)) { set_error("svs_wgrad: bad argument"); return SVS_EINVAL; }
  Args a;
  a.p[0] = Pair{a0, a0_h, b0, (size_t)sa0, (size_t)sh0, (size_t)sb0, 0};
  a.n_pairs = 1;
  if (a1) {
    if (!b1) { set_error("svs_wgrad: second pair needs b1"); return SVS_EINVAL; }
    a.p[1] = Pair{a1, a1_h, b1, (size_t)sa1, (size_t)sh1, (size_t)sb1, 0};
    a.n_pairs = 2;
  }
  a.n_tiles = (n_points + 31) / 32;
  a.n_valid_points = n_points;
  a.b_extra = b_extra; a.stride_extra = (size_t)s_extra;
  a.dW = dW; a.ldw = ldw; a.db = db;
  hipStream_t s = (hipStream_t)hip_stream;
  const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  if (b_extra) {
    constexpr int lds = (256 + 288) * kPitch * 4;
    static hipError_t e9 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<9>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e9 != hipSuccess) { set_error("svs_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e9)); return (int)e9; }
    wgrad_kernel<9><<<grid, 256, lds, s>>>(a);
  } else {
    constexpr int lds = (256 + 256) * kPitch * 4;
    static hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<8>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e8 != hipSuccess) { set_error("svs_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e8)); return (int)e8; }
    wgrad_kernel<8><<<grid, 256, lds, s>>>(a);
  }
  return check_launch("svs_wgrad");
}

}  // extern "C"
