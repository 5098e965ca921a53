// Weight-gradient GEMM for the fused MLPs: dW[o][i] (+)= sum over points p of A[o,p] * B[i,p], with A and B
// stored as wave-tile activation blocks (svs_mlp.hip: 256 feature rows x 32 points per block, float4 index
// (i/4)*64+lane).  This is the "K = number of points" contraction of the training backward
// (d loss / d W_l = a_bar_l h_l^T + g_hat_l u_l^T).
//
// One workgroup (4 waves, one per SIMD) owns a full 256 x (32*NTB) accumulator (wave w: output tiles 2w, 2w+1 x
// all input tiles) and walks over its share of the (point tile, operand pair) items -- split-K over workgroups.
// An item's A and B blocks are staged global -> registers -> LDS as [point][feature] rows of pitch 260/292 floats:
// a lane's float4 (4 consecutive features of one point) is ONE conflict-free ds_write_b128, and the MFMA operand
// reads (32 consecutive features of one point per half-wave) are conflict-free ds_read_b32.  The loads of item i+1
// are in flight while item i's MFMAs run (two LDS buffers, one barrier per item).  Partial sums are flushed with
// float atomics, two 128-byte row segments per wave-instruction (the full-rate shape, MI355X_MICROARCH.md
// "Global float atomics").
#include "svs_common.h"
#include "svs_mlp_layout.h"

namespace svs {
namespace wgrad {

struct Pair {
  const float* a;        // blocks [n_tiles][stride_a]: A rows (gradients w.r.t. layer outputs)
  const float* a_h;      // optional: A is multiplied by softplus'(.) = 1 - exp(-100 h) of this block (same layout)
  const float* b;        // blocks: B rows (layer inputs)
  size_t stride_a, stride_h, stride_b;   // floats between consecutive point tiles
};

struct Args {
  Pair p[2];
  int n_pairs;
  int n_tiles;           // point tiles (32 points each)
  int n_valid_points;    // points beyond this index contribute nothing (ragged last tile)
  const float* b_extra;  // optional 9th B tile: [n_tiles][stride_extra], 16 registers x 64 lanes (pair 0 only)
  size_t stride_extra;
  float* dW;             // [256][ldw] accumulated with atomics (caller zeroes)
  int ldw;
  float* db;             // [256] row sums of A of pair 0 (bias gradient) or nullptr
};

__device__ __forceinline__ float dsoftplus_from_h(float h) {
  return 1.0f - __builtin_amdgcn_exp2f(h * (-100.0f * 1.44269504088896341f));
}

struct Staging {
  f32x4 a[8], b[8], h[8], x;
  bool has_h;
};

template <int NTB>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(Args a) {
  constexpr int PA = 260, PB = NTB == 9 ? 292 : 260;       // row pitches (floats), both = 4 mod 32
  constexpr int BUF = 32 * (PA + PB);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, col = lane & 31;
  f32x16 acc[2][NTB];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int i = 0; i < NTB; ++i) acc[o][i] = (f32x16)(0.0f);
  float bias_acc = 0.0f;

  const int my_tiles = a.n_tiles > (int)blockIdx.x ? (a.n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int n_items = my_tiles * a.n_pairs;
  Staging st;
  int st_live = 32;

  auto issue = [&](int item) {
    const int t = blockIdx.x + (item / a.n_pairs) * gridDim.x, pi = item % a.n_pairs;
    const Pair& p = a.p[pi];
    const f32x4* ga = reinterpret_cast<const f32x4*>(p.a + (size_t)t * p.stride_a);
    const f32x4* gb = reinterpret_cast<const f32x4*>(p.b + (size_t)t * p.stride_b);
#pragma unroll
    for (int k = 0; k < 8; ++k) { st.a[k] = ga[k * 256 + tid]; st.b[k] = gb[k * 256 + tid]; }
    st.has_h = p.a_h != nullptr;
    if (st.has_h) {       // the softplus' factor is applied at commit time, so that these loads stay in flight
      const f32x4* gh = reinterpret_cast<const f32x4*>(p.a_h + (size_t)t * p.stride_h);
#pragma unroll
      for (int k = 0; k < 8; ++k) st.h[k] = gh[k * 256 + tid];
    }
    if (NTB == 9) {
      st.x = (a.b_extra && pi == 0) ? reinterpret_cast<const f32x4*>(a.b_extra + (size_t)t * a.stride_extra)[tid]
                                    : (f32x4)(0.0f);
    }
    st_live = a.n_valid_points - t * 32;
  };
  auto commit = [&](int buf) {
    float* la = smem + buf * BUF;
    float* lb = la + 32 * PA;
    if (st.has_h) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        st.a[k][0] *= dsoftplus_from_h(st.h[k][0]); st.a[k][1] *= dsoftplus_from_h(st.h[k][1]);
        st.a[k][2] *= dsoftplus_from_h(st.h[k][2]); st.a[k][3] *= dsoftplus_from_h(st.h[k][3]);
      }
    }
    if (st_live < 32) {
      // ragged last tile: points beyond the batch contribute nothing
#pragma unroll
      for (int k = 0; k < 8; ++k) if ((tid & 31) >= st_live) st.a[k] = (f32x4)(0.0f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int q = k * 256 + tid, i4 = q >> 6, ln = q & 63;
      const int feat = 32 * (i4 >> 2) + 8 * (i4 & 3) + 4 * (ln >> 5), pt = ln & 31;
      *reinterpret_cast<f32x4*>(la + pt * PA + feat) = st.a[k];
      *reinterpret_cast<f32x4*>(lb + pt * PB + feat) = st.b[k];
    }
    if (NTB == 9) {
      const int i4 = tid >> 6, ln = tid & 63;
      *reinterpret_cast<f32x4*>(lb + (ln & 31) * PB + 256 + 8 * i4 + 4 * (ln >> 5)) = st.x;
    }
  };

  if (n_items > 0) { issue(0); commit(0); }
  __syncthreads();
  for (int item = 0; item < n_items; ++item) {
    const int buf = item & 1;
    const bool more = item + 1 < n_items;
    if (more) issue(item + 1);
    const float* la = smem + buf * BUF;
    const float* lb = la + 32 * PA;
    if (a.db && (item % a.n_pairs) == 0) {
      float s = 0.0f;
#pragma unroll 8
      for (int q = 0; q < 32; ++q) s += la[q * PA + tid];
      bias_acc += s;
    }
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const int pt = 2 * s + half;
      const float a0 = la[pt * PA + 32 * (2 * wave) + col];
      const float a1 = la[pt * PA + 32 * (2 * wave + 1) + col];
#pragma unroll
      for (int i = 0; i < NTB; ++i) {
        const float b = lb[pt * PB + 32 * i + col];
        acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][i], 0, 0, 0);
        acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][i], 0, 0, 0);
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
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
  if (a.db) atomicAdd(&a.db[tid], bias_acc);
}

}  // namespace wgrad
}  // namespace svs

using namespace svs;
using namespace svs::wgrad;

extern "C" {

// dW[256][ldw] += sum_p A(p) B(p)^T over n_pairs (<= 2) operand pairs; db[256] += row sums of A of pair 0.
// a/b/a_h: wave-tile blocks with the given strides (floats between point tiles); a_h optional (A *= softplus'(h)).
// b_extra: optional 16 extra B rows per tile (columns 256..271 of dW; ldw >= 288).
int svs_wgrad(const float* a0, const float* a0_h, const float* b0, long long sa0, long long sh0, long long sb0,
              const float* a1, const float* a1_h, const float* b1, long long sa1, long long sh1, long long sb1,
              const float* b_extra, long long s_extra, int n_points, float* dW, int ldw, float* db, void* hip_stream) {
  if (!a0 || !b0 || !dW || n_points <= 0 || ldw < 256 || ldw > 288 || (b_extra && ldw < 288)) {
    set_error("svs_wgrad: bad argument"); return SVS_EINVAL;
  }
  Args a;
  a.p[0] = Pair{a0, a0_h, b0, (size_t)sa0, (size_t)sh0, (size_t)sb0};
  a.n_pairs = 1;
  if (a1) {
    if (!b1) { set_error("svs_wgrad: second pair needs b1"); return SVS_EINVAL; }
    a.p[1] = Pair{a1, a1_h, b1, (size_t)sa1, (size_t)sh1, (size_t)sb1};
    a.n_pairs = 2;
  }
  a.n_tiles = (n_points + 31) / 32;
  a.n_valid_points = n_points;
  a.b_extra = b_extra; a.stride_extra = (size_t)s_extra;
  a.dW = dW; a.ldw = ldw; a.db = db;
  hipStream_t s = (hipStream_t)hip_stream;
  const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  if (b_extra) {
    constexpr int lds = 2 * 32 * (260 + 292) * 4;
    static hipError_t e9 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<9>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e9 != hipSuccess) { set_error("svs_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e9)); return (int)e9; }
    wgrad_kernel<9><<<grid, 256, lds, s>>>(a);
  } else {
    constexpr int lds = 2 * 32 * (260 + 260) * 4;
    static hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<8>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e8 != hipSuccess) { set_error("svs_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e8)); return (int)e8; }
    wgrad_kernel<8><<<grid, 256, lds, s>>>(a);
  }
  return check_launch("svs_wgrad");
}

}  // extern "C"
