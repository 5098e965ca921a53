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
#include <cstdlib>

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


// --------------------------------------------------------------------------------------------------------------
// fp16x2 variant: the same contraction on v_mfma_f32_32x32x16_f16 with two-piece fp16 operands (svs_mlp_h2_dev.h).
//
// The contraction index is the POINT, which lives on the lanes of the activation blocks, so both operands are
// transposed through LDS: a staging thread converts its float4 (4 consecutive features of one point) into hi / mid
// fp16 pieces and stores each as ONE ds_write_b64 into a [point][feature] image; the MFMA fragments (8 consecutive
// points of one feature per lane) come back through ds_read_b64_tr_b16, the hardware transpose read.  Images are
// [32 points][128 features] sub-tiles with 256-byte rows and the chunk swizzle of cdna_hip_programming.md T10 (b):
// both the 8-byte writes and the transposed reads are conflict-free.
//
// Gradient-like operands (A of pair 0, B of pair 1) span many orders of magnitude and can be far below fp16's range:
// they are multiplied by a power of two s chosen from their maximum magnitude (absmax, produced on the device by the
// kernels that write them) so that the largest element is ~2^10, and the accumulators are multiplied by 1/s before
// the flush.  Elements below 2^-13 of the maximum keep an absolute precision of 2^-35 of it.
// --------------------------------------------------------------------------------------------------------------
namespace h2 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
#define SVS_LDS(T, ptr) ((__attribute__((address_space(3))) T*)(ptr))

constexpr int kSub = 32 * 256;         // bytes of one [32 points][128 features] fp16 sub-image
constexpr int kPiece = 2 * kSub;       // 256 features
constexpr int kOperand = 2 * kPiece;   // hi piece, mid piece
constexpr int kExtraPiece = 32 * 64;   // 9th B tile: [32 points][32 features], 64-byte rows
__host__ __device__ constexpr int buf_bytes(int ntb) { return 2 * kOperand + (ntb == 9 ? 2 * kExtraPiece : 0); }

__device__ __forceinline__ f16x8 tr_frag(const unsigned char* lds, int a0, int a1) {
  struct Pair { s16x4 lo, hi; } v;
  v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SVS_LDS(s16x4, lds + a0));
  v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SVS_LDS(s16x4, lds + a1));
  return __builtin_bit_cast(f16x8, v);
}

// power-of-two scale from the maximum magnitude: s * absmax in [2^10, 2^11)
__device__ __forceinline__ void scale_from_absmax(const float* absmax, float& s, float& inv_s) {
  s = 1.0f; inv_s = 1.0f;
  if (!absmax) return;
  int e = (int)((__float_as_uint(*absmax) >> 23) & 0xff);
  e = e < 12 ? 12 : (e > 250 ? 250 : e);
  s = __uint_as_float((unsigned)(264 - e) << 23);
  inv_s = __uint_as_float((unsigned)(e - 10) << 23);
}

template <int NTB, int dbg = 0>
__global__ __launch_bounds__(256, 1) void wgrad_h2_kernel(Args a, const float* __restrict__ absmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
  constexpr int BUF = buf_bytes(NTB);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float s_grad, inv_s;
  scale_from_absmax(absmax, s_grad, inv_s);

  f32x16 acc[2][NTB];
  f32x16 accb[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    accb[o] = (f32x16)(0.0f);
#pragma unroll
    for (int i = 0; i < NTB; ++i) acc[o][i] = (f32x16)(0.0f);
  }

  // ---- reader addresses (bytes inside an operand piece), see the layout notes above
  const int ri = lane & 15, rg = (lane >> 4) & 1, rh = lane >> 5, rq = ri >> 2, rp = ri & 3;
  int rbase[2], rt[4], rx[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    rbase[e] = 256 * (8 * rh + 4 * e + rq) + 16 * ((2 * rg + (rp >> 1)) ^ ((2 * rh + e) & 3)) + 8 * (rp & 1);
    rx[e] = 64 * (8 * rh + 4 * e + rq) + 32 * rg + 8 * rp;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) rt[j] = 64 * (j ^ rq);
  int aaddr[2][2];   // A operand: output tiles 2*wave, 2*wave+1
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int ft = 2 * wave + o;
#pragma unroll
    for (int e = 0; e < 2; ++e) aaddr[o][e] = (ft >> 2) * kSub + rbase[e] + 64 * ((ft & 3) ^ rq);
  }
  // ---- writer addresses
  const int wp = lane & 31, whf = lane >> 5;
  const int wbase = 256 * wp + 16 * (wave ^ ((wp >> 2) & 3)) + 8 * whf;
  const int wxoff = 64 * wp + 16 * wave + 8 * whf;

  const int my_tiles = a.n_tiles > (int)blockIdx.x ? (a.n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int n_items = my_tiles * a.n_pairs;
  Staging st;
  int st_live = 32, st_pair = 0;

  auto issue = [&](int item) {
    const int t = blockIdx.x + (item / a.n_pairs) * gridDim.x, pi = item % a.n_pairs;
    const Pair& p = a.p[pi];
    const f32x4* ga = reinterpret_cast<const f32x4*>(p.a + (size_t)t * p.stride_a);
    const f32x4* gb = reinterpret_cast<const f32x4*>(p.b + (size_t)t * p.stride_b);
#pragma unroll
    for (int k = 0; k < 8; ++k) { st.a[k] = ga[k * 256 + tid]; st.b[k] = gb[k * 256 + tid]; }
    st.has_h = p.a_h != nullptr;
    if (st.has_h) {
      const f32x4* gh = reinterpret_cast<const f32x4*>(p.a_h + (size_t)t * p.stride_h);
#pragma unroll
      for (int k = 0; k < 8; ++k) st.h[k] = gh[k * 256 + tid];
    }
    if (NTB == 9) {
      st.x = (a.b_extra && pi == 0) ? reinterpret_cast<const f32x4*>(a.b_extra + (size_t)t * a.stride_extra)[tid]
                                    : (f32x4)(0.0f);
    }
    st_live = a.n_valid_points - t * 32;
    st_pair = pi;
  };
  auto put = [&](unsigned char* piece_hi, unsigned char* piece_mid, int off, const f32x4& v) {
    f16x4 hi, mid;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const _Float16 h = (_Float16)v[j];
      hi[j] = h;
      mid[j] = (_Float16)(v[j] - (float)h);
    }
    *SVS_LDS(f16x4, piece_hi + off) = hi;
    *SVS_LDS(f16x4, piece_mid + off) = mid;
  };
  auto commit = [&](int buf) {
    unsigned char* la = smem_h2 + buf * BUF;
    unsigned char* lb = la + kOperand;
    const float sa = st_pair == 0 ? s_grad : 1.0f, sb = st_pair == 0 ? 1.0f : s_grad;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f32x4 va = st.a[k];
      if (st.has_h) {
        va[0] *= dsoftplus_from_h(st.h[k][0]); va[1] *= dsoftplus_from_h(st.h[k][1]);
        va[2] *= dsoftplus_from_h(st.h[k][2]); va[3] *= dsoftplus_from_h(st.h[k][3]);
      }
      if (wp >= st_live) va = (f32x4)(0.0f);     // ragged last tile: points beyond the batch contribute nothing
      const int off = (k >> 2) * kSub + wbase + 64 * ((k & 3) ^ (wp & 3));
      put(la, la + kPiece, off, va * sa);
      put(lb, lb + kPiece, off, st.b[k] * sb);
    }
    if (NTB == 9) {
      unsigned char* lx = la + 2 * kOperand;
      put(lx, lx + kExtraPiece, wxoff, st.x * sb);
    }
  };

  f16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;

  if (n_items > 0) { issue(0); commit(0); }
  __syncthreads();
  for (int item = 0; item < n_items; ++item) {
    const int buf = item & 1;
    const bool more = item + 1 < n_items;
    if (more && dbg != 3) issue(item + 1);
    const unsigned char* la = smem_h2 + buf * BUF;
    const unsigned char* lb = la + kOperand;
    const unsigned char* lx = la + 2 * kOperand;
    const bool want_bias = a.db && (item % a.n_pairs) == 0;
    if (dbg != 1)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 ah[2], am[2];
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        ah[o] = tr_frag(la + ks * 4096, aaddr[o][0], aaddr[o][1]);
        am[o] = tr_frag(la + kPiece + ks * 4096, aaddr[o][0], aaddr[o][1]);
      }
      auto bfrag = [&](int i, f16x8& bh, f16x8& bm) {
        if (i < 8) {
          const unsigned char* base = lb + (i >> 2) * kSub + ks * 4096;
          bh = tr_frag(base, rbase[0] + rt[i & 3], rbase[1] + rt[i & 3]);
          bm = tr_frag(base + kPiece, rbase[0] + rt[i & 3], rbase[1] + rt[i & 3]);
        } else {
          bh = tr_frag(lx + ks * 1024, rx[0], rx[1]);
          bm = tr_frag(lx + kExtraPiece + ks * 1024, rx[0], rx[1]);
        }
      };
      f16x8 bh, bm;
      bfrag(0, bh, bm);
      if (want_bias) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          accb[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[o], ones, accb[o], 0, 0, 0);
          accb[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[o], ones, accb[o], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < NTB; ++i) {
        f16x8 nh, nm;
        __builtin_amdgcn_sched_barrier(0);
        acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[0], bh, acc[0][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < NTB) bfrag(i + 1, nh, nm);       // next B tile: behind one MFMA, ahead of five
        __builtin_amdgcn_sched_barrier(0);
        acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[1], bh, acc[1][i], 0, 0, 0);
        acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bm, acc[0][i], 0, 0, 0);
        acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bm, acc[1][i], 0, 0, 0);
        acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh, acc[0][i], 0, 0, 0);
        acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh, acc[1][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < NTB) { bh = nh; bm = nm; }
      }
    }
    if (more && dbg != 2) commit(buf ^ 1);
    __syncthreads();
  }
  // flush: C[row = rho(r) + 4*half][col]; two 128-byte row segments per wave-instruction
  const int half = lane >> 5, col = lane & 31;
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int i = 0; i < NTB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (2 * wave + o) + rho(r) + 4 * half;
        const int c = 32 * i + col;
        if (c < a.ldw) atomicAdd(&a.dW[(size_t)row * a.ldw + c], acc[o][i][r] * inv_s);
      }
  if (a.db && col == 0) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(&a.db[32 * (2 * wave + o) + rho(r) + 4 * half], accb[o][r] * inv_s);
  }
}

}  // namespace h2

}  // namespace wgrad
}  // namespace svs

using namespace svs;
using namespace svs::wgrad;
namespace h2 = svs::wgrad::h2;

extern "C" {

// dW[256][ldw] += sum_p A(p) B(p)^T over n_pairs (<= 2) operand pairs; db[256] += row sums of A of pair 0.
// a/b/a_h: wave-tile blocks with the given strides (floats between point tiles); a_h optional (A *= softplus'(h)).
// b_extra: optional 16 extra B rows per tile (columns 256..271 of dW; ldw >= 288).
int svs_wgrad(const float* a0, const float* a0_h, const float* b0, long long sa0, long long sh0, long long sb0,
              const float* a1, const float* a1_h, const float* b1, long long sa1, long long sh1, long long sb1,
              const float* b_extra, long long s_extra, int n_points, int precision, const float* absmax, float* dW,
              int ldw, float* db, void* hip_stream) {
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
  if (precision == mlp::kFmtF16x2) {
    if (b_extra) {
      constexpr int lds = 2 * h2::buf_bytes(9);
      static hipError_t e9 = hipFuncSetAttribute(reinterpret_cast<const void*>(h2::wgrad_h2_kernel<9>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e9 != hipSuccess) { set_error("svs_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e9)); return (int)e9; }
      h2::wgrad_h2_kernel<9><<<grid, 256, lds, s>>>(a, absmax);
    } else {
      constexpr int lds = 2 * h2::buf_bytes(8);
      static hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(h2::wgrad_h2_kernel<8>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e8 != hipSuccess) { set_error("svs_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e8)); return (int)e8; }
      {
        const char* m = getenv("SVS_WG_MODE"); const int mode = m ? atoi(m) : 0;
        static hipError_t ee = (hipFuncSetAttribute(reinterpret_cast<const void*>(h2::wgrad_h2_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
                                hipFuncSetAttribute(reinterpret_cast<const void*>(h2::wgrad_h2_kernel<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds),
                                hipFuncSetAttribute(reinterpret_cast<const void*>(h2::wgrad_h2_kernel<8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        (void)ee;
        if (mode == 1) h2::wgrad_h2_kernel<8, 1><<<grid, 256, lds, s>>>(a, absmax);
        else if (mode == 2) h2::wgrad_h2_kernel<8, 2><<<grid, 256, lds, s>>>(a, absmax);
        else if (mode == 3) h2::wgrad_h2_kernel<8, 3><<<grid, 256, lds, s>>>(a, absmax);
        else h2::wgrad_h2_kernel<8><<<grid, 256, lds, s>>>(a, absmax);
      }
    }
    return check_launch("svs_wgrad");
  }
  if (precision != mlp::kFmtF32) { set_error("svs_wgrad: unknown precision %d", precision); return SVS_EINVAL; }
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
