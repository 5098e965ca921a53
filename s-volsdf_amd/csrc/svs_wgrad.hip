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
#include "svs_ticket.h"
#include <cstdlib>

namespace svs {
namespace wgrad {

struct Pair {
  const float* a;        // blocks [n_tiles][stride_a]: A rows (gradients w.r.t. layer outputs)
  const float* b;        // blocks: B rows (layer inputs)
  size_t stride_a, stride_b;   // floats between consecutive point tiles
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
  det::Ticket ticket;    // deterministic mode: the workgroups flush in block order (svs_ticket.h)
};

struct Staging {
  f32x4 a[8], b[8], x;
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
    if (NTB == 9) {
      st.x = (a.b_extra && pi == 0) ? reinterpret_cast<const f32x4*>(a.b_extra + (size_t)t * a.stride_extra)[tid]
                                    : (f32x4)(0.0f);
    }
    st_live = a.n_valid_points - t * 32;
  };
  auto commit = [&](int buf) {
    float* la = smem + buf * BUF;
    float* lb = la + 32 * PA;
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
  det::wait_turn(a.ticket, blockIdx.x);
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
  det::pass_turn(a.ticket, blockIdx.x);
}


// --------------------------------------------------------------------------------------------------------------
// fp16x2 variant: the same contraction on v_mfma_f32_32x32x16_f16, reading the operand blocks in the forms the sweeps
// store them in (svs_blocks_h2.h): no conversion, no split, no staging registers.
//   pair 0:  A = abar_l / zbar_l / fbar  scaled block, per-point scale
//            B = h_l / r_l / feature     pair block
//   pair 1:  A = ghat_l                  stored unscaled
//            B = u_l                     scaled block, per-point scale
//   narrow:  B = the 16 / 32 extra input rows of a radiance network's first layer, ONE float32 tile (split here)
// GP = true (precision SVS_MMA_F16X2): every operand with both fp16 pieces, three MFMAs per k-step and B tile (hi hi + mid
// hi + hi mid: the float32 accuracy class), 64 KiB per item, a ring of 2 item slots.
// GP = false (SVS_MMA_F16X2_HALF): hi planes only (the scaled blocks hold nothing else), one MFMA, 32 KiB per item, a ring
// of 4 slots with 3 items in flight.
//
// The contraction index is the POINT, which lives on the lanes of the fragments.  A 16-KiB plane is copied into LDS as it
// stands by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a ring of four item slots (A plane, B
// plane), three items ahead of the one being multiplied; the MFMA fragments (8 consecutive points of one feature per
// lane) are read TRANSPOSED out of the plane with ds_read_b64_tr_b16 -- the slot permutation of the planes
// (piece_slot(), svs_blocks_h2.h) is what makes these reads conflict-free: for the 32 lanes of a half
//     byte = 1024 s + 256 (a >> 1) + 128 ((a & 1) ^ (s & 1)) + 64 (p & 1) + 16 q + 8 (p >> 1)
// with s = feature / 16 (two values per half: the two 16-lane groups), a = point / 4, q = point % 4, p = the lane's
// column quad: 32 distinct 8-byte slots of one 256-byte line.
//
// Scaled operands carry value * s_p with one power of two s_p per point (the point's largest element at ~2^4); a
// contraction over points needs ONE scale, so every A fragment is multiplied by the factors s / s_p of its 8 points in
// fp16 (exact: powers of two) -- for pair 1 the factors of B's points are applied to A, which is the same product -- with
// s chosen from the published maximum of the scaled operand (absmax) so that the largest element of the launch is
// ~2^10; the accumulators are multiplied by 1 / s before the flush.  Factors are clamped to 2^15; points beyond the
// batch get factor 0.
// --------------------------------------------------------------------------------------------------------------
namespace h2 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
#define SVS_LDS(T, ptr) ((__attribute__((address_space(3))) T*)(ptr))

constexpr int kPlane = 16384;          // bytes of one fp16 plane of a block (svs_blocks_h2.h)
constexpr int kExtraPiece = 32 * 64;   // narrow B tile: [32 points][32 features] fp16, 64-byte rows (hi, then mid)
constexpr int kNarrowImg = 8192;       // a narrow job has no B plane: its float32 B tile lands (LDS-DMA) at the B plane's
                                       // place, its fp16 image (hi, mid) kNarrowImg bytes further
constexpr int kRecBytes = 8 * 256;     // per ring slot: every wave's copy of the scaled operand's record (64 floats)
constexpr int kFactorBytes = 8 * 64;   // per ring slot: 8 waves x 32 fp16 factors
template <bool GP>
struct Ring {
  static constexpr int kPlanes = GP ? 2 : 1;            // planes per operand
  static constexpr int kSlot = 2 * kPlanes * kPlane;    // ring slot: A plane(s), B plane(s)
  static constexpr int kB = kPlanes * kPlane;           // offset of the B planes inside a slot
  static constexpr int kRing = GP ? 2 : 4;
  static constexpr int kAhead = kRing - 1;              // items in flight beyond the one being multiplied
  static constexpr int kSlotAll = kSlot + kRecBytes + kFactorBytes;
  static constexpr int kLdsBytes = kRing * kSlotAll;
};

// LDS reads of the main loop go through inline asm: hipcc's waitcnt pass makes every LDS access it can see wait for ALL
// LDS-DMA in flight (vmcnt(0): it cannot tell the ring slots apart), which would un-pipeline the ring.  The reads are
// ordered against the copies by the counted vmcnt waits + barriers of the loop; lds_wait() is the lgkmcnt side.
struct Frag { s16x4 lo, hi; };
__device__ __forceinline__ void tr_issue(Frag& v, unsigned a0, unsigned a1) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v.lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v.hi) : "v"(a1));
}
__device__ __forceinline__ f16x8 frag_of(const Frag& v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ void lds_read128(f32x4& v, unsigned a) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); }
__device__ __forceinline__ void lds_read32(float& v, unsigned a) { asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); }
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);      // hipcc hoists register-only MFMAs above an asm wait otherwise (rule 18)
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)SVS_LDS(const void, p);
}

// power-of-two scale from the maximum magnitude: s * absmax in [2^14, 2^15) -- as high as fp16 allows (the re-scaled A
// fragments of points whose gradients are far below the launch's maximum go subnormal in fp16: every bit of range above
// them is a bit of precision for those points)
__device__ __forceinline__ void scale_from_absmax(const float* absmax, float& s, float& inv_s) {
  s = 1.0f; inv_s = 1.0f;
  if (!absmax) return;
  int e = (int)((__float_as_uint(*absmax) >> 23) & 0xff);
  e = e < 16 ? 16 : (e > 250 ? 250 : e);
  s = __uint_as_float((unsigned)(268 - e) << 23);
  inv_s = __uint_as_float((unsigned)(e - 14) << 23);
}

// One launch covers a list of jobs (the weight gradients of several layers): each job gets a contiguous range of
// workgroups, proportional to its work, which split its point tiles among themselves -- about one workgroup per CU
// in total, so the number of partial sums flushed with atomics per layer is ~256 / n_jobs instead of 256.
struct Job {
  Pair p[2];
  const float* rec[2];   // per pair: the records [tile][64] of its scaled operand (pair 0: A, pair 1: B), or nullptr
  int n_pairs;
  int n_tiles;           // point tiles (32 points each)
  int n_valid_points;    // points beyond this index contribute nothing (ragged last tile)
  int b_tiles;           // 8: B blocks are 256-row blocks; 1: B blocks are single 32-row float32 tiles (1024 floats)
  int col0;              // first dW column of B tile 0
  float* dW;             // [256][ldw] accumulated with atomics (caller zeroes)
  int ldw;
  float* db;             // [256] row sums of A of pair 0 (bias gradient) or nullptr
  const float* absmax;   // device float: max magnitude of the scaled operands, or nullptr (scales ignored)
  int wg_begin, wg_count;
  det::Ticket ticket;    // deterministic mode: the workgroups that add into this job's dW columns flush in turn (svs_ticket.h)
};
constexpr int kMaxJobs = 20;
struct MultiArgs { Job job[kMaxJobs]; int n_jobs; int touch; int reverse; };
static_assert(sizeof(MultiArgs) <= 4096, "kernel arguments");

constexpr int kThreadsW = 512;          // 8 waves: wave w owns output tile w (32 rows) x all B tiles

typedef const __attribute__((address_space(1))) void* gvoid;

template <bool GP>
__global__ __launch_bounds__(kThreadsW, 1) void wgrad_h2_multi_kernel(MultiArgs ma) {
  typedef Ring<GP> R;
  constexpr int kSlot = R::kSlot, kSlotAll = R::kSlotAll, kRing = R::kRing, kAhead = R::kAhead, kB = R::kB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
  int jj = 0;
  for (int q = 1; q < ma.n_jobs; ++q) if ((int)blockIdx.x >= ma.job[q].wg_begin) jj = q;
  const Job& a = ma.job[jj];
  const int wg = (int)blockIdx.x - a.wg_begin, nwg = a.wg_count;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool narrow = a.b_tiles == 1;
  float s_grad, inv_s;
  scale_from_absmax(a.absmax, s_grad, inv_s);

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x16)(0.0f);
  float bsum = 0.0f;     // bias gradient: this lane's row (lane & 31 of the wave's output tile) over its half's points

  unsigned char* ring = smem_h2;

  // ---- transposed-read addresses inside a plane (see the header): this lane is row q', column quad p' of its 16-lane
  // group; group g of its half reads k-step (feature block) 2 tile + g; k-step ks of the contraction adds 512 bytes;
  // the second read of a fragment (points + 4) flips bit 7
  const int rq = (lane & 15) >> 2, rp = lane & 3, rg = (lane >> 4) & 1, rh = lane >> 5;
  const int rd0 = 1024 * rg + 256 * rh + 128 * rg + 64 * (rp & 1) + 16 * rq + 8 * (rp >> 1);
  const int a_rd = 2048 * wave + rd0;
  // narrow B image reads ([32 points][32 features], 64-byte rows)
  const int rx0 = 64 * (8 * rh + rq) + 32 * rg + 8 * rp, rx1 = rx0 + 64 * 4;
  // narrow B image writer (threads 0..255): float4 tid of the float32 tile = registers 4 (tid >> 6) .. +3 of lane tid & 63
  const int wp = lane & 31, whf = lane >> 5;
  const int wxoff = 64 * wp + 16 * (wave & 3) + 8 * whf;

  const bool two_pairs = a.n_pairs == 2;               // (1 or 2)
  const int my_tiles = a.n_tiles > wg ? (a.n_tiles - 1 - wg) / nwg + 1 : 0;
  const int n_items = two_pairs ? 2 * my_tiles : my_tiles;
  const int n_valid = a.n_valid_points;
  const bool has_db = a.db != nullptr;
  // the point tile of an item.  ma.reverse (experiment, SVS_WGRAD_REVERSE=1): the tiles in DESCENDING order -- the sweep that ran
  // just before this launch wrote and read the highest tiles last, so the GEMM would start on what may still sit in the 256 MB
  // Infinity Cache
  const bool rev = ma.reverse != 0;
  // (the job's operand pointers and strides in registers: read through `a` -- kernel-argument memory -- inside issue() they were
  // scalar loads per item, each waited for with lgkmcnt(0), in front of the copies)
  const float* const pa0 = a.p[0].a; const float* const pb0 = a.p[0].b;
  const float* const pa1 = a.p[1].a; const float* const pb1 = a.p[1].b;
  const size_t sa0 = a.p[0].stride_a, sb0 = a.p[0].stride_b, sa1 = a.p[1].stride_a, sb1 = a.p[1].stride_b;
  const float* const rec0 = a.rec[0]; const float* const rec1 = a.rec[1];
  const bool has_scale = a.absmax != nullptr;
  auto tile_of = [&](int item) {
    const int k = two_pairs ? item >> 1 : item;
    return rev ? wg + (my_tiles - 1 - k) * nwg : wg + k * nwg;
  };

  // ---- issue the LDS-DMA of one item into its ring slot: wave w copies the 1-KiB fragments w, w + 8 of every plane, its
  // own copy of the scaled operand's record, and (narrow job) quarter w & 3 of the float32 B tile.  Everything goes
  // through LDS-DMA: with no register loads in the loop the only vmcnt waits are the counted ones below.
  auto issue = [&](int item) {
    const int t = tile_of(item), pi = two_pairs ? item & 1 : 0;
    const float* pa = (pi ? pa1 : pa0) + (size_t)t * (pi ? sa1 : sa0);
    const float* pb = (pi ? pb1 : pb0) + (size_t)t * (pi ? sb1 : sb0);
    unsigned char* slot = ring + (item % kRing) * kSlotAll;
    // (the planes of a block are contiguous in memory and in the slot: piece index 16 = the mid plane's first piece)
#pragma unroll
    for (int j = 0; j < 2 * R::kPlanes; ++j) {
      const int piece = wave + 8 * j;
      __builtin_amdgcn_global_load_lds((gvoid)(reinterpret_cast<const f32x4*>(pa) + piece * 64 + lane),
                                       SVS_LDS(void, slot + piece * 1024), 16, 0, 0);
      if (!narrow)
        __builtin_amdgcn_global_load_lds((gvoid)(reinterpret_cast<const f32x4*>(pb) + piece * 64 + lane),
                                         SVS_LDS(void, slot + kB + piece * 1024), 16, 0, 0);
    }
    if (narrow)
      __builtin_amdgcn_global_load_lds((gvoid)(reinterpret_cast<const f32x4*>(pb) + (wave & 3) * 64 + lane),
                                       SVS_LDS(void, slot + kB + (wave & 3) * 1024), 16, 0, 0);
    // scale record of the scaled operand: pair 0 scales A (abar-like), pair 1 B (u); applied to A either way
    if (has_scale)
      __builtin_amdgcn_global_load_lds((gvoid)((pi ? rec1 : rec0) + (size_t)t * 64 + lane),
                                       SVS_LDS(void, slot + kSlot + wave * 256), 4, 0, 0);
  };
  // ---- L2 prefetch of the item AFTER the one being copied (GP ring: 2 slots of 64 KiB fill the LDS, so only ONE item can be
  // in flight into LDS while one is multiplied -- 64 KiB per CU is less than HBM's latency-bandwidth product wants).  Every
  // thread reads 4 bytes of one 128-byte line of the item after next into a register nobody reads: the line comes into
  // the XCD's L2, and the LDS-DMA of that item, issued one iteration later, finds it there.  One extra vector-memory
  // operation per wave and item, issued BEHIND the item's copies (loads return in order: the counted wait below lets it
  // stay in flight).
  unsigned sink = 0;
  auto touch = [&](int item) {
    const int t = tile_of(item), pi = two_pairs ? item & 1 : 0;
    const Pair& p = a.p[pi];
    constexpr int kLines = R::kPlanes * (kPlane / 128);          // 128-byte lines per operand block
    const int nb = narrow ? 32 : kLines;
    const float* q = tid < kLines ? p.a + (size_t)t * p.stride_a + tid * 32
                                  : p.b + (size_t)t * p.stride_b + (tid - kLines) * 32;
    if (tid < kLines + nb) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(q) : "memory");
  };
  // number of vector-memory operations issue() makes per wave and item
  const int ops = (narrow ? 2 * R::kPlanes + 1 : 4 * R::kPlanes) + (has_scale ? 1 : 0);

  // ---- per-item set-up once the wave's own copies have landed: its factor table; (narrow job) the fp16 B image -- whose
  // float32 source quarters were copied by waves 0..3, hence after the barrier
  auto stage_factors = [&](int item) {
    unsigned char* slot = ring + (item % kRing) * kSlotAll;
    const int t = tile_of(item);
    const int live = n_valid - t * 32;
    float f = 1.0f;
    if (has_scale) {
      // s / s_p: the record is a power of two in [2^-107, 2^107] (PointScale::pow2_for); anything else (a block that
      // was never written) leaves the factor at 1
      float rec;
      lds_read32(rec, lds_addr(slot + kSlot + wave * 256 + 4 * wp));
      lds_wait();
      const unsigned eb = (__float_as_uint(rec) >> 23) & 0xff;
      if (eb >= 20 && eb <= 240) f = __builtin_fminf(s_grad * __uint_as_float((254u << 23) - __float_as_uint(rec)), 32768.0f);
    }
    if (wp >= live) f = 0.0f;
    // (inline assembly: a store hipcc can see is made to wait for every LDS-DMA in flight)
    if (lane < 32) asm volatile("ds_write_b16 %0, %1" :: "v"(lds_addr(slot + kSlot + kRecBytes + wave * 64 + 2 * lane)), "v"((_Float16)f) : "memory");
  };
  auto stage_narrow = [&](int item) {
    unsigned char* slot = ring + (item % kRing) * kSlotAll;
    if (tid < 256) {
      f32x4 v;
      lds_read128(v, lds_addr(slot + kB + tid * 16));
      lds_wait();
      f16x4 hi, mid;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const _Float16 h = (_Float16)v[j];
        hi[j] = h;
        mid[j] = (_Float16)(v[j] - (float)h);
      }
      *SVS_LDS(f16x4, slot + kB + kNarrowImg + wxoff) = hi;
      *SVS_LDS(f16x4, slot + kB + kNarrowImg + kExtraPiece + wxoff) = mid;
    }
  };

  const f16x2 one2 = {(_Float16)1.0f, (_Float16)1.0f};

  // ---- prologue: kAhead items in flight
  for (int i0 = 0; i0 < kAhead && i0 < n_items; ++i0) issue(i0);
  const bool touching = GP && ma.touch;
  bool touched = false;                  // a touch is the youngest outstanding operation of this wave
  if (touching && kAhead < n_items) { touch(kAhead); touched = true; }
  // ---- what a wave does for item `it` once its own copies of it have landed: its factor table.  The wait names how many
  // operations of later items (up to kAhead - 1 of them) may stay outstanding.  (Doing this for item + 1 in front of the last two
  // B tiles of item, so that the table's LDS round trip overlaps the other wave's MFMAs: measured equal or slower, 787-791
  // against 777-786 us alone -- the copy is barely ahead of the multiply phase and the early wait sometimes stalls.)
  auto pre_stage = [&](int it) {
    const int left = n_items - 1 - it;
    const int later = (left < kAhead - 1 ? left : kAhead - 1) * ops;        // 0, ops or 2 ops: 0, 3, 4, 5, 6, 8, 10 (GP: 0)
    if (later == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (later == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (later == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (later == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (later == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (touched) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");       // (GP: later == 0; the touch stays in flight)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage_factors(it);
  };
  for (int item = 0; item < n_items; ++item) {
    const int pi = two_pairs ? item & 1 : 0;
    const bool p0 = pi == 0;
    // the item's own operations are the oldest outstanding ones of this wave; once every wave has waited for its own, everybody's
    // have landed behind the barrier -- which also says that every wave is done with the slot of item - 1, the one
    // item + kAhead is about to be copied into
    pre_stage(item);
    // (raw s_barrier: __syncthreads() carries a fence that hipcc lowers to vmcnt(0), which would also wait for the
    // next item's copies)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (narrow) {
      stage_narrow(item);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    auto issue_next = [&]() {
      if (item + kAhead < n_items) issue(item + kAhead);
      touched = false;
      if (touching && item + kAhead + 1 < n_items) { touch(item + kAhead + 1); touched = true; }
    };
#ifdef SVS_WGRAD_A_SERIAL
    issue_next();
#endif
    const unsigned la = lds_addr(ring + (item % kRing) * kSlotAll);
    const unsigned lb = la + kB;
    const unsigned ftab = la + kSlot + kRecBytes + wave * 64;
    const unsigned nimg = lb + kNarrowImg;
    const bool want_bias = has_db && p0;
    const bool two = narrow;             // only the narrow float32 tile is split into two pieces
#ifndef SVS_WGRAD_A_SERIAL
    // The A fragments of BOTH contraction k-steps (hi and mid pieces, their factors) are requested together, the next item's
    // copies are issued behind the requests, and the fragments are waited for once: the item period is the multiply phase plus
    // what precedes it serially in every wave (section 3f of NOTES/r06.md), and the form before (-DSVS_WGRAD_A_SERIAL: copies
    // first, then per k-step hi then mid, a full LDS round trip each) put four round trips there.
    Frag fa2[2], fam2[2];
    f32x4 fraw2[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      tr_issue(fa2[ks], la + 512 * ks + a_rd, la + 512 * ks + (a_rd ^ 128));
      if (GP) tr_issue(fam2[ks], la + kPlane + 512 * ks + a_rd, la + kPlane + 512 * ks + (a_rd ^ 128));
      lds_read128(fraw2[ks], ftab + 32 * ks + 16 * rh);   // the factors of the fragment's 8 points (16 ks + 8 half + 0..7)
    }
    issue_next();        // (the next item's copies are issued while these reads are on their way)
    lds_wait();
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#ifndef SVS_WGRAD_A_SERIAL
      const f16x8 ah = frag_of(fa2[ks]) * __builtin_bit_cast(f16x8, fraw2[ks]);
      const f16x8 am = GP ? frag_of(fam2[ks]) * __builtin_bit_cast(f16x8, fraw2[ks]) : ah;
#else
      Frag fa;
      f32x4 fraw;
      tr_issue(fa, la + 512 * ks + a_rd, la + 512 * ks + (a_rd ^ 128));
      lds_read128(fraw, ftab + 32 * ks + 16 * rh);
      lds_wait();
      const f16x8 ah = frag_of(fa) * __builtin_bit_cast(f16x8, fraw);
      f16x8 am = ah;
      if (GP) {
        Frag fam;
        tr_issue(fam, la + kPlane + 512 * ks + a_rd, la + kPlane + 512 * ks + (a_rd ^ 128));
        lds_wait();
        am = frag_of(fam) * __builtin_bit_cast(f16x8, fraw);
      }
#endif
      if (want_bias) {
        // row sums of A: the fragment holds 8 points of row lane & 31 (the other lane half holds the other 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f16x2 h2v = {ah[2 * j], ah[2 * j + 1]};
          bsum = __builtin_amdgcn_fdot2(h2v, one2, bsum, false);
          if (GP) {
            const f16x2 m2v = {am[2 * j], am[2 * j + 1]};
            bsum = __builtin_amdgcn_fdot2(m2v, one2, bsum, false);
          }
        }
      }
      // the two waves of a SIMD cover each other's LDS latency: no software pipelining of the B fragments (requesting tile
      // i + 1's fragments before tile i's MFMAs, two register sets and a counted lgkmcnt: bit-identical, <= 1 % -- NOTES/r06.md 3f)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i > 0 && narrow) break;
#if defined(SVS_WGRAD_DIAG) && (SVS_WGRAD_DIAG & 8)   // diagnostic build: the copies and the barriers only (no B fragment reads, no MFMAs)
        asm volatile("" :: "v"(ah), "v"(am));
        continue;
#endif
        Frag fh, fm;
        if (!narrow) {
          const unsigned rd = lb + 2048 * i + 512 * ks + rd0;
          tr_issue(fh, rd, rd ^ 128);
          if (GP) tr_issue(fm, rd + kPlane, (rd ^ 128) + kPlane);
        } else {
          tr_issue(fh, nimg + ks * 1024 + rx0, nimg + ks * 1024 + rx1);
          tr_issue(fm, nimg + kExtraPiece + ks * 1024 + rx0, nimg + kExtraPiece + ks * 1024 + rx1);
        }
        lds_wait();
#if defined(SVS_WGRAD_DIAG) && (SVS_WGRAD_DIAG & 4)   // diagnostic build: the fragments are read, nothing is multiplied
        asm volatile("" :: "v"(fh.lo), "v"(fh.hi), "v"(fm.lo), "v"(fm.hi));
#else
        if (two || GP) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, frag_of(fm), acc[i], 0, 0, 0);
        if (GP) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, frag_of(fh), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, frag_of(fh), acc[i], 0, 0, 0);
#endif
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(sink) : "memory");          // the last touch has returned; `sink` lived until here
  // flush: C[row = rho(r) + 4*half][col]; two 128-byte row segments per wave-instruction
  const int half = lane >> 5, col = lane & 31;
  det::wait_turn(a.ticket, wg);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i > 0 && narrow) break;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * wave + rho(r) + 4 * half;
      const int c = a.col0 + 32 * i + col;
      // (-DSVS_WGRAD_DIAG=<mask>, tools/dev/ab_defs.sh + tools/dev/time_wgrad.py: 1 no flush, 2 plain stores, 4 fragments read but
      // nothing multiplied, 8 copies and barriers only; never defined in the product build)
#if defined(SVS_WGRAD_DIAG) && (SVS_WGRAD_DIAG & 1)   // diagnostic build: no flush (the guard keeps the accumulators alive)
      if (c < a.ldw && acc[i][r] == 1.2345e-31f) a.dW[(size_t)row * a.ldw + c] = acc[i][r];
#elif defined(SVS_WGRAD_DIAG) && (SVS_WGRAD_DIAG & 2) // diagnostic build: plain stores instead of atomics (wrong sums)
      if (c < a.ldw) a.dW[(size_t)row * a.ldw + c] = acc[i][r] * inv_s;
#else
      if (c < a.ldw) atomicAdd(&a.dW[(size_t)row * a.ldw + c], acc[i][r] * inv_s);
#endif
    }
  }
  if (a.db) {
    bsum += __shfl_xor(bsum, 32);
    if (half == 0) atomicAdd(&a.db[32 * wave + col], bsum * inv_s);
  }
  det::pass_turn(a.ticket, wg);
}

}  // namespace h2

}  // namespace wgrad
}  // namespace svs

using namespace svs;
using namespace svs::wgrad;
namespace h2 = svs::wgrad::h2;

extern "C" {

// the C-ABI job record (include/svolsdf_hip.h)
struct svs_wgrad_job {
  const float *a0, *b0; long long sa0, sb0;
  const float *a1, *b1; long long sa1, sb1;
  const float* b_extra; long long s_extra;
  int n_points, ldw;
  float *dW, *db;
  const float* absmax;
  const float *rec0, *rec1;     // fp16x2: records [tile][64] of the scaled operands (a0's; b1's), required with absmax
};

// Weight gradients of several layers in one call.  Per job: dW[256][ldw] += sum_p A(p) B(p)^T over one or two
// operand pairs; db[256] += row sums of A of pair 0.  fp16x2: one launch for all jobs; float32: one launch per job.
int svs_wgrad_multi(const svs_wgrad_job* jobs, int n_jobs, int precision, void* hip_stream) {
  if (!jobs || n_jobs < 1) { set_error("svs_wgrad_multi: no jobs"); return SVS_EINVAL; }
  int n_expanded = 0;
  for (int j = 0; j < n_jobs; ++j) n_expanded += jobs[j].b_extra ? 2 : 1;
  if (n_expanded > h2::kMaxJobs) { set_error("svs_wgrad_multi: at most %d jobs (b_extra counts twice)", h2::kMaxJobs); return SVS_EINVAL; }
  for (int j = 0; j < n_jobs; ++j) {
    const svs_wgrad_job& q = jobs[j];
    if (!q.a0 || !q.b0 || !q.dW || q.n_points <= 0 || q.ldw < 256 || q.ldw > 288 || (q.b_extra && q.ldw < 288) || (q.a1 && !q.b1)) {
      set_error("svs_wgrad_multi: bad argument in job %d", j); return SVS_EINVAL;
    }
    if (mlp::is_h2(precision) && q.absmax && (!q.rec0 || (q.a1 && !q.rec1))) {
      set_error("svs_wgrad_multi: job %d has absmax but no scale records", j); return SVS_EINVAL;
    }
  }
  hipStream_t s = (hipStream_t)hip_stream;
  if (mlp::is_h2(precision)) {
    const bool gp = precision == mlp::kFmtF16x2;
    h2::MultiArgs ma;
    int n = 0;
    long long work[h2::kMaxJobs], total = 0;
    for (int j = 0; j < n_jobs; ++j) {
      const svs_wgrad_job& q = jobs[j];
      h2::Job& J = ma.job[n];
      J.p[0] = Pair{q.a0, q.b0, (size_t)q.sa0, (size_t)q.sb0};
      J.rec[0] = q.rec0; J.rec[1] = q.rec1;
      J.n_pairs = 1;
      if (q.a1) { J.p[1] = Pair{q.a1, q.b1, (size_t)q.sa1, (size_t)q.sb1}; J.n_pairs = 2; }
      else { J.p[1] = J.p[0]; J.rec[1] = J.rec[0]; }
      J.n_tiles = (q.n_points + 31) / 32; J.n_valid_points = q.n_points;
      J.b_tiles = 8; J.col0 = 0; J.dW = q.dW; J.ldw = q.ldw; J.db = q.db; J.absmax = q.absmax;
      work[n] = (long long)J.n_tiles * J.n_pairs * 10;
      total += work[n++];
      if (q.b_extra) {
        // the 16 extra B rows (dW columns 256..271) as a narrow job of its own: same A, one 32-row B tile
        h2::Job& X = ma.job[n];
        X = J;
        X.p[0].b = q.b_extra; X.p[0].stride_b = (size_t)q.s_extra; X.n_pairs = 1;
        X.b_tiles = 1; X.col0 = 256; X.db = nullptr;
        // (a narrow item copies 36 of a wide item's 64 KiB and multiplies one B tile of eight, but an item's time is mostly
        // the latency of its copy: priced at 4 / 10 of a wide item, as until round 6, its workgroups were the radiance launch's
        // longest -- 0.33 ms alone; 5 ... 7: 0.29-0.30; 8, 10: 0.30-0.31 (profiles/r06_wgrad_what_bounds_it.txt))
        static const int narrow_work = [] { const char* e = getenv("SVS_WGRAD_NARROW_WORK"); return e ? atoi(e) : 6; }();
        work[n] = (long long)X.n_tiles * narrow_work;
        total += work[n++];
      }
    }
    ma.n_jobs = n;
    static const int touch_env = [] { const char* e = getenv("SVS_WGRAD_TOUCH"); return e ? atoi(e) : 0; }();
    ma.touch = touch_env;
    static const int reverse_env = [] { const char* e = getenv("SVS_WGRAD_REVERSE"); return e ? atoi(e) : 0; }();
    ma.reverse = reverse_env;
    // One workgroup per CU in total, at least one per job and no more than a job has tiles; the rest are handed out one at a
    // time to the job whose workgroups carry the most work (the launch lasts as long as its longest workgroup: with jobs of
    // very different sizes -- a step's two ray groups in one launch -- rounding each share on its own left a small job's
    // single workgroup with a third more tiles than anybody else)
    (void)total;
    const int n_wg_total = 256;
    int count[h2::kMaxJobs];
    int used = 0;
    for (int j = 0; j < n; ++j) { count[j] = 1; ++used; }
    while (used < n_wg_total) {
      int best = -1;
      double load = 0.0;
      for (int j = 0; j < n; ++j) {
        if (count[j] >= ma.job[j].n_tiles) continue;
        const double l = (double)work[j] / count[j];
        if (l > load) { load = l; best = j; }
      }
      if (best < 0) break;
      ++count[best]; ++used;
    }
    int begin = 0;
    for (int j = 0; j < n; ++j) {
      ma.job[j].wg_begin = begin; ma.job[j].wg_count = count[j];
      begin += count[j];
      ma.job[j].ticket = det::Ticket{nullptr, 0u, 0u};
    }
    if (deterministic()) {
      // jobs that add into the same columns of the same accumulator (the ray groups of a step) share a turn counter; their
      // workgroups take tickets in job order, then block order
      int owner[h2::kMaxJobs], n_owner = 0;
      for (int j = 0; j < n; ++j) {
        owner[j] = -1;
        for (int k = 0; k < j; ++k)
          if (ma.job[k].dW == ma.job[j].dW && ma.job[k].col0 == ma.job[j].col0) { owner[j] = owner[k]; break; }
        if (owner[j] < 0) owner[j] = n_owner++;
      }
      unsigned* turn = det::take_slots((unsigned)n_owner);
      if (!turn) { set_error("svs_wgrad_multi: no turn counters (deterministic mode)"); return SVS_EINVAL; }
      unsigned total[h2::kMaxJobs] = {0};
      for (int j = 0; j < n; ++j) { ma.job[j].ticket.turn = turn + owner[j]; ma.job[j].ticket.first = total[owner[j]]; total[owner[j]] += (unsigned)count[j]; }
      for (int j = 0; j < n; ++j) ma.job[j].ticket.total = total[owner[j]];
    }
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(h2::wgrad_h2_multi_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, h2::Ring<false>::kLdsBytes);
    static hipError_t ep = hipFuncSetAttribute(reinterpret_cast<const void*>(h2::wgrad_h2_multi_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, h2::Ring<true>::kLdsBytes);
    if (e != hipSuccess || ep != hipSuccess) {
      set_error("svs_wgrad_multi: hipFuncSetAttribute: %s", hipGetErrorString(e != hipSuccess ? e : ep)); return (int)(e != hipSuccess ? e : ep);
    }
    if (gp) h2::wgrad_h2_multi_kernel<true><<<begin, h2::kThreadsW, h2::Ring<true>::kLdsBytes, s>>>(ma);
    else h2::wgrad_h2_multi_kernel<false><<<begin, h2::kThreadsW, h2::Ring<false>::kLdsBytes, s>>>(ma);
    return check_launch("svs_wgrad_multi");
  }
  if (precision != mlp::kFmtF32) { set_error("svs_wgrad_multi: unknown precision %d", precision); return SVS_EINVAL; }
  for (int j = 0; j < n_jobs; ++j) {
    const svs_wgrad_job& q = jobs[j];
    Args a;
    a.p[0] = Pair{q.a0, q.b0, (size_t)q.sa0, (size_t)q.sb0};
    a.n_pairs = 1;
    if (q.a1) { a.p[1] = Pair{q.a1, q.b1, (size_t)q.sa1, (size_t)q.sb1}; a.n_pairs = 2; }
    a.n_tiles = (q.n_points + 31) / 32;
    a.n_valid_points = q.n_points;
    a.b_extra = q.b_extra; a.stride_extra = (size_t)q.s_extra;
    a.dW = q.dW; a.ldw = q.ldw; a.db = q.db;
    const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
    a.ticket = det::Ticket{det::take_slots(1), 0u, (unsigned)grid};
    if (q.b_extra) {
      constexpr int lds = 2 * 32 * (260 + 292) * 4;
      static hipError_t e9 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<9>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e9 != hipSuccess) { set_error("svs_wgrad_multi: hipFuncSetAttribute: %s", hipGetErrorString(e9)); return (int)e9; }
      wgrad_kernel<9><<<grid, 256, lds, s>>>(a);
    } else {
      constexpr int lds = 2 * 32 * (260 + 260) * 4;
      static hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<8>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e8 != hipSuccess) { set_error("svs_wgrad_multi: hipFuncSetAttribute: %s", hipGetErrorString(e8)); return (int)e8; }
      wgrad_kernel<8><<<grid, 256, lds, s>>>(a);
    }
    if (int rc = check_launch("svs_wgrad_multi")) return rc;
  }
  return SVS_OK;
}

// single job (see svs_wgrad_multi)
int svs_wgrad(const float* a0, const float* b0, long long sa0, long long sb0, const float* a1, const float* b1,
              long long sa1, long long sb1, const float* b_extra, long long s_extra, int n_points, int precision,
              const float* absmax, const float* rec0, const float* rec1, float* dW, int ldw, float* db, void* hip_stream) {
  svs_wgrad_job q{a0, b0, sa0, sb0, a1, b1, sa1, sb1, b_extra, s_extra, n_points, ldw, dW, db, absmax, rec0, rec1};
  return svs_wgrad_multi(&q, 1, precision, hip_stream);
}

}  // extern "C"
