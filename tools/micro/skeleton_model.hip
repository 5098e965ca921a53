// The skeleton of the fused MLP kernels (svs_mlp_h2_dev.h: one wave per SIMD, a tile = 16 k-steps of 3 MFMAs, A fragments read
// from an LDS ring one k-step ahead, the next 32 KB chunk of weights fetched by LDS-DMA behind the first k-steps, a counted
// vmcnt + s_barrier per tile, side tiles loaded / stored around it) with its parts switched on one at a time: cycles per
// k-step (s_memtime).  MFMA alone would be 98.      hipcc -O3 --offload-arch=gfx950 tools/micro/skeleton_model.hip -o ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kChunkF4 = 2048;      // 32 KB
constexpr int kTiles = 64;          // 8 layers x 8 tiles per "kernel"

#define VALU(x, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))
// (MFMAs as volatile assembly as well: LLVM moves the pure intrinsic across the volatile VALU slices otherwise)
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define VEXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define VLOG(x) asm volatile("v_log_f32 %0, %0" : "+v"(x))

enum { LDSREAD = 1, DMA = 2, BARRIER = 4, VALU16 = 8, SIDE = 16, STORES = 32, VALU8 = 64, SOFTPLUS = 128, SOFTPLUS_IND = 256 };

template <int F>
__global__ __launch_bounds__(256, 1) void k(const f32x4* __restrict__ w, const f32x4* __restrict__ side, f32x4* __restrict__ dst,
                                            float* out, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4* buf = reinterpret_cast<f32x4*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  f16x8 bh[16], bm[16];
  for (int s = 0; s < 16; ++s)
    for (int j = 0; j < 8; ++j) { bh[s][j] = (_Float16)(0.001f * (lane + j + s)); bm[s][j] = (_Float16)(1e-6f * (lane - j)); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 0.5f + lane * 0.001f + j;
  const float c0 = 0.999f, c1 = 0.0001f;
  for (int i = threadIdx.x; i < 2 * kChunkF4; i += 256) buf[i] = w[i];
  __syncthreads();
  const f32x4* sp = side + ((size_t)blockIdx.x * 4 + wave) * kTiles * 12 * 64 + lane;
  f32x4* dp = dst + ((size_t)blockIdx.x * 4 + wave) * kTiles * 4 * 64 + lane;
  f32x4 sa[12], sb[12];      // side tiles: two generations in flight
  for (int i = 0; i < 12; ++i) { sa[i] = f32x4{0, 0, 0, 0}; sb[i] = sa[i]; }
  f32x16 keep = {0};
  int cur = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    const f32x4* g = w;
#pragma unroll 1
    for (int t = 0; t < kTiles; t += 2) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {      // two tiles per loop body so that the side generations alternate statically
        f32x4* sn = tt ? sb : sa;            // loaded during this tile, consumed at the top of the tile after next
        f32x4* sc = tt ? sa : sb;            // loaded during the previous tile: untouched here
        (void)sc;
        const f16x8* a_ptr = reinterpret_cast<const f16x8*>(buf + cur * kChunkF4) + lane;
        f32x16 acc = {0};
        f16x8 ah, am;
        if (F & LDSREAD) { ah = a_ptr[0]; am = a_ptr[64]; } else { ah = bh[0]; am = bm[0]; }
        if (F & SIDE) {      // first use of the tile loaded two tiles ago (same generation array), at the top of the tile
#pragma unroll
          for (int i = 0; i < 12; ++i) v[i & 7] += sn[i][0];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          f16x8 nh, nm;
          __builtin_amdgcn_sched_barrier(0);
          MFMA(acc, am, bh[s]);
          __builtin_amdgcn_sched_barrier(0);
          if ((F & LDSREAD) && s + 1 < 16) { nh = a_ptr[(2 * s + 2) * 64]; nm = a_ptr[(2 * s + 3) * 64]; }
          if (F & (VALU16 | VALU8)) {
#pragma unroll
            for (int j = 0; j < ((F & VALU16) ? 8 : 4); ++j) VALU(v[j], c0, c1);
          }
          if (F & SOFTPLUS) { VALU(v[0], c0, c1); VALU(v[0], c0, c1); VEXP(v[0]); }                       // dependent chain, as the kernel
          if (F & SOFTPLUS_IND) { VALU(v[0], c0, c1); VALU(v[1], c0, c1); VEXP(v[2]); }                   // the same mix, independent
          __builtin_amdgcn_sched_barrier(0);
          MFMA(acc, ah, bm[s]);
          __builtin_amdgcn_sched_barrier(0);
          if (F & (VALU16 | VALU8)) {
#pragma unroll
            for (int j = 0; j < ((F & VALU16) ? 8 : 4); ++j) VALU(v[j], c0, c1);
          }
          if (F & SOFTPLUS) { VALU(v[0], c0, c1); VLOG(v[0]); VALU(v[1], c0, c1); }
          if (F & SOFTPLUS_IND) { VALU(v[3], c0, c1); VLOG(v[4]); VALU(v[5], c0, c1); }
          __builtin_amdgcn_sched_barrier(0);
          MFMA(acc, ah, bh[s]);
          __builtin_amdgcn_sched_barrier(0);
          if (F & SOFTPLUS) VALU(v[1], v[0], c1);
          if (F & SOFTPLUS_IND) VALU(v[6], c0, c1);
          if ((F & DMA) && s < 8) {
            const int idx = s * 256 + wb;
            const unsigned lane_bytes = lane * 16u;
            const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(buf + (cur ^ 1) * kChunkF4 + idx);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_bytes), "s"(g + idx), "s"(lds_base) : "memory");
          }
          if ((F & STORES) && (s == 9 || s == 15)) {
            const int q = (t + tt) * 4 + (s == 15 ? 2 : 0);
            __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, dp + q * 64);
            __builtin_nontemporal_store(f32x4{v[4], v[5], v[6], v[7]}, dp + (q + 1) * 64);
          }
          if ((F & SIDE) && s >= 10) {
            const int q = ((t + tt) * 12 + (s - 10) * 2);
            sn[(s - 10) * 2] = __builtin_nontemporal_load(sp + q * 64);
            sn[(s - 10) * 2 + 1] = __builtin_nontemporal_load(sp + (q + 1) * 64);
          }
          if ((F & LDSREAD) && s + 1 < 16) { ah = nh; am = nm; } else if (s + 1 < 16) { ah = bh[s + 1]; am = bm[s + 1]; }
        }
        g += kChunkF4;
        if (g >= w + 64 * kChunkF4) g = w;
#pragma unroll
        for (int i = 0; i < 16; ++i) keep[i] += acc[i];
        if (F & BARRIER) {
          constexpr int N = ((F & SIDE) ? 12 : 0) + ((F & STORES) ? 4 : 0);
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
        }
        cur ^= 1;
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float sum = 0;
  for (int j = 0; j < 8; ++j) sum += v[j];
  for (int j = 0; j < 16; ++j) sum += keep[j];
  for (int i = 0; i < 12; ++i) sum += sa[i][1] + sb[i][1];
  if (sum == 12345.678f) out[0] = sum;
  if (lane == 0) out[1 + blockIdx.x * 4 + wave] = (float)(t1 - t0) / (reps * kTiles * 16.0f);
}

template <int F>
void run(const char* what, const f32x4* w, const f32x4* side, f32x4* dst, float* d, int blocks) {
  const int reps = 4;
  const size_t lds = 100 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<F><<<blocks, 256, lds>>>(w, side, dst, d, 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<F><<<blocks, 256, lds>>>(w, side, dst, d, reps);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h(1 + blocks * 4);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  double m = 0; for (size_t i = 1; i < h.size(); ++i) m += h[i];
  m /= (h.size() - 1);
  printf("%-78s cycles per k-step %.1f  (%.0f MHz, %.3f ms)\n", what, m, m * reps * kTiles * 16 / (ms * 1e3), ms);
}

int main() {
  const int blocks = 256;
  float* d; hipMalloc(&d, 4 * (1 + 1024 * 4));
  f32x4 *w, *side, *dst;
  hipMalloc(&w, 64 * kChunkF4 * 16); hipMemset(w, 0, 64 * kChunkF4 * 16);
  const size_t side_b = (size_t)blocks * 4 * kTiles * 12 * 64 * 16, dst_b = (size_t)blocks * 4 * kTiles * 4 * 64 * 16;
  hipMalloc(&side, side_b); hipMemset(side, 0, side_b);
  hipMalloc(&dst, dst_b);
  run<0>("MFMA (operands in registers)", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER | DMA | SOFTPLUS>("skeleton + softplus slices (3 | 3 | 1 VALU, exp and log, dependent)", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER | DMA | SOFTPLUS_IND>("skeleton + the same instruction mix, independent", w, side, dst, d, blocks);
  run<SOFTPLUS>("MFMA from registers + softplus slices", w, side, dst, d, blocks);
  run<LDSREAD>("+ A fragments from LDS, one k-step ahead", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER>("+ barrier per tile", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER | DMA>("+ LDS-DMA of the next chunk (8 pieces)", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER | DMA | VALU8>("+ 8 v_fma per k-step", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER | DMA | VALU16>("+ 16 v_fma per k-step", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER | DMA | VALU16 | STORES>("+ 16 v_fma, 4 stores per tile", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER | DMA | VALU16 | SIDE>("+ 16 v_fma, 12 side loads per tile", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER | DMA | VALU16 | SIDE | STORES>("+ 16 v_fma, 12 side loads, 4 stores per tile", w, side, dst, d, blocks);
  run<LDSREAD | BARRIER | DMA | SIDE | STORES>("no VALU: DMA, 12 side loads, 4 stores per tile", w, side, dst, d, blocks);
  run<BARRIER | DMA>("MFMA from registers + barrier + DMA (no LDS reads)", w, side, dst, d, blocks);
  run<VALU16>("MFMA from registers + 16 v_fma", w, side, dst, d, blocks);
  run<LDSREAD | VALU16>("MFMA, LDS reads, 16 v_fma (no barrier, no DMA)", w, side, dst, d, blocks);
  return 0;
}
