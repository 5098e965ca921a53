// Ordered compaction offsets of a byte mask: offset[i] = number of non-zero mask bytes before i, *count = their total.
// Three launches over blocks of 4096 bytes (256 threads x 16): per-block counts, one workgroup turning them into bases,
// per-block offsets.  The per-block counts / bases live in offset[4096 b] (the block's own first slot, which the third
// launch overwrites last), so no workspace beyond the n ints of `offset` is needed.  Used by the depth-fusion point list
// (svs_fusion.hip, runner.py:377-386) and the evaluator's compaction (svs_cloud.hip): both keep the row-major order.
// (A single workgroup walking the whole mask -- the first version -- took 3.5 ms for a 1200 x 1600 mask.)
#pragma once
#include "svs_common.h"

namespace svs {
namespace scan {

constexpr int kItems = 16, kThreads = 256, kBlock = kItems * kThreads;

// the 16 flags of thread t of block b (zero beyond n)
__device__ __forceinline__ void load_flags(const uint8_t* __restrict__ mask, long long n, long long i0, bool aligned, int (&f)[kItems]) {
  if (aligned && i0 + kItems <= n) {
    const uint4 v = *reinterpret_cast<const uint4*>(mask + i0);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < kItems; ++k) f[k] = ((w[k >> 2] >> (8 * (k & 3))) & 0xffu) != 0;
  } else {
#pragma unroll
    for (int k = 0; k < kItems; ++k) f[k] = (i0 + k < n) && mask[i0 + k] != 0;
  }
}

// exclusive prefix of `v` over the 256 threads of the block (and the block total)
__device__ __forceinline__ int block_exclusive(int v, int& total) {
  __shared__ int wsum[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) {
    if (w < wave) base += wsum[w];
    total += wsum[w];
  }
  __syncthreads();
  return base + inc - v;
}

static __global__ __launch_bounds__(kThreads) void block_count_kernel(const uint8_t* __restrict__ mask, long long n, bool aligned, int* __restrict__ offset) {
  const long long i0 = (long long)blockIdx.x * kBlock + kItems * threadIdx.x;
  int f[kItems];
  load_flags(mask, n, i0, aligned, f);
  int s = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) s += f[k];
  int total;
  block_exclusive(s, total);
  if (threadIdx.x == 0) offset[(long long)blockIdx.x * kBlock] = total;
}

constexpr int kBaseThreads = 1024;
static __global__ __launch_bounds__(kBaseThreads) void block_base_kernel(int* __restrict__ offset, int n_blocks, int* __restrict__ count) {
  __shared__ int part[kBaseThreads];
  const int t = threadIdx.x;
  const int per = (n_blocks + kBaseThreads - 1) / kBaseThreads;
  const int lo = t * per < n_blocks ? t * per : n_blocks, hi = lo + per < n_blocks ? lo + per : n_blocks;
  int s = 0;
  for (int b = lo; b < hi; ++b) s += offset[(long long)b * kBlock];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < kBaseThreads; d <<= 1) {
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int b = lo; b < hi; ++b) {
    const int c = offset[(long long)b * kBlock];
    offset[(long long)b * kBlock] = run;
    run += c;
  }
  if (t == kBaseThreads - 1) *count = part[t];
}

static __global__ __launch_bounds__(kThreads) void block_offsets_kernel(const uint8_t* __restrict__ mask, long long n, bool aligned,
                                                                 int* __restrict__ offset) {
  __shared__ int base_s;
  const long long b0 = (long long)blockIdx.x * kBlock, i0 = b0 + kItems * threadIdx.x;
  if (threadIdx.x == 0) base_s = offset[b0];
  int f[kItems];
  load_flags(mask, n, i0, aligned, f);
  int s = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) s += f[k];
  int total;
  int run = block_exclusive(s, total);           // (its barriers also publish base_s and order the read of offset[b0] first)
  run += base_s;
  if (aligned && i0 + kItems <= n) {
    int o[kItems];
#pragma unroll
    for (int k = 0; k < kItems; ++k) { o[k] = run; run += f[k]; }
#pragma unroll
    for (int q = 0; q < kItems / 4; ++q)
      *reinterpret_cast<int4*>(offset + i0 + 4 * q) = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      if (i0 + k < n) offset[i0 + k] = run;
      run += f[k];
    }
  }
}

// enqueue the three launches; n == 0 leaves *count = 0
static inline void mask_offsets(const uint8_t* mask, long long n, int* offset, int* count, hipStream_t s) {
  if (n <= 0) { (void)hipMemsetAsync(count, 0, sizeof(int), s); return; }
  const int n_blocks = (int)((n + kBlock - 1) / kBlock);
  const bool aligned = ((uintptr_t)mask % 16 == 0) && ((uintptr_t)offset % 16 == 0);
  block_count_kernel<<<n_blocks, kThreads, 0, s>>>(mask, n, aligned, offset);
  block_base_kernel<<<1, kBaseThreads, 0, s>>>(offset, n_blocks, count);
  block_offsets_kernel<<<n_blocks, kThreads, 0, s>>>(mask, n, aligned, offset);
}

}  // namespace scan
}  // namespace svs
