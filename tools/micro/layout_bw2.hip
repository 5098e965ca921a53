// Micro-benchmark (round 4): which activation-block layout streams best?  Every wave walks L blocks of 32 KiB in 8 tile steps
// of 4 KiB (read from `in`, write to `out`), all waves walking (l, t) together -- the access pattern of the backward sweeps.
//   B  (current): [layer][wave tile][hi plane 16 KiB | mid plane 16 KiB]; a tile step = 2 KiB of each plane (16 KiB apart)
//   C           : [layer][tile step][wave tile][4 KiB]: the waves' pieces of one step are contiguous (a copy kernel's footprint)
//   D           : [layer][wave tile][tile step][4 KiB]: a wave's 32 KiB contiguous, pieces of one step in one run
//   copy        : plain grid-stride copy of the same bytes
// DEPTH = tile steps of loads in flight per wave before the first store (1 or 2).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlockF4 = 2048;   // 32 KiB
template <int LAYOUT>
__device__ __forceinline__ size_t piece(int l, int wt, int t, int half, int n_tiles) {   // offset in f32x4 of a 2 KiB half piece
  if (LAYOUT == 0) return ((size_t)l * n_tiles + wt) * kBlockF4 + (size_t)half * 1024 + (size_t)t * 128;
  if (LAYOUT == 1) return (((size_t)l * 8 + t) * n_tiles + wt) * 256 + (size_t)half * 128;
  return (((size_t)l * n_tiles + wt) * 8 + t) * 256 + (size_t)half * 128;
}
template <int LAYOUT, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int n_tiles, int L) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wt = blockIdx.x * 4 + wave;
  if (wt >= n_tiles) return;
  f32x4 v[DEPTH][4];
  const int steps = L * 8;
  for (int s = 0; s < DEPTH - 1 && s < steps; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[s % DEPTH][i] = __builtin_nontemporal_load(in + piece<LAYOUT>(s >> 3, wt, s & 7, i >> 1, n_tiles) + (i & 1) * 64 + lane);
  for (int s = 0; s < steps; ++s) {
    const int sn = s + DEPTH - 1;
    if (sn < steps)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[sn % DEPTH][i] = __builtin_nontemporal_load(in + piece<LAYOUT>(sn >> 3, wt, sn & 7, i >> 1, n_tiles) + (i & 1) * 64 + lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_nontemporal_store(v[s % DEPTH][i] * 1.0001f, out + piece<LAYOUT>(s >> 3, wt, s & 7, i >> 1, n_tiles) + (i & 1) * 64 + lane);
  }
}
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i) * 1.0001f, out + i);
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  const int n_tiles = 3200, L = 8;
  const size_t bytes = (size_t)n_tiles * L * kBlockF4 * 16;
  f32x4 *in, *out; hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMemset(in, 0, bytes);
  const int g = (n_tiles + 3) / 4;
  for (int rep = 0; rep < 2; ++rep) {
    printf("B depth1 %.0f depth2 %.0f | C depth1 %.0f depth2 %.0f | D depth1 %.0f depth2 %.0f | copy %.0f %.0f GB/s\n",
           2 * bytes / timeit([&] { stream_kernel<0, 1><<<g, 256>>>(in, out, n_tiles, L); }) / 1e6,
           2 * bytes / timeit([&] { stream_kernel<0, 2><<<g, 256>>>(in, out, n_tiles, L); }) / 1e6,
           2 * bytes / timeit([&] { stream_kernel<1, 1><<<g, 256>>>(in, out, n_tiles, L); }) / 1e6,
           2 * bytes / timeit([&] { stream_kernel<1, 2><<<g, 256>>>(in, out, n_tiles, L); }) / 1e6,
           2 * bytes / timeit([&] { stream_kernel<2, 1><<<g, 256>>>(in, out, n_tiles, L); }) / 1e6,
           2 * bytes / timeit([&] { stream_kernel<2, 2><<<g, 256>>>(in, out, n_tiles, L); }) / 1e6,
           2 * bytes / timeit([&] { copy_kernel<<<2048, 256>>>(in, out, bytes / 16); }) / 1e6,
           2 * bytes / timeit([&] { copy_kernel<<<8192, 256>>>(in, out, bytes / 16); }) / 1e6);
  }
  // one wave per SIMD like the fused kernels (they hold 512 registers): launch bounds via 64 KB of dynamic LDS per workgroup
  return 0;
}
