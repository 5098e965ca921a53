// Micro-benchmark: does the activation-block layout matter for HBM throughput?
// Every wave streams L blocks of 32 KiB (read, then write another buffer), all waves walking l = 0..L-1 together.
//   layout A (current): [wave tile][layer][block]   -> concurrent accesses are 256 KiB apart
//   layout B          : [layer][wave tile][block]   -> concurrent accesses are contiguous
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlockF4 = 2048;   // 32 KiB
template <int LAYOUT, int NT>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int n_tiles, int L) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wt = blockIdx.x * 4 + wave;
  if (wt >= n_tiles) return;
  for (int l = 0; l < L; ++l) {
    const size_t blk = LAYOUT == 0 ? (size_t)wt * L + l : (size_t)l * n_tiles + wt;
    const f32x4* s = in + blk * kBlockF4 + lane;
    f32x4* d = out + blk * kBlockF4 + lane;
    f32x4 v[8];
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = NT ? __builtin_nontemporal_load(s + (8 * t + i) * 64) : s[(8 * t + i) * 64];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4 w = v[i] * 1.0001f;
        if (NT) __builtin_nontemporal_store(w, d + (8 * t + i) * 64); else d[(8 * t + i) * 64] = w;
      }
    }
  }
}
template <int LAYOUT, int NT>
float run(const f32x4* in, f32x4* out, int n_tiles, int L) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  stream_kernel<LAYOUT, NT><<<(n_tiles + 3) / 4, 256>>>(in, out, n_tiles, L);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) stream_kernel<LAYOUT, NT><<<(n_tiles + 3) / 4, 256>>>(in, out, n_tiles, L);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  const int n_tiles = 3200, L = 8;
  const size_t bytes = (size_t)n_tiles * L * kBlockF4 * 16;
  f32x4 *in, *out; hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMemset(in, 0, bytes);
  for (int rep = 0; rep < 2; ++rep) {
    printf("layout A plain %.1f GB/s | nt %.1f GB/s\n", 2 * bytes / run<0, 0>(in, out, n_tiles, L) / 1e6, 2 * bytes / run<0, 1>(in, out, n_tiles, L) / 1e6);
    printf("layout B plain %.1f GB/s | nt %.1f GB/s\n", 2 * bytes / run<1, 0>(in, out, n_tiles, L) / 1e6, 2 * bytes / run<1, 1>(in, out, n_tiles, L) / 1e6);
  }
  return 0;
}
