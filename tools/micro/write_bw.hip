// What a WRITE-ONLY stream reaches on this chip (the cost-volume producer writes 503 MB and reads 24: its roof is the write
// rate, not the 8 TB/s read + write peak):   hipcc -O3 --offload-arch=gfx950 tools/micro/write_bw.hip -o /tmp/wbw && /tmp/wbw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: float4 stores, 1: non-temporal float4, 2: uint2 (8 B per lane) stores, 3: non-temporal uint2
__global__ __launch_bounds__(256) void fill(float* __restrict__ out, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  if (MODE < 2) {
    f32x4* o = reinterpret_cast<f32x4*>(out);
    const f32x4 v = {1.0f, 2.0f, 3.0f, (float)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
      if (MODE == 1) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    }
  } else {
    u32x2* o = reinterpret_cast<u32x2*>(out);
    const u32x2 v = {1u, blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * n16; i += stride) {
      if (MODE == 3) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    }
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void copy(const float* __restrict__ in, float* __restrict__ out, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  const f32x4* s = reinterpret_cast<const f32x4*>(in);
  f32x4* o = reinterpret_cast<f32x4*>(out);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) o[i] = s[i];
}

template <typename F> static float time_ms(F f, int reps = 20) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t bytes = 503316480;              // the stage-1 split volume of config 3
  const size_t n16 = bytes / 16;
  float *a, *b;
  (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
  (void)hipMemset(a, 0, bytes);
  for (int grid : {1024, 4096, 16384}) {
    const float t0 = time_ms([&] { fill<0><<<grid, 256>>>(b, n16); });
    const float t1 = time_ms([&] { fill<1><<<grid, 256>>>(b, n16); });
    const float t2 = time_ms([&] { fill<2><<<grid, 256>>>(b, n16); });
    const float t3 = time_ms([&] { fill<3><<<grid, 256>>>(b, n16); });
    const float tc = time_ms([&] { copy<0><<<grid, 256>>>(a, b, n16); });
    printf("grid %5d: write-only 503 MB  float4 %.3f ms = %.2f TB/s | nt float4 %.3f = %.2f | uint2 %.3f = %.2f | nt uint2 %.3f = %.2f || copy (r+w) %.3f ms = %.2f TB/s\n",
           grid, t0, bytes / t0 / 1e9, t1, bytes / t1 / 1e9, t2, bytes / t2 / 1e9, t3, bytes / t3 / 1e9, tc, 2.0 * bytes / tc / 1e9);
  }
  const float tm = time_ms([&] { (void)hipMemsetAsync(b, 0, bytes, 0); });
  printf("hipMemsetAsync 503 MB: %.3f ms = %.2f TB/s\n", tm, bytes / tm / 1e9);
  return 0;
}
