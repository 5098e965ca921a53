// Micro-benchmark: what does ISSUING a 1-KiB vector-memory instruction cost its wave on gfx950, alone and between MFMAs?
// One workgroup of 4 waves per CU (one wave per SIMD, like the fused-MLP kernels).  Each test issues N instructions of a
// kind (global_load_dwordx4, global_store_dwordx4, global_load_lds_dwordx4 = LDS-DMA) and stamps s_memtime around the
// ISSUE (no wait for completion inside the stamps); the "mfma" variants put three 32x32x16 f16 MFMAs (96 cycles of matrix
// core) in front of every instruction and report the time per group, to compare with the bare MFMA groups.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/vmem_issue.hip -o tools/micro/vmem_issue && tools/micro/vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
typedef const __attribute__((address_space(1))) void* gptr;
constexpr int N = 16;

template <int KIND, bool MFMA>
__global__ __launch_bounds__(256, 1) void k(const f32x4* __restrict__ in, f32x4* __restrict__ out, float* res, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4* src = in + ((size_t)blockIdx.x * 4 + wave) * N * 64 + lane;
  f32x4* dst = out + ((size_t)blockIdx.x * 4 + wave) * N * 64 + lane;
  f32x4 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = (f32x4)(float(lane + i));
  f32x16 acc = (f32x16)(0.0f);
  f16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (MFMA) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
    }
    if (KIND == 1) v[i] = __builtin_nontemporal_load(src + i * 64);
    if (KIND == 2) __builtin_nontemporal_store(v[i], dst + i * 64);
    if (KIND == 3) __builtin_amdgcn_global_load_lds((gptr)(src + i * 64), LDSP(smem + (wave * N + i) * 1024), 16, 0, 0);
    if (KIND == 4) { v[i] = v[i] * 1.5f + 2.0f; }                                  // 4 VALU for comparison
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < N; ++i) s += v[i][0] + v[i][3];
  if (KIND == 3) s += reinterpret_cast<float*>(smem)[threadIdx.x];
  for (int r = 0; r < 16; ++r) s += acc[r];
  res[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND, bool MFMA>
void run(const char* name, const f32x4* in, f32x4* out, float* res, unsigned long long* cyc, int blocks) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  for (int rep = 0; rep < 3; ++rep) k<KIND, MFMA><<<blocks, 256, 128 * 1024>>>(in, out, res, cyc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-28s median %6.1f  p10 %6.1f  p90 %6.1f  shader cycles (s_memtime) per instruction / group\n", name, h[h.size() / 2] / double(N),
         h[h.size() / 10] / double(N), h[h.size() * 9 / 10] / double(N));
}

int main() {
  const int blocks = 256;
  f32x4 *in, *out; float* res; unsigned long long* cyc;
  const size_t n = (size_t)blocks * 4 * N * 64;
  hipMalloc(&in, n * 16); hipMalloc(&out, n * 16); hipMalloc(&res, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
  hipMemset(in, 0, n * 16);
  run<0, true>("3 mfma", in, out, res, cyc, blocks);
  run<1, false>("load x4", in, out, res, cyc, blocks);
  run<2, false>("store x4", in, out, res, cyc, blocks);
  run<3, false>("lds-dma x4", in, out, res, cyc, blocks);
  run<4, false>("4 valu", in, out, res, cyc, blocks);
  run<1, true>("3 mfma + load x4", in, out, res, cyc, blocks);
  run<2, true>("3 mfma + store x4", in, out, res, cyc, blocks);
  run<3, true>("3 mfma + lds-dma x4", in, out, res, cyc, blocks);
  run<4, true>("3 mfma + 4 valu", in, out, res, cyc, blocks);
  return 0;
}
