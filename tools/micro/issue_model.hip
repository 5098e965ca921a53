// What one wave per SIMD can overlap on gfx950: cycles per loop body (s_memtime) for MFMA chains, VALU, transcendental and
// v_fma_mix work, alone and interleaved.  Answers whether the fused MLP kernels (one wave per SIMD, in-order issue) hide their
// epilogue arithmetic behind the MFMAs.     hipcc -O3 --offload-arch=gfx950 tools/micro/issue_model.hip -o /tmp/issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VALU(x, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))
#define VEXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define VRCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x))
#define VMIX(x, h, c) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(h), "v"(c))
#define VPK(x, a) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define VCVT(x, h) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x) : "v"(h))
#define ACCRD(x, acc) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc))

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
  f32x16 acc0 = {0}, acc1 = {0};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 0.5f + lane * 0.001f + j;
  double pk[4] = {1.0, 2.0, 3.0, 4.0};
  const float c0 = 0.999f, c1 = 0.0001f;
  unsigned hw = 0x3c003c00u + lane;
  lds[threadIdx.x * 4] = lane;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      constexpr bool MF = MODE == 0 || MODE == 1 || MODE == 2 || MODE == 3 || MODE == 7 || MODE == 8 || MODE == 10 || MODE == 11 || MODE == 12;
      constexpr int NV = MODE == 1 ? 8 : MODE == 2 ? 16 : MODE == 3 ? 24 : MODE == 4 ? 16 : MODE == 7 ? 12 : MODE == 11 ? 32 : MODE == 12 ? 16 : 0;
      constexpr int NE = MODE == 6 ? 4 : MODE == 7 ? 2 : 0;
      // three MFMAs with the VALU work in the two gaps between them (as the kernels' epilogue slices sit)
      if (MF) {
        if (MODE == 8 && (s & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < NV / 2; ++j) VALU(v[j & 7], c0, c1);
#pragma unroll
      for (int j = 0; j < NE / 2; ++j) VEXP(v[j & 7]);
      if (MODE == 12) { float x; ACCRD(x, acc1[s]); v[0] += x; }
      __builtin_amdgcn_sched_barrier(0);
      if (MF) {
        if (MODE == 8 && (s & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = NV / 2; j < NV; ++j) VALU(v[j & 7], c0, c1);
#pragma unroll
      for (int j = NE / 2; j < NE; ++j) VEXP(v[4 + (j & 3)]);
      if (MODE == 5) {
#pragma unroll
        for (int j = 0; j < 16; ++j) VMIX(v[j & 7], hw, c0);
      }
      if (MODE == 9) {
#pragma unroll
        for (int j = 0; j < 16; ++j) VPK(pk[j & 3], pk[(j + 1) & 3]);
      }
      if (MODE == 13) {
#pragma unroll
        for (int j = 0; j < 16; ++j) VCVT(v[j & 7], hw);
      }
      if (MODE == 14) {
#pragma unroll
        for (int j = 0; j < 4; ++j) VRCP(v[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MF) {
        if (MODE == 8 && (s & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 10) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 r0 = *reinterpret_cast<f32x4*>(&lds[(lane * 4 + s * 256) & 8191]);
        f32x4 r1 = *reinterpret_cast<f32x4*>(&lds[(lane * 4 + s * 256 + 1024) & 8191]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        v[0] += r0[0] + r1[1];
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float sum = 0;
  for (int j = 0; j < 8; ++j) sum += v[j];
  for (int j = 0; j < 16; ++j) sum += acc0[j] + acc1[j];
  for (int j = 0; j < 4; ++j) sum += (float)pk[j];
  if (sum == 12345.678f) out[0] = sum;
  if (lane == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = (float)(t1 - t0) / (iters * 8.0f);
}

template <int MODE>
void run(const char* what, float* d, int waves_per_simd) {
  const int iters = 2000, blocks = 256 * waves_per_simd;
  const size_t lds = waves_per_simd == 1 ? 100 * 1024 : 32 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<MODE><<<blocks, 256, lds>>>(d, 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256, lds>>>(d, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h(1 + blocks * 4);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  double m = 0; for (size_t i = 1; i < h.size(); ++i) m += h[i];
  m /= (h.size() - 1);
  printf("%-52s waves/SIMD %d  cycles per body %.1f  (%.0f MHz)\n", what, waves_per_simd, m, m * iters * 8 / (ms * 1e3));
}

int main() {
  float* d; hipMalloc(&d, 4 * (1 + 1024 * 4));
  for (int w = 1; w <= 2; ++w) {
    run<0>("3 MFMA 32x32x16 f16 (one accumulator)", d, w);
    run<8>("3 MFMA, two accumulators alternating", d, w);
    run<4>("16 v_fma_f32", d, w);
    run<1>("3 MFMA + 8 v_fma_f32", d, w);
    run<2>("3 MFMA + 16 v_fma_f32", d, w);
    run<3>("3 MFMA + 24 v_fma_f32", d, w);
    run<11>("3 MFMA + 32 v_fma_f32", d, w);
    run<12>("3 MFMA + 16 v_fma_f32 + accvgpr_read", d, w);
    run<5>("16 v_fma_mix_f32", d, w);
    run<13>("16 v_cvt_f32_f16_sdwa", d, w);
    run<9>("16 v_pk_mul_f32", d, w);
    run<6>("4 v_exp_f32", d, w);
    run<14>("4 v_rcp_f32", d, w);
    run<7>("3 MFMA + 12 v_fma_f32 + 2 v_exp_f32", d, w);
    run<10>("3 MFMA + 2 ds_read_b128 + lgkmcnt(0)", d, w);
  }
  return 0;
}
