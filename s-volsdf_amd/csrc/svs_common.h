// Shared device/host helpers for the S-VolSDF HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SVS_OK 0
#define SVS_EINVAL (-1)
#define SVS_ESHAPE (-2)

// Activation blocks are written once and read once by a later kernel: non-temporal stores keep them from evicting the
// packed weight stream (2-4.6 MB per network, re-read by every workgroup) from the 4 MB L2 of an XCD.
#ifdef SVS_NO_STREAM_HINTS
#define SVS_STREAM_STORE(v, p) (*(p) = (v))
#define SVS_STREAM_LOAD(p) (*(p))
#else
#define SVS_STREAM_STORE(v, p) __builtin_nontemporal_store((v), (p))
#ifdef SVS_NO_STREAM_LOADS
#define SVS_STREAM_LOAD(p) (*(p))
#else
#define SVS_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#endif
#endif

namespace svs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return SVS_OK;
}

// Row of a 32x32 MFMA accumulator tile held in register r by a lane of half h (h = lane >> 5):
// row = (r & 3) + 8 * (r >> 2) + 4 * h, column = lane & 31  (C/D layout of v_mfma_f32_32x32x2_f32).
// Workgroups are dealt to the 8 XCDs (each with its own L2) round-robin in launch order.  xcd_chunked(i, n): the work item
// workgroup i of n takes so that XCD k processes the contiguous range [k n/8, (k+1) n/8) -- neighbouring tiles, which share
// halo rows, then meet in one L2.  (n not a multiple of 8: identity.)  Used by conv0 (svs_conv_pair.hip: 5-7 % at the
// three stage sizes of config 3); measured neutral for conv2 / prob and harmful for the transposed-convolution GEMMs, whose
// parity classes interleave in the output (conv7 0.025 -> 0.034 ms, conv9 0.049 -> 0.062): not applied there.
__device__ __forceinline__ unsigned xcd_chunked(unsigned i, unsigned n) {
#ifdef SVS_NO_XCD_REMAP
  return i;
#else
  return (n & 7u) ? i : (i & 7u) * (n >> 3) + (i >> 3);
#endif
}

__host__ __device__ constexpr int rho(int r) { return (r & 3) + 8 * (r >> 2); }

// ------------------------------------------------------------------------------------------
// exp / expm1 of the numeric contract: the float32 routines torch-CPU evaluates, restated
// operation by operation -- Sleef 3.x `Sleef_expf8_u10` (xexpf) and `Sleef_expm1f8_u10` (xexpm1f = expk2f(a) - 1 in
// double-float arithmetic), FMA build, sleefsimdsp.c / df.h.  Only IEEE float32 add / mul / fma, so the device returns
// the library's bits (tests: `svs_selftest_exp` against the fixture `primitives`).  The translation units that use
// them are compiled with -ffp-contract=off: every fma below is one the library issues, and no other is formed.
// ------------------------------------------------------------------------------------------
namespace sleef {
constexpr float R_LN2 = 1.4426950216293335f;       // 0x3fb8aa3b
constexpr float L2U = 0.693145751953125f;          // 0x3f317200
constexpr float L2L = 1.428606765330187e-06f;      // 0x35bfbe8e

__device__ __forceinline__ float pow2i(int e) { return __int_as_float((e + 127) << 23); }
// vldexp2_vf_vf_vi2: two multiplications (the second one may round into the denormal range, as the library's does)
__device__ __forceinline__ float ldexp2(float u, int q) {
  const int h = q >> 1;
  return (u * pow2i(h)) * pow2i(q - h);
}
struct df { float x, y; };
__device__ __forceinline__ df add2(float x, float y) {           // dfadd2_vf2_vf_vf
  const float s = x + y, v = s - x;
  return {s, (x - (s - v)) + (y - v)};
}
__device__ __forceinline__ df add2(df x, float y) {              // dfadd2_vf2_vf2_vf
  const df r = add2(x.x, y);
  return {r.x, r.y + x.y};
}
__device__ __forceinline__ df add2(df x, df y) {                 // dfadd2_vf2_vf2_vf2
  const df r = add2(x.x, y.x);
  return {r.x, r.y + (x.y + y.y)};
}
__device__ __forceinline__ df mul(df x, float y) {               // dfmul_vf2_vf2_vf
  const float s = x.x * y;
  return {s, __builtin_fmaf(x.y, y, __builtin_fmaf(x.x, y, -s))};
}
__device__ __forceinline__ df mul(df x, df y) {                  // dfmul_vf2_vf2_vf2
  const float s = x.x * y.x;
  return {s, __builtin_fmaf(x.x, y.y, __builtin_fmaf(x.y, y.x, __builtin_fmaf(x.x, y.x, -s)))};
}
__device__ __forceinline__ df squ(df x) {                        // dfsqu_vf2_vf2
  const float s = x.x * x.x;
  return {s, __builtin_fmaf(x.x + x.x, x.y, __builtin_fmaf(x.x, x.x, -s))};
}
}  // namespace sleef

__device__ __forceinline__ float sleef_expf(float d) {
  if (d != d) return d;
  if (d < -104.0f) return 0.0f;
  if (d > 100.0f) return __builtin_inff();
  const int q = (int)__builtin_rintf(d * sleef::R_LN2);
  const float qf = (float)q;
  float s = __builtin_fmaf(qf, -sleef::L2U, d);
  s = __builtin_fmaf(qf, -sleef::L2L, s);
  float u = 0.00019852761761285365f;
  u = __builtin_fmaf(u, s, 0.0013930435525253415f);
  u = __builtin_fmaf(u, s, 0.008333360776305199f);
  u = __builtin_fmaf(u, s, 0.041666485369205475f);
  u = __builtin_fmaf(u, s, 0.1666666716337204f);
  u = __builtin_fmaf(u, s, 0.5f);
  u = __builtin_fmaf(s * s, u, s) + 1.0f;
  return sleef::ldexp2(u, q);
}

__device__ __forceinline__ float sleef_expm1f(float a) {
  using namespace sleef;
  if (a != a) return a;
  if (a > 88.72283172607421875f) return __builtin_inff();
  if (a < -16.635532379150390625f) return -1.0f;
  if (a == 0.0f) return a;                                      // keeps -0
  const int q = (int)__builtin_rintf((a + 0.0f) * R_LN2);
  const float qf = (float)q;
  df s = add2(df{a, 0.0f}, qf * -L2U);
  s = add2(s, qf * -L2L);
  float u = 0.00019809602235909551f;
  u = __builtin_fmaf(u, s.x, 0.0013942564837634563f);
  u = __builtin_fmaf(u, s.x, 0.008333456702530384f);
  u = __builtin_fmaf(u, s.x, 0.04166637361049652f);
  df t = add2(mul(s, u), 0.1666666567325592f);
  t = add2(mul(s, t), 0.5f);
  t = add2(s, mul(squ(s), t));
  {                                                             // dfadd_vf2_vf_vf2(1, t)
    const float r = 1.0f + t.x;
    t = df{r, ((1.0f - r) + t.x) + t.y};
  }
  t.x = ldexp2(t.x, q);
  t.y = ldexp2(t.y, q);
  const df d = add2(t, -1.0f);
  return d.x + d.y;
}

// torch `x.norm(2, dim=-1)` of a float32 3-vector on CPU: the vectorised reduce kernel contracts acc + x*x into a fused
// multiply-add: sqrt(fma(c, c, fma(b, b, a*a))) (oracle/svs_oracle.py::norm3, checked against torch on 200 000 rows).
__device__ __forceinline__ float norm3(float a, float b, float c) {
  return __builtin_sqrtf(__builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a)));
}

// torch.sum(float32 row, dim=-1) on CPU: ATen cascade_sum (aten/src/ATen/native/cpu/SumKernel.cpp: vectorized_inner_sum ->
// row_sum -> multi_row_sum) in its actual order -- 8-lane vectors (the AVX2 kernel, which also serves AVX-512 hosts),
// 4 independent accumulators, level 0 folded into level 1 every 16 groups (m >= 512), the left-over vectors into
// accumulator 0, the four accumulators left to right, then the scalar tail and the 8 lanes left to right; rows shorter
// than 8 take the scalar path (scalar_inner_sum: the same with 1-lane "vectors").  x: m floats in LDS, visible to the
// wave; the result is returned in every lane.  Lane k*W + l plays lane l of accumulator k.
__device__ __forceinline__ float aten_row_sum(const float* x, int m, int lane) {
  const int W = m < 8 ? 1 : 8;
  const int nv = m / W;
  const int size = nv >> 2;
  const int k = lane / W, l = lane - k * W;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  if (lane < 4 * W) {
    int i = 0;
    while (i + 16 <= size) {
      for (int j = 0; j < 16; ++j, ++i) a0 = a0 + x[((i << 2) + k) * W + l];
      a1 = a1 + a0; a0 = 0.0f;
      if ((i & 0xF0) == 0) {
        a2 = a2 + a1; a1 = 0.0f;
        if ((i & 0xF00) == 0) { a3 = a3 + a2; a2 = 0.0f; }
      }
    }
    for (; i < size; ++i) a0 = a0 + x[((i << 2) + k) * W + l];
    a0 = a0 + a1; a0 = a0 + a2; a0 = a0 + a3;
  }
  float p0 = a0;
  if (lane < W)
    for (int i = size << 2; i < nv; ++i) p0 = p0 + x[i * W + l];
  const float p1 = __shfl(a0, lane + W), p2 = __shfl(a0, lane + 2 * W), p3 = __shfl(a0, lane + 3 * W);
  p0 = ((p0 + p1) + p2) + p3;
  if (W == 1) return __shfl(p0, 0);
  float fin = 0.0f;
  for (int i = nv * 8; i < m; ++i) fin = fin + x[i];
#pragma unroll
  for (int j = 0; j < 8; ++j) fin = fin + __shfl(p0, j);
  return fin;
}

// Laplace density, volsdf/model/density.py:21-26, float32 op order of the reference.
__device__ __forceinline__ float laplace_density(float sdf, float beta) {
  const float alpha = 1.0f / beta;
  const float sgn = (sdf > 0.0f) ? 1.0f : ((sdf < 0.0f) ? -1.0f : 0.0f);
  const float e = sleef_expm1f(-__builtin_fabsf(sdf) / beta);
  return alpha * (0.5f + (0.5f * sgn) * e);
}

}  // namespace svs
