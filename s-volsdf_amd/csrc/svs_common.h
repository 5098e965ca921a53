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
// Deterministic exp / expm1 (the numeric contract of DESIGN.md): float64 Cody-Waite reduction
// and a degree-13 Horner polynomial using IEEE add/mul only (no fma), one rounding to float32.
// The sampler / compositing translation units are compiled with -ffp-contract=off.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double det_exp64(double x) {
  const double k = __builtin_rint(x * 1.4426950408889634);
  const double r = (x - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
  double p = 1.6059043836821613e-10;
  p = p * r + 2.0876756987868100e-09;
  p = p * r + 2.5052108385441720e-08;
  p = p * r + 2.7557319223985888e-07;
  p = p * r + 2.7557319223985893e-06;
  p = p * r + 2.4801587301587302e-05;
  p = p * r + 1.9841269841269841e-04;
  p = p * r + 1.3888888888888889e-03;
  p = p * r + 8.3333333333333332e-03;
  p = p * r + 4.1666666666666664e-02;
  p = p * r + 1.6666666666666666e-01;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  const long long bits = ((long long)((int)k + 1023)) << 52;   // 2^k, k in [-151, 129]
  return p * __longlong_as_double(bits);
}

__device__ __forceinline__ float det_exp(float xf) {
  const double x = (double)xf;
  if (x != x) return xf;
  if (x > 88.72283935546875) return __builtin_inff();
  if (x < -104.0) return 0.0f;
  return (float)det_exp64(x);
}

__device__ __forceinline__ float det_expm1(float xf) {
  const double x = (double)xf;
  if (x != x) return xf;
  if (x > 88.72283935546875) return __builtin_inff();
  if (x < -104.0) return -1.0f;
  const double ax = x < 0 ? -x : x;
  if (ax < 9.5367431640625e-07) return (float)(x + (x * x) * 0.5);
  return (float)(det_exp64(x) - 1.0);
}

// Laplace density, volsdf/model/density.py:21-26, float32 op order of the reference.
__device__ __forceinline__ float laplace_density(float sdf, float beta) {
  const float alpha = 1.0f / beta;
  const float sgn = (sdf > 0.0f) ? 1.0f : ((sdf < 0.0f) ? -1.0f : 0.0f);
  const float e = det_expm1(-__builtin_fabsf(sdf) / beta);
  return alpha * (0.5f + (0.5f * sgn) * e);
}

}  // namespace svs
