// Per-point power-of-two scaling shared by the fp16x2 backward sweeps (svs_mlp_bwd_h2.hip, svs_bg_h2.hip); see the
// header of svs_mlp_bwd_h2.hip.
#pragma once
#include "svs_mlp_h2_dev.h"
#include "svs_blocks_h2.h"

namespace svs {
namespace mlp {

struct PointScale {
  float s_in, inv_in;   // the operand being consumed holds true * s_in
  float s_out;          // the operand being produced is split as true * s_out
  float m;              // running max |true| of the operand being produced (this lane's rows)
  float floor_m;        // lower bound of the maxima
  float gmax;           // max over everything tracked so far

  // s * mx in [2^4, 2^5)
  static __device__ __forceinline__ float pow2_for(float mx) {
    int e = (int)((__float_as_uint(mx) >> 23) & 0xff);
    e = e < 24 ? 24 : (e > 230 ? 230 : e);
    return __uint_as_float((unsigned)(258 - e) << 23);
  }
  static __device__ __forceinline__ float inv_pow2(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }
  // m0: max |true| of the first operand over the whole column (both lane halves)
  __device__ __forceinline__ void start(float m0, float fl) {
    floor_m = fl; gmax = m0;
    s_in = pow2_for(__builtin_fmaxf(m0, fl)); inv_in = inv_pow2(s_in);
    s_out = s_in; m = 0.0f;
  }
  // the first operand comes from a stored half block (svs_blocks_h2.h): its scale is the stored one; m0 its maximum
  __device__ __forceinline__ void start_stored(float s_stored, float m0, float fl) {
    floor_m = fl; gmax = m0;
    s_in = s_stored; inv_in = inv_pow2(s_in);
    s_out = pow2_for(__builtin_fmaxf(m0, fl)); m = 0.0f;
  }
  __device__ __forceinline__ void track(float v) { m = __builtin_fmaxf(m, __builtin_fabsf(v)); }
  // the operand just produced becomes the one consumed
  __device__ __forceinline__ void next() {
    m = __builtin_fmaxf(m, __shfl_xor(m, 32));
    gmax = __builtin_fmaxf(gmax, m);
    s_in = s_out; inv_in = inv_pow2(s_in);
    s_out = pow2_for(__builtin_fmaxf(m, floor_m));
    m = 0.0f;
  }
};

__device__ __forceinline__ void publish_max(float* slot, float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, d));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(v));
}

__device__ __forceinline__ void split_tile_scaled(const f32x16& y, int t, Pieces2& p, float s) {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = y[r] * s;
  split_tile(v, t, p);
}

}  // namespace mlp
}  // namespace svs
