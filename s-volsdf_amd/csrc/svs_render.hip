// Ray set-up and alpha compositing for gfx950.
//
// Reference: volsdf/utils/rend_util.py:60-95,143-156 (get_camera_params, lift), volsdf/model/network.py:213-222
// (ray set-up), :281-295 (volume_rendering) and :237-256,270-276 (the weighted reductions).
// Compiled with -ffp-contract=off: same numeric contract as the sampler (svs_sampler.hip).
#include "svs_common.h"

namespace svs {
namespace render {

// ---- a1: pixel -> world ray ---------------------------------------------------------------------------
__global__ void rays_kernel(const float* __restrict__ uv, const float* __restrict__ pose, const float* __restrict__ K,
                            int R, float* __restrict__ dirs, float* __restrict__ cam, float* __restrict__ depth_scale) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) { cam[0] = pose[3]; cam[1] = pose[7]; cam[2] = pose[11]; }
  if (r >= R) return;
  const float fx = K[0], sk = K[1], cx = K[2], fy = K[5], cy = K[6];
  const float u = uv[2 * r], v = uv[2 * r + 1];
  // lift(), z = 1 (rend_util.py:152-153), evaluated left to right like the reference expression
  const float xl = ((((u - cx) + (cy * sk) / fy) - (sk * v) / fy) / fx) * 1.0f;
  const float yl = ((v - cy) / fy) * 1.0f;
  const float zl = 1.0f;
  float w[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float rot = (pose[4 * i] * xl + pose[4 * i + 1] * yl) + pose[4 * i + 2] * zl;
    w[i] = (rot + pose[4 * i + 3]) - pose[4 * i + 3];     // world point minus camera centre (rend_util.py:92)
  }
  const float n = __builtin_fmaxf(__builtin_sqrtf((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]), 1e-12f);
  dirs[3 * r] = w[0] / n; dirs[3 * r + 1] = w[1] / n; dirs[3 * r + 2] = w[2] / n;
  const float nc = __builtin_fmaxf(__builtin_sqrtf((xl * xl + yl * yl) + zl * zl), 1e-12f);
  depth_scale[r] = zl / nc;                                // network.py:216-217
}

// ---- a8: compositing ------------------------------------------------------------------------------------
constexpr int kMaxS = 256;

__device__ __forceinline__ float wave_cumsum_excl_out(const float* in, float* out, int m, int lane) {
  // canonical inclusive cumsum (see svs_sampler.hip); duplicated here so this file stands alone
  const int c = (m + 63) >> 6;
  const int lo = lane * c;
  const int hi = (lo + c < m) ? lo + c : m;
  double tot = 0.0;
  for (int j = lo; j < hi; ++j) tot = tot + (double)in[j];
  double scan = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(scan, d);
    if (lane >= d) scan = scan + o;
  }
  double off = __shfl_up(scan, 1);
  if (lane == 0) off = 0.0;
  double acc = 0.0;
  float last = 0.0f;
  for (int j = lo; j < hi; ++j) {
    acc = acc + (double)in[j];
    last = (float)(off + acc);
    out[j] = last;
  }
  return __shfl(last, (m - 1) / c);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

struct CompositeArgs {
  int R, S;
  const float* z;            // (R,S)
  const float* sdf;          // (R*S)
  const float* rgb;          // (R*S,3)
  const float* normals;      // (R*S,3) or nullptr
  const float* depth_scale;  // (R)
  const float* beta_param;   // device scalar: density.beta; beta = |beta| + beta_min (density.py:28-30)
  float beta_min;
  float* weights;            // (R,S)
  float* rgb_values;         // (R,3)
  float* depth_values;       // (R)
  float* depth_vals;         // (R,S) = z * depth_scale
  float* normal_map;         // (R,3) or nullptr
};

__global__ __launch_bounds__(64) void composite_kernel(CompositeArgs a) {
  __shared__ float zs[kMaxS], fe[kMaxS], sfe[kMaxS];
  const int r = blockIdx.x, lane = threadIdx.x, S = a.S;
  const float beta = __builtin_fabsf(*a.beta_param) + a.beta_min;
  for (int i = lane; i < S; i += 64) zs[i] = a.z[(size_t)r * S + i];
  __syncthreads();
  for (int i = lane; i < S; i += 64) {
    const float dist = i < S - 1 ? zs[i + 1] - zs[i] : 1e10f;
    fe[i] = dist * laplace_density(a.sdf[(size_t)r * S + i], beta);
  }
  __syncthreads();
  for (int i = lane; i < S; i += 64) sfe[i] = i == 0 ? 0.0f : fe[i - 1];
  __syncthreads();
  wave_cumsum_excl_out(sfe, sfe, S, lane);
  __syncthreads();
  float sw = 0.0f, swz = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
  const float ds = a.depth_scale[r];
  for (int i = lane; i < S; i += 64) {
    const size_t p = (size_t)r * S + i;
    const float w = (1.0f - det_exp(-fe[i])) * det_exp(-sfe[i]);
    a.weights[p] = w;
    a.depth_vals[p] = zs[i] * ds;
    sw += w; swz += w * zs[i];
    c0 += w * a.rgb[3 * p]; c1 += w * a.rgb[3 * p + 1]; c2 += w * a.rgb[3 * p + 2];
    if (a.normal_map) {
      const float g0 = a.normals[3 * p], g1 = a.normals[3 * p + 1], g2 = a.normals[3 * p + 2];
      const float nn = __builtin_sqrtf((g0 * g0 + g1 * g1) + g2 * g2);
      n0 += w * (g0 / nn); n1 += w * (g1 / nn); n2 += w * (g2 / nn);
    }
  }
  sw = wave_sum(sw); swz = wave_sum(swz); c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
  if (a.normal_map) { n0 = wave_sum(n0); n1 = wave_sum(n1); n2 = wave_sum(n2); }
  if (lane == 0) {
    a.rgb_values[3 * r] = c0; a.rgb_values[3 * r + 1] = c1; a.rgb_values[3 * r + 2] = c2;
    a.depth_values[r] = ds * (swz / (sw + 1e-8f));
    if (a.normal_map) { a.normal_map[3 * r] = n0; a.normal_map[3 * r + 1] = n1; a.normal_map[3 * r + 2] = n2; }
  }
}

}  // namespace render
}  // namespace svs

using namespace svs;
using namespace svs::render;

extern "C" {

int svs_rays_from_uv(const float* uv, const float* pose, const float* intrinsics, int n_rays, float* ray_dirs,
                     float* cam_loc, float* depth_scale, void* hip_stream) {
  if (!uv || !pose || !intrinsics || !ray_dirs || !cam_loc || !depth_scale || n_rays <= 0) {
    set_error("svs_rays_from_uv: null/invalid argument"); return SVS_EINVAL;
  }
  rays_kernel<<<(n_rays + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(uv, pose, intrinsics, n_rays, ray_dirs, cam_loc,
                                                                        depth_scale);
  return check_launch("svs_rays_from_uv");
}

int svs_composite(int n_rays, int n_samples, const float* z, const float* sdf, const float* rgb, const float* normals,
                  const float* depth_scale, const float* beta_param, float beta_min, float* weights, float* rgb_values,
                  float* depth_values, float* depth_vals, float* normal_map, void* hip_stream) {
  if (!z || !sdf || !rgb || !depth_scale || !beta_param || !weights || !rgb_values || !depth_values || !depth_vals ||
      n_rays <= 0) {
    set_error("svs_composite: null/invalid argument"); return SVS_EINVAL;
  }
  if (n_samples < 2 || n_samples > kMaxS) { set_error("svs_composite: n_samples must be in [2,%d]", kMaxS); return SVS_ESHAPE; }
  if (normal_map && !normals) { set_error("svs_composite: normal_map needs normals"); return SVS_EINVAL; }
  CompositeArgs a{n_rays, n_samples, z, sdf, rgb, normals, depth_scale, beta_param, beta_min, weights, rgb_values,
                  depth_values, depth_vals, normal_map};
  composite_kernel<<<n_rays, 64, 0, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_composite");
}

}  // extern "C"
