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
  const float n = __builtin_fmaxf(norm3(w[0], w[1], w[2]), 1e-12f);          // F.normalize: x / max(x.norm(2), eps)
  dirs[3 * r] = w[0] / n; dirs[3 * r + 1] = w[1] / n; dirs[3 * r + 2] = w[2] / n;
  const float nc = __builtin_fmaxf(norm3(xl, yl, zl), 1e-12f);
  depth_scale[r] = zl / nc;                                // network.py:216-217
}

// ---- a9: the eikonal points of a train-mode forward (network.py:258-266) -----------------------------------------------
// out[0..R) = the uniform draws; out[R + r] = cam + z_eik[r] * dirs[r] (one product, one sum per component, as the
// reference's broadcast expression evaluates it)
__global__ void eikonal_points_kernel(const float* __restrict__ uniform, const float* __restrict__ cam,
                                      const float* __restrict__ z_eik, const float* __restrict__ dirs, int R,
                                      float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * R) return;
  out[i] = uniform[i];
  const int r = i / 3, c = i - 3 * r;
  out[3 * R + i] = cam[c] + z_eik[r] * dirs[i];
}

// A step's small host inputs (the train-mode draws, pixels, camera: 0.2-0.8 MB) read by a kernel straight from the
// pinned staging buffer (hipHostMalloc memory is mapped into the device's address space).  A hipMemcpyAsync does the same
// transfer, but the first kernel behind it starts ~30 us after it ends (cache maintenance and a barrier around the blit);
// behind this kernel the next one starts at once.
__global__ void stage_in_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long n4, const float* src_tail,
                                float* dst_tail, int n_tail) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    dst[i] = __builtin_nontemporal_load(src + i);
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

// BG model: the sampler's depths (R, n) -> the n - 1 foreground depths, dense, and the last one (the sphere exit,
// network_bg.py:60-62: z_max = z_vals[:, -1]; z_vals = z_vals[:, :-1])
__global__ void split_last_kernel(const float* __restrict__ z, int R, int n, float* __restrict__ head, float* __restrict__ last) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * n) return;
  const int r = i / n, c = i - r * n;
  if (c == n - 1) last[r] = z[i];
  else head[r * (n - 1) + c] = z[i];
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

// ---- float64 helpers of the compositing BACKWARD kernels.  The forward kernels follow the bit-exact float32 contract of
// the sampler; their reverse passes have no such contract (the reference gets these gradients from float32 autograd) and
// d loss / d beta is a sum with 1 / beta^3 factors and heavy cancellation: in float32 it was 4e-4 off float64 autograd at
// beta = 0.005.  Everything between the float32 inputs and the float32 outputs of the reverse passes is therefore double.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ double dexp(double x) { return x < -745.0 ? 0.0 : exp(x); }
// out[j] = sum_{i <= j} in[i] (in place allowed); returns the total
__device__ __forceinline__ double wave_cumsum_incl(const double* in, double* out, int m, int lane) {
  const int c = (m + 63) >> 6;
  const int lo = lane * c;
  const int hi = (lo + c < m) ? lo + c : m;
  double tot = 0.0;
  for (int j = lo; j < hi; ++j) tot += in[j];
  double scan = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(scan, d);
    if (lane >= d) scan += o;
  }
  double acc = scan - tot;
  for (int j = lo; j < hi; ++j) { acc += in[j]; out[j] = acc; }
  // the total is returned as the very value stored in out[m - 1]: callers form suffix sums as total - out[i], which must be
  // exactly zero for the last element (its free energy carries dist = 1e10)
  return __shfl(acc, (m - 1) / c);
}
// Laplace density and its derivatives in double (density.py:21-30)
struct DensityD {
  double sigma, dsig_dsdf, dsig_dbeta;
  __device__ __forceinline__ DensityD(double s, double beta) {
    const double as = s < 0.0 ? -s : s;
    const double e = dexp(-as / beta);
    const double sgn = s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : 0.0);
    sigma = (1.0 / beta) * (0.5 + 0.5 * sgn * (e - 1.0));
    // sign(0) = 0 in the reference's density: its gradient w.r.t. sdf vanishes there as well
    dsig_dsdf = s != 0.0 ? -0.5 * e / (beta * beta) : 0.0;
    dsig_dbeta = -sigma / beta + 0.5 * sgn * e * as / (beta * beta * beta);
  }
};

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
    const float w = (1.0f - sleef_expf(-fe[i])) * sleef_expf(-sfe[i]);
    a.weights[p] = w;
    a.depth_vals[p] = zs[i] * ds;
    sw += w; swz += w * zs[i];
    c0 += w * a.rgb[3 * p]; c1 += w * a.rgb[3 * p + 1]; c2 += w * a.rgb[3 * p + 2];
    if (a.normal_map) {
      const float g0 = a.normals[3 * p], g1 = a.normals[3 * p + 1], g2 = a.normals[3 * p + 2];
      const float nn = norm3(g0, g1, g2);            // gradients.norm(2, -1)
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

// ---- a8 backward: d(loss)/d(sdf, rgb, beta) from d(loss)/d(rgb_values, weights, depth_values) -----------------
// Forward (network.py:281-295): fe_i = dist_i * sigma(sdf_i, beta), w_i = (1 - exp(-fe_i)) * exp(-sum_{j<i} fe_j);
// rgb_values = sum w_i c_i, depth_values = ds * sum(w_i z_i) / (sum w_i + 1e-8).  Hand-derived reverse pass:
//   dL/dfe_k = dw_k * T_k * exp(-fe_k) - sum_{i>k} dw_i * w_i
//   dsigma/dsdf = -exp(-|s|/beta) / (2 beta^2),  dsigma/dbeta = -sigma/beta + sgn(s) exp(-|s|/beta) |s| / (2 beta^3)
struct CompositeBwdArgs {
  int R, S;
  const float* z; const float* sdf; const float* rgb; const float* depth_scale;
  const float* beta_param; float beta_min;
  const float* d_rgb_values;   // (R,3)
  const float* d_weights;      // (R,S) or nullptr
  const float* d_depth_values; // (R) or nullptr
  float* d_sdf;                // (R*S)
  float* d_rgb;                // (R*S,3)
  float* d_beta_ray;           // (R): per-ray d loss / d beta (summed by beta_reduce_kernel)
};

__global__ __launch_bounds__(64) void composite_bwd_kernel(CompositeBwdArgs a) {
  __shared__ double zs[kMaxS], fe[kMaxS], tr[kMaxS], dw[kMaxS], pre[kMaxS];
  const int r = blockIdx.x, lane = threadIdx.x, S = a.S;
  const double beta = (double)(__builtin_fabsf(*a.beta_param) + a.beta_min);
  for (int i = lane; i < S; i += 64) zs[i] = (double)a.z[(size_t)r * S + i];
  __syncthreads();
  for (int i = lane; i < S; i += 64) {
    const double dist = i < S - 1 ? (double)(float)(zs[i + 1] - zs[i]) : 1e10;      // (the forward's float32 distance)
    fe[i] = dist * DensityD((double)a.sdf[(size_t)r * S + i], beta).sigma;
  }
  __syncthreads();
  // tr[i] = sum_{j < i} fe_j as the inclusive scan of the shifted sequence: the last free energy (dist = 1e10) must not
  // enter any partial sum, it would cost the prefixes of its lane's chunk 1e-5 of absolute accuracy
  for (int i = lane; i < S; i += 64) tr[i] = i == 0 ? 0.0 : fe[i - 1];
  __syncthreads();
  wave_cumsum_incl(tr, tr, S, lane);
  __syncthreads();
  double sw = 0.0, swz = 0.0;
  for (int i = lane; i < S; i += 64) {
    const double T = dexp(-tr[i]);
    tr[i] = T;
    const double w = (1.0 - dexp(-fe[i])) * T;
    pre[i] = w;                         // weights, reused below
    sw += w; swz += w * zs[i];
  }
  sw = wave_sum(sw); swz = wave_sum(swz);
  __syncthreads();
  const double ds = (double)a.depth_scale[r];
  const double g0 = a.d_rgb_values[3 * r], g1 = a.d_rgb_values[3 * r + 1], g2 = a.d_rgb_values[3 * r + 2];
  const double gd = a.d_depth_values ? (double)a.d_depth_values[r] * ds : 0.0;
  const double den = sw + 1e-8;
  for (int i = lane; i < S; i += 64) {
    const size_t p = (size_t)r * S + i;
    const double w = pre[i];
    const double c0 = a.rgb[3 * p], c1 = a.rgb[3 * p + 1], c2 = a.rgb[3 * p + 2];
    a.d_rgb[3 * p] = (float)(w * g0); a.d_rgb[3 * p + 1] = (float)(w * g1); a.d_rgb[3 * p + 2] = (float)(w * g2);
    double d = (c0 * g0 + c1 * g1) + c2 * g2;
    if (a.d_weights) d += (double)a.d_weights[p];
    d += gd * (zs[i] * den - swz) / (den * den);
    dw[i] = d;
    pre[i] = d * w;                     // summand of the suffix sums
  }
  __syncthreads();
  const double tot = wave_cumsum_incl(pre, pre, S, lane);   // inclusive prefix of dw_i * w_i
  __syncthreads();
  double dbeta = 0.0;
  for (int i = lane; i < S; i += 64) {
    const size_t p = (size_t)r * S + i;
    const double suffix = tot - pre[i];                          // sum_{j>i} dw_j w_j
    const double dfe = dw[i] * tr[i] * dexp(-fe[i]) - suffix;
    const double dist = i < S - 1 ? (double)(float)(zs[i + 1] - zs[i]) : 1e10;
    const double dsig = dfe * dist;
    const DensityD dn((double)a.sdf[p], beta);
    a.d_sdf[p] = (float)(dsig * dn.dsig_dsdf);
    // exp(-fe) underflows long before dist = 1e10 matters; guard the 0 * inf of the last sample
    if (dfe != 0.0) dbeta += dsig * dn.dsig_dbeta;
  }
  dbeta = wave_sum(dbeta);
  if (lane == 0) a.d_beta_ray[r] = (float)dbeta;
}

// d loss / d density.beta = sign(beta_param) * sum over rays (float64, fixed order -> reproducible)
__global__ __launch_bounds__(256) void beta_reduce_kernel(const float* __restrict__ d_beta_ray, int R,
                                                          const float* __restrict__ beta_param, float* __restrict__ out,
                                                          int accumulate) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < R; i += 256) acc += (double)d_beta_ray[i];
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    const float bp = *beta_param;
    const float sg = bp > 0.0f ? 1.0f : (bp < 0.0f ? -1.0f : 0.0f);
    const float v = sg * (float)t;
    out[0] = accumulate ? out[0] + v : v;
  }
}

// ---- a10: MVS prior lookup (VolOpt.cost_mapping, volsdf/vsdf.py:382-452) --------------------------------
struct LookupView {
  float fx, fy, cx, cy, sk;
  float c2w[12];            // rows of the 3x4 camera-to-world matrix
  const float* cost;        // probability volume (D,H,W)
  const float* z_near;      // depth hypotheses[0]   (H,W)
  const float* z_far;       // depth hypotheses[-1]  (H,W)
  int D, H, W;
};
constexpr int kMaxViews = 4;
struct LookupArgs {
  const float* xyz;         // (P,3) explicit world points, or nullptr -> cam + z*dir
  const float* cam; const float* dirs; const float* z; int S;
  int P, n_views, same_view, inverse_depth;
  const int* same_view_dev; // optional: the rendered view's index read from the device (captured launch sequences)
  float half_w, half_h;     // (W_img-1)/2, (H_img-1)/2 of the SceneDataset resolution (vsdf.py:397,414-415)
  LookupView v[kMaxViews];
  float* pj; float* pi; unsigned char* valid;
};

// F.grid_sample(bilinear, zeros, align_corners=True), one channel
__device__ __forceinline__ float sample2d(const float* __restrict__ img, int H, int W, float gx, float gy) {
  const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
  const float fx0 = __builtin_floorf(ix), fy0 = __builtin_floorf(iy);
  const float tx = ix - fx0, ty = iy - fy0;
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float xx = fx0 + (float)(c & 1), yy = fy0 + (float)(c >> 1);
    if (xx >= 0.0f && xx <= (float)(W - 1) && yy >= 0.0f && yy <= (float)(H - 1)) {
      const float w = ((c & 1) ? tx : 1.0f - tx) * ((c >> 1) ? ty : 1.0f - ty);
      acc += w * img[(int)yy * W + (int)xx];
    }
  }
  return acc;
}
__device__ __forceinline__ float sample3d(const float* __restrict__ vol, int D, int H, int W, float gx, float gy, float gz) {
  const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1),
              iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
  const float fx0 = __builtin_floorf(ix), fy0 = __builtin_floorf(iy), fz0 = __builtin_floorf(iz);
  const float tx = ix - fx0, ty = iy - fy0, tz = iz - fz0;
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float xx = fx0 + (float)(c & 1), yy = fy0 + (float)((c >> 1) & 1), zz = fz0 + (float)(c >> 2);
    if (xx >= 0.0f && xx <= (float)(W - 1) && yy >= 0.0f && yy <= (float)(H - 1) && zz >= 0.0f && zz <= (float)(D - 1)) {
      const float w = (((c & 1) ? tx : 1.0f - tx) * (((c >> 1) & 1) ? ty : 1.0f - ty)) * ((c >> 2) ? tz : 1.0f - tz);
      acc += w * vol[((size_t)(int)zz * H + (int)yy) * W + (int)xx];
    }
  }
  return acc;
}

// One wave per (64 samples, view): the views of a sample are independent chains of dependent gathers (depth range: 2 x 4
// texels, then the 8 texels of the probability volume), so they run as the waves of one workgroup instead of one after the
// other in a thread (25 -> 12 us for 256 rays x 98 samples x 3 views); wave 0 adds them up in view order, as the reference's
// loop does (vsdf.py:399-449).
__global__ __launch_bounds__(64 * kMaxViews) void cost_lookup_kernel(LookupArgs a) {
  __shared__ float cs[kMaxViews][64];
  __shared__ unsigned char ok[kMaxViews][64];
  const int lane = threadIdx.x & 63;
  const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // this wave's view
  const int p = blockIdx.x * 64 + lane;
  if (p < a.P) {
    float X, Y, Z;
    if (a.xyz) { X = a.xyz[3 * p]; Y = a.xyz[3 * p + 1]; Z = a.xyz[3 * p + 2]; }
    else {
      const int r = p / a.S;
      const float zz = a.z[p];
      X = a.cam[0] + zz * a.dirs[3 * r]; Y = a.cam[1] + zz * a.dirs[3 * r + 1]; Z = a.cam[2] + zz * a.dirs[3 * r + 2];
    }
    const LookupView& v = a.v[j];
    // xyz_j = (xyz - t) @ R   (row vector times the c2w rotation = world -> camera), vsdf.py:402-403
    const float dx = X - v.c2w[3], dy = Y - v.c2w[7], dz = Z - v.c2w[11];
    const float px = (dx * v.c2w[0] + dy * v.c2w[4]) + dz * v.c2w[8];
    const float py = (dx * v.c2w[1] + dy * v.c2w[5]) + dz * v.c2w[9];
    float pz = (dx * v.c2w[2] + dy * v.c2w[6]) + dz * v.c2w[10];
    float y = (py / pz) * v.fy + v.cy;
    float x = ((px / pz) * v.fx + v.cx) + ((y - v.cy) * v.sk) / v.fy;
    x = x / a.half_w - 1.0f;
    y = y / a.half_h - 1.0f;
    bool inval = (pz < 1e-5f) || (x > 1.001f) || (x < -1.001f) || (y > 1.001f) || (y < -1.001f);
    if (inval) { x = -99.0f; y = -99.0f; pz = -99.0f; }
    const float nearv = sample2d(v.z_near, v.H, v.W, x, y);
    float farv = sample2d(v.z_far, v.H, v.W, x, y);
    float zn;
    if (a.inverse_depth) {
      if (inval) farv = 1e-8f;
      zn = (2.0f * (1.0f - nearv / pz)) / (1.0f - nearv / farv) - 1.0f;
    } else {
      zn = (2.0f * (pz - nearv)) / (farv - nearv) - 1.0f;
    }
    inval = (nearv < 1e-5f) || (farv < 1e-5f) || (zn > 1.01f) || (zn < -1.01f) || inval;
    if (inval) { x = -99.0f; y = -99.0f; zn = -99.0f; }
    cs[j][lane] = sample3d(v.cost, v.D, v.H, v.W, x, y, zn);
    ok[j][lane] = inval ? 0 : 1;
  }
  __syncthreads();
  if (j != 0 || p >= a.P) return;
  const int same_view = a.same_view_dev ? *a.same_view_dev : a.same_view;
  float pj = 0.0f, pi = 0.0f;
  bool valid = false;
  for (int k = 0; k < a.n_views; ++k) {
    const float c = cs[k][lane];
    if (k == same_view) pi = c;
    else { pj += c; valid = valid || ok[k][lane]; }
  }
  a.pj[p] = pj;
  a.pi[p] = valid ? pi : 0.0f;
  a.valid[p] = valid ? 1 : 0;
}

// ---- a11: VolSDFLoss.forward (volsdf/model/loss.py:80-114) + its gradient w.r.t. the model outputs ---------
struct LossArgs {
  int R, S, n_eik;
  int R_norm, n_eik_norm;    // denominators of the means (>= R, n_eik when the batch is processed in several groups)
  const float* rgb_values;   // (R,3)
  const float* rgb_gt;       // (R,3) the target of the rgb term (rgb, or rgb_smooth in the annealed phase)
  const float* grad_theta;   // (n_eik,3) or nullptr
  const float* weights;      // (R,S)
  const float* pi; const float* pj;   // (R,S) or nullptr
  const float* depth_values; // (R)
  float rgb_weight, eikonal_weight, mvs_weight, sparse_weight;
  float gce, confi, anneal_sparse;    // anneal_sparse > 0 <=> annealed phase (masked rgb + sparse term)
  int annealed;
  const float* anneal_dev;   // optional {annealed (0/1), anneal_sparse} on the device (captured launch sequences)
  float* losses;             // out[5]: rgb, eikonal, mvs, sparse, total
  float* d_rgb_values;       // (R,3)   d total / d rgb_values
  float* d_grad_theta;       // (n_eik,3)
  float* d_weights;          // (R,S)
  float* d_depth_values;     // (R)
};

// one wave per ray (coalesced over the samples); per-ray partial sums go to `partial` and a single-block second
// kernel adds them in a fixed order (reproducible)
__global__ __launch_bounds__(256) void loss_rays_kernel(LossArgs a, double* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);      // rays first, then groups of 64 eikonal points
  const bool has_mvs = a.pi != nullptr;
  if (unit < a.R) {
    const int r = unit;
    double conf = 0.0, lm = 0.0;
    float dls[4];                                              // S <= 256: up to 4 samples per lane
    if (has_mvs) {
      int k = 0;
      for (int s = lane; s < a.S; s += 64, ++k) {
        const size_t p = (size_t)r * a.S + s;
        const float pw = a.pi[p] * a.pj[p];
        const float w = a.weights[p];
        conf += (double)pw;
        float l, dl;
        if (a.gce == 1.0f) { l = -pw * w; dl = -pw; }
        else if (a.gce == 0.0f) { l = -pw * __logf(w + 1e-8f); dl = -pw / (w + 1e-8f); }
        else { const float wq = __powf(w, a.gce); l = (-pw * wq) * __logf(w + 1e-8f); dl = (-pw * wq) / (w + 1e-8f); }
        lm += (double)l;
        dls[k & 3] = dl;
      }
      for (int d = 32; d >= 1; d >>= 1) { conf += __shfl_xor(conf, d); lm += __shfl_xor(lm, d); }
    }
    const bool mvs_on = has_mvs && a.mvs_weight > 0.0f && conf > (double)a.confi;
    {
      const float sc = mvs_on ? a.mvs_weight / (float)a.R_norm : 0.0f;
      int k = 0;
      for (int s = lane; s < a.S; s += 64, ++k) a.d_weights[(size_t)r * a.S + s] = has_mvs ? dls[k & 3] * sc : 0.0f;
    }
    if (lane == 0) {
      const bool masked = (a.anneal_dev ? a.anneal_dev[0] != 0.0f : a.annealed != 0) && has_mvs;
      const float anneal_sparse = a.anneal_dev ? a.anneal_dev[1] : a.anneal_sparse;
      const bool rgb_on = !masked || conf < 1e-8;
      double l1 = 0.0;
      for (int c = 0; c < 3; ++c) {
        const float d = a.rgb_values[3 * r + c] - a.rgb_gt[3 * r + c];
        l1 += (double)__builtin_fabsf(d);
        // torch.sign: sign(NaN) = NaN -- a non-finite colour poisons the step's gradient, which the NaN guard then drops
        // as a whole (volsdf/vsdf.py:454-463), exactly what happens in the reference (e.g. the ray through the centre of
        // the bounding sphere in the inverted-sphere background, network_bg.py:196-197: rot_axis = 0 / 0)
        const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : d);
        a.d_rgb_values[3 * r + c] = rgb_on ? a.rgb_weight * sg / (3.0f * (float)a.R_norm) : 0.0f;
      }
      float dd = 0.0f;
      double sp = 0.0;
      if (masked && conf < (double)a.confi) {
        const float dep = a.depth_values[r] + 1e-3f;
        sp = 1.0 / (double)dep;
        dd = -(a.sparse_weight * anneal_sparse) / (dep * dep) / (float)a.R_norm;
      }
      a.d_depth_values[r] = dd;
      double* o = partial + (size_t)unit * 4;
      o[0] = rgb_on ? l1 / 3.0 : 0.0; o[1] = 0.0; o[2] = mvs_on ? lm : 0.0; o[3] = sp;
    }
  } else {
    const int i = (unit - a.R) * 64 + lane;
    double e = 0.0;
    if (i < a.n_eik) {
      const float g0 = a.grad_theta[3 * i], g1 = a.grad_theta[3 * i + 1], g2 = a.grad_theta[3 * i + 2];
      const float n = norm3(g0, g1, g2);             // grad_theta.norm(2, dim=1), loss.py:50
      e = (double)(n - 1.0f) * (double)(n - 1.0f);
      const float k = n > 0.0f ? a.eikonal_weight * 2.0f * (n - 1.0f) / (n * (float)a.n_eik_norm) : (n != n ? n : 0.0f);
      a.d_grad_theta[3 * i] = k * g0; a.d_grad_theta[3 * i + 1] = k * g1; a.d_grad_theta[3 * i + 2] = k * g2;
    }
    for (int d = 32; d >= 1; d >>= 1) e += __shfl_xor(e, d);
    if (lane == 0) { double* o = partial + (size_t)unit * 4; o[0] = 0.0; o[1] = e; o[2] = 0.0; o[3] = 0.0; }
  }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(LossArgs a, const double* __restrict__ partial, int n_units) {
  __shared__ double sh[4][4];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int u = threadIdx.x; u < n_units; u += 256)
    for (int k = 0; k < 4; ++k) acc[k] += partial[(size_t)u * 4 + k];
  for (int k = 0; k < 4; ++k) {
    for (int d = 32; d >= 1; d >>= 1) acc[k] += __shfl_xor(acc[k], d);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[4];
    for (int k = 0; k < 4; ++k) t[k] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
    const float rgb = (float)(t[0] / a.R_norm), eik = a.n_eik_norm ? (float)(t[1] / a.n_eik_norm) : 0.0f;
    const float mvs = (float)(t[2] / a.R_norm), sp = (float)(t[3] / a.R_norm);
    a.losses[0] = rgb; a.losses[1] = eik; a.losses[2] = mvs; a.losses[3] = sp;
    const float anneal_sparse = a.anneal_dev ? a.anneal_dev[1] : a.anneal_sparse;
    a.losses[4] = a.rgb_weight * rgb + a.eikonal_weight * eik + a.mvs_weight * mvs + (a.sparse_weight * anneal_sparse) * sp;
  }
}

// ---- a9 (BG model): inverted-sphere samples and fg/bg compositing (volsdf/model/network_bg.py) --------------------
// torch.linspace(start,end,n)[i] in float32 (see svs_sampler.hip)
__device__ __forceinline__ float linspace_at_r(float start, float end, int n, int i) {
  const float step = (end - start) / (float)(n - 1);
  return i < n / 2 ? __builtin_fmaf(step, (float)i, start) : __builtin_fmaf(-step, (float)(n - 1 - i), end);
}

struct BgPointsArgs {
  int R, N;                 // rays, inverse-sphere samples per ray (32)
  const float* cam; int cam_stride;
  const float* dirs;        // (R,3)
  const float* jitter;      // (R,N) uniform draws in train mode or nullptr
  float radius;             // scene_bounding_sphere
  float* z_bg;              // (R,N) inverse depths, descending (1/r -> 0), as flipped at network_bg.py:82
  float* pts;               // (R*N,4)
  float* depth_real;        // (R,N)
};

// UniformSampler(1, 0, N, far=1) (ray_sampler.py:22-43 via :215-216) + depth2pts_outside (network_bg.py:182-214).
// One thread per (ray, sample).
__global__ void bg_points_kernel(BgPointsArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.R * a.N) return;
  const int r = idx / a.N, k = idx % a.N;
  const int i = a.N - 1 - k;                              // position before the flip
  auto zu = [&](int j) { const float t = linspace_at_r(0.0f, 1.0f, a.N, j); return 0.0f * (1.0f - t) + 1.0f * t; };
  float z = zu(i);
  if (a.jitter) {
    const float upper = i < a.N - 1 ? 0.5f * (zu(i + 1) + zu(i)) : zu(a.N - 1);
    const float lower = i > 0 ? 0.5f * (zu(i) + zu(i - 1)) : zu(0);
    z = lower + (upper - lower) * a.jitter[(size_t)r * a.N + i];
  }
  const float depth = z * (float)(1.0 / (double)a.radius);
  a.z_bg[idx] = depth;
  const float* o = a.cam + (size_t)r * a.cam_stride;
  const float* d = a.dirs + 3 * (size_t)r;
  const float o_dot_d = (d[0] * o[0] + d[1] * o[1]) + d[2] * o[2];
  const float under = o_dot_d * o_dot_d - (((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]) - a.radius * a.radius);
  const float d_sphere = __builtin_sqrtf(under) - o_dot_d;
  float ps[3], pm[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { ps[c] = o[c] + d_sphere * d[c]; pm[c] = o[c] - o_dot_d * d[c]; }
  const float pm_norm = norm3(pm[0], pm[1], pm[2]);
  float ax[3] = {o[1] * ps[2] - o[2] * ps[1], o[2] * ps[0] - o[0] * ps[2], o[0] * ps[1] - o[1] * ps[0]};
  const float axn = norm3(ax[0], ax[1], ax[2]);
#pragma unroll
  for (int c = 0; c < 3; ++c) ax[c] = ax[c] / axn;
  const float phi = asinf(pm_norm / a.radius);
  const float theta = asinf(pm_norm * depth);
  const float ang = phi - theta;
  const float ca = cosf(ang), sa = sinf(ang);
  const float cr[3] = {ax[1] * ps[2] - ax[2] * ps[1], ax[2] * ps[0] - ax[0] * ps[2], ax[0] * ps[1] - ax[1] * ps[0]};
  const float adp = (ax[0] * ps[0] + ax[1] * ps[1]) + ax[2] * ps[2];
  float pn[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) pn[c] = (ps[c] * ca + cr[c] * sa) + (ax[c] * adp) * (1.0f - ca);
  const float pnn = norm3(pn[0], pn[1], pn[2]);
  a.pts[4 * (size_t)idx] = pn[0] / pnn; a.pts[4 * (size_t)idx + 1] = pn[1] / pnn; a.pts[4 * (size_t)idx + 2] = pn[2] / pnn;
  a.pts[4 * (size_t)idx + 3] = depth;
  const float dd = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
  const float d1 = -o_dot_d / dd;
  const float ray_d_cos = 1.0f / norm3(d[0], d[1], d[2]);
  a.depth_real[idx] = ((1.0f / (depth + 1e-6f)) * cosf(theta)) * ray_d_cos + d1;
}

struct CompositeBgArgs {
  int R, S, Nb;              // rays, fg samples (97), bg samples (32)
  const float* z;            // (R,S) fg sample distances
  const float* z_max;        // (R) sphere exit
  const float* sdf;          // (R*S)
  const float* rgb;          // (R*S,3)
  const float* normals;      // (R*S,3) or nullptr
  const float* depth_scale;  // (R)
  const float* beta_param; float beta_min;
  const float* z_bg;         // (R,Nb) descending inverse depths
  const float* bg_out0;      // (R*Nb) raw bg output[:,0]; density = |.|
  const float* bg_rgb;       // (R*Nb,3)
  const float* bg_depth;     // (R,Nb) conventional depths of the bg samples
  float* weights;            // (R,S)
  float* bg_trans;           // (R) transmittance behind the last fg sample
  float* bg_weights;         // (R,Nb)
  float* rgb_values;         // (R,3)
  float* depth_values;       // (R)
  float* depth_values_all;   // (R)
  float* depth_vals;         // (R,S)
  float* normal_map;         // (R,3) or nullptr
};

// VolSDFNetworkBG.volume_rendering / bg_volume_rendering and the composition (network_bg.py:76-125, 147-180)
__global__ __launch_bounds__(64) void composite_bg_kernel(CompositeBgArgs a) {
  __shared__ float zs[kMaxS], fe[kMaxS], sfe[kMaxS], bfe[64], bsf[64];
  const int r = blockIdx.x, lane = threadIdx.x, S = a.S, Nb = a.Nb;
  const float beta = __builtin_fabsf(*a.beta_param) + a.beta_min;
  for (int i = lane; i < S; i += 64) zs[i] = a.z[(size_t)r * S + i];
  for (int i = lane; i < Nb; i += 64) bsf[i] = a.z_bg[(size_t)r * Nb + i];
  __syncthreads();
  for (int i = lane; i < S; i += 64) {
    const float dist = i < S - 1 ? zs[i + 1] - zs[i] : a.z_max[r] - zs[i];
    fe[i] = dist * laplace_density(a.sdf[(size_t)r * S + i], beta);
  }
  for (int i = lane; i < Nb; i += 64) {
    const float dist = i < Nb - 1 ? bsf[i] - bsf[i + 1] : 1e10f;
    bfe[i] = dist * __builtin_fabsf(a.bg_out0[(size_t)r * Nb + i]);
  }
  __syncthreads();
  for (int i = lane; i <= S; i += 64) sfe[i] = i == 0 ? 0.0f : fe[i - 1];      // S + 1 entries: the last is the total
  for (int i = lane; i < Nb; i += 64) bsf[i] = i == 0 ? 0.0f : bfe[i - 1];
  __syncthreads();
  wave_cumsum_excl_out(sfe, sfe, S + 1, lane);
  __syncthreads();
  wave_cumsum_excl_out(bsf, bsf, Nb, lane);
  __syncthreads();
  const float tbg = sleef_expf(-sfe[S]);
  const float ds = a.depth_scale[r];
  float sw = 0.0f, swz = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, swa = 0.0f, swd = 0.0f;
  for (int i = lane; i < S; i += 64) {
    const size_t p = (size_t)r * S + i;
    const float w = (1.0f - sleef_expf(-fe[i])) * sleef_expf(-sfe[i]);
    a.weights[p] = w;
    const float dv = zs[i] * ds;
    a.depth_vals[p] = dv;
    sw += w; swz += w * dv; swa += w; swd += w * (ds * zs[i]);
    c0 += w * a.rgb[3 * p]; c1 += w * a.rgb[3 * p + 1]; c2 += w * a.rgb[3 * p + 2];
    if (a.normal_map) {
      const float g0 = a.normals[3 * p], g1 = a.normals[3 * p + 1], g2 = a.normals[3 * p + 2];
      const float nn = norm3(g0, g1, g2);            // gradients.norm(2, -1)
      n0 += w * (g0 / nn); n1 += w * (g1 / nn); n2 += w * (g2 / nn);
    }
  }
  float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
  for (int i = lane; i < Nb; i += 64) {
    const size_t p = (size_t)r * Nb + i;
    const float bw = (1.0f - sleef_expf(-bfe[i])) * sleef_expf(-bsf[i]);
    a.bg_weights[p] = bw;
    b0 += bw * a.bg_rgb[3 * p]; b1 += bw * a.bg_rgb[3 * p + 1]; b2 += bw * a.bg_rgb[3 * p + 2];
    const float wa = tbg * bw;
    swa += wa; swd += wa * (ds * a.bg_depth[p]);
  }
  sw = wave_sum(sw); swz = wave_sum(swz); c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
  b0 = wave_sum(b0); b1 = wave_sum(b1); b2 = wave_sum(b2); swa = wave_sum(swa); swd = wave_sum(swd);
  if (a.normal_map) { n0 = wave_sum(n0); n1 = wave_sum(n1); n2 = wave_sum(n2); }
  if (lane == 0) {
    a.bg_trans[r] = tbg;
    a.rgb_values[3 * r] = c0 + tbg * b0; a.rgb_values[3 * r + 1] = c1 + tbg * b1; a.rgb_values[3 * r + 2] = c2 + tbg * b2;
    a.depth_values[r] = swz / (sw + 1e-8f);
    a.depth_values_all[r] = swd / (swa + 1e-8f);
    if (a.normal_map) { a.normal_map[3 * r] = n0; a.normal_map[3 * r + 1] = n1; a.normal_map[3 * r + 2] = n2; }
  }
}

struct CompositeBgBwdArgs {
  int R, S, Nb;
  const float* z; const float* z_max; const float* sdf; const float* rgb; const float* depth_scale;
  const float* beta_param; float beta_min;
  const float* z_bg; const float* bg_out0; const float* bg_rgb;
  const float* d_rgb_values;   // (R,3)
  const float* d_weights;      // (R,S) or nullptr
  const float* d_depth_values; // (R) or nullptr (the fg depth of network_bg.py:111-112)
  const float* d_depth_all;    // (R) or nullptr: gradient of depth_values_all (network_bg.py:105-107) -- what the sparsity
                               // term of the loss reads (loss.py:72-73); needs bg_depth
  const float* bg_depth;       // (R,Nb) conventional depths of the inverse-sphere samples (with d_depth_all)
  float* d_sdf;                // (R*S)
  float* d_rgb;                // (R*S,3)
  float* d_bg_out0;            // (R*Nb)
  float* d_bg_rgb;             // (R*Nb,3)
  float* d_beta_ray;           // (R)
};

// backward of composite_bg_kernel with respect to sdf, rgb, beta, the bg density logits and the bg colours
__global__ __launch_bounds__(64) void composite_bg_bwd_kernel(CompositeBgBwdArgs a) {
  // (float64 throughout, see DensityD)
  __shared__ double zs[kMaxS], fe[kMaxS], tr[kMaxS], dw[kMaxS], pre[kMaxS], bz[64], bfe[64], btr[64], bdw[64], bpre[64];
  const int r = blockIdx.x, lane = threadIdx.x, S = a.S, Nb = a.Nb;
  const double beta = (double)(__builtin_fabsf(*a.beta_param) + a.beta_min);
  for (int i = lane; i < S; i += 64) zs[i] = (double)a.z[(size_t)r * S + i];
  for (int i = lane; i < Nb; i += 64) bz[i] = (double)a.z_bg[(size_t)r * Nb + i];
  __syncthreads();
  const double zmax = (double)a.z_max[r];
  auto fg_dist = [&](int i) { return (double)(float)((i < S - 1 ? zs[i + 1] : zmax) - zs[i]); };   // the forward's float32 distances
  auto bg_dist = [&](int i) { return i < Nb - 1 ? (double)(float)(bz[i] - bz[i + 1]) : 1e10; };
  for (int i = lane; i < S; i += 64) fe[i] = fg_dist(i) * DensityD((double)a.sdf[(size_t)r * S + i], beta).sigma;
  for (int i = lane; i < Nb; i += 64) {
    const double o = (double)a.bg_out0[(size_t)r * Nb + i];
    bfe[i] = bg_dist(i) * (o < 0.0 ? -o : o);
  }
  __syncthreads();
  // exclusive prefixes as inclusive scans of the shifted sequences (the last background free energy, dist = 1e10, must not
  // enter any partial sum); tr[S] = the total foreground free energy
  for (int i = lane; i <= S; i += 64) tr[i] = i == 0 ? 0.0 : fe[i - 1];
  for (int i = lane; i < Nb; i += 64) btr[i] = i == 0 ? 0.0 : bfe[i - 1];
  __syncthreads();
  wave_cumsum_incl(tr, tr, S + 1, lane);
  __syncthreads();
  wave_cumsum_incl(btr, btr, Nb, lane);
  __syncthreads();
  const double tbg = dexp(-tr[S]);
  const double ds = (double)a.depth_scale[r];
  const double g0 = a.d_rgb_values[3 * r], g1 = a.d_rgb_values[3 * r + 1], g2 = a.d_rgb_values[3 * r + 2];
  // background: weights, colour sums, and the sums of depth_values_all = swd / (swa + 1e-8) over [w_fg, tbg * bw] with
  // depths ds * [z, bg_depth]
  const double ga = a.d_depth_all ? (double)a.d_depth_all[r] : 0.0;
  double bdot = 0.0;       // sum_c g_c * sum_k bw_k cb_kc
  double swa = 0.0, swd = 0.0;
  for (int i = lane; i < Nb; i += 64) {
    const size_t p = (size_t)r * Nb + i;
    const double T = dexp(-btr[i]);
    btr[i] = T;
    const double bw = (1.0 - dexp(-bfe[i])) * T;
    const double c0 = a.bg_rgb[3 * p], c1 = a.bg_rgb[3 * p + 1], c2 = a.bg_rgb[3 * p + 2];
    a.d_bg_rgb[3 * p] = (float)((tbg * bw) * g0); a.d_bg_rgb[3 * p + 1] = (float)((tbg * bw) * g1);
    a.d_bg_rgb[3 * p + 2] = (float)((tbg * bw) * g2);
    const double cg = (c0 * g0 + c1 * g1) + c2 * g2;
    bdot += bw * cg;
    bdw[i] = tbg * cg;                 // d loss / d bw_i (colour term; the depth term follows once the sums are known)
    bpre[i] = bw;
    if (a.d_depth_all) { swa += tbg * bw; swd += (tbg * bw) * (ds * (double)a.bg_depth[p]); }
  }
  bdot = wave_sum(bdot);
  // foreground: weights and their sums
  double sw = 0.0, swz = 0.0;
  for (int i = lane; i < S; i += 64) {
    const double T = dexp(-tr[i]);
    tr[i] = T;
    const double w = (1.0 - dexp(-fe[i])) * T;
    pre[i] = w;
    sw += w; swz += w * (zs[i] * ds);
  }
  sw = wave_sum(sw); swz = wave_sum(swz);
  double dall = 1.0, qnum = 0.0;       // depth_values_all's denominator and numerator
  if (a.d_depth_all) {
    swa = wave_sum(swa); swd = wave_sum(swd);
    dall = (swa + sw) + 1e-8; qnum = swd + swz;
  }
  __syncthreads();
  // d depth_all / d (a weight with depth dv) = (dv * dall - qnum) / dall^2; through tbg it also reaches every fg free energy
  double bdep = 0.0;        // sum_k bw_k q_k: d depth_all / d tbg
  for (int i = lane; i < Nb; i += 64) {
    const double bw = bpre[i];
    if (a.d_depth_all) {
      const double q = ((ds * (double)a.bg_depth[(size_t)r * Nb + i]) * dall - qnum) / (dall * dall);
      bdw[i] += ga * tbg * q;
      bdep += bw * q;
    }
    bpre[i] = bdw[i] * bw;
  }
  if (a.d_depth_all) bdot += ga * wave_sum(bdep);
  __syncthreads();
  const double gd = a.d_depth_values ? (double)a.d_depth_values[r] : 0.0;
  const double den = sw + 1e-8;
  for (int i = lane; i < S; i += 64) {
    const size_t p = (size_t)r * S + i;
    const double w = pre[i];
    const double c0 = a.rgb[3 * p], c1 = a.rgb[3 * p + 1], c2 = a.rgb[3 * p + 2];
    a.d_rgb[3 * p] = (float)(w * g0); a.d_rgb[3 * p + 1] = (float)(w * g1); a.d_rgb[3 * p + 2] = (float)(w * g2);
    double d = (c0 * g0 + c1 * g1) + c2 * g2;
    if (a.d_weights) d += (double)a.d_weights[p];
    d += gd * ((zs[i] * ds) * den - swz) / (den * den);
    if (a.d_depth_all) d += ga * ((zs[i] * ds) * dall - qnum) / (dall * dall);
    dw[i] = d;
    pre[i] = d * w;
  }
  __syncthreads();
  const double tot = wave_cumsum_incl(pre, pre, S, lane);
  __syncthreads();
  const double btot = wave_cumsum_incl(bpre, bpre, Nb, lane);
  __syncthreads();
  double dbeta = 0.0;
  for (int i = lane; i < S; i += 64) {
    const size_t p = (size_t)r * S + i;
    const double suffix = tot - pre[i];
    // every free energy also attenuates the background term: d (tbg * B) / d fe_i = -tbg * B
    const double dfe = (dw[i] * tr[i] * dexp(-fe[i]) - suffix) - tbg * bdot;
    const double dsig = dfe * fg_dist(i);
    const DensityD dn((double)a.sdf[p], beta);
    a.d_sdf[p] = (float)(dsig * dn.dsig_dsdf);
    dbeta += dsig * dn.dsig_dbeta;
  }
  for (int i = lane; i < Nb; i += 64) {
    const size_t p = (size_t)r * Nb + i;
    const double suffix = btot - bpre[i];
    const double dfe = bdw[i] * btr[i] * dexp(-bfe[i]) - suffix;
    const float o = a.bg_out0[p];
    const double sg = o > 0.0f ? 1.0 : (o < 0.0f ? -1.0 : 0.0);
    // exp(-fe) underflows long before dist = 1e10 matters; guard the 0 * inf of the last sample
    a.d_bg_out0[p] = dfe != 0.0 ? (float)((dfe * bg_dist(i)) * sg) : 0.0f;
  }
  dbeta = wave_sum(dbeta);
  if (lane == 0) a.d_beta_ray[r] = (float)dbeta;
}

}  // namespace render
}  // namespace svs

using namespace svs;
using namespace svs::render;

extern "C" {

size_t svs_loss_workspace_bytes(int n_rays, int n_eik) { return (size_t)(n_rays + (n_eik + 63) / 64 + 4) * 4 * sizeof(double); }

int svs_rays_from_uv(const float* uv, const float* pose, const float* intrinsics, int n_rays, float* ray_dirs,
                     float* cam_loc, float* depth_scale, void* hip_stream) {
  if (!uv || !pose || !intrinsics || !ray_dirs || !cam_loc || !depth_scale || n_rays <= 0) {
    set_error("svs_rays_from_uv: null/invalid argument"); return SVS_EINVAL;
  }
  rays_kernel<<<(n_rays + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(uv, pose, intrinsics, n_rays, ray_dirs, cam_loc,
                                                                        depth_scale);
  return check_launch("svs_rays_from_uv");
}

int svs_stage_in(const void* pinned_host, void* device, size_t bytes, void* hip_stream) {
  if (!pinned_host || !device || bytes == 0 || (bytes & 3) || (((uintptr_t)pinned_host | (uintptr_t)device) & 15)) {
    set_error("svs_stage_in: 16-byte aligned buffers, a multiple of 4 bytes"); return SVS_EINVAL;
  }
  const long long n4 = (long long)(bytes >> 4);
  const int n_tail = (int)((bytes & 15) >> 2);
  long long blocks = (n4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  stage_in_kernel<<<(int)blocks, 256, 0, (hipStream_t)hip_stream>>>((const f32x4*)pinned_host, (f32x4*)device, n4,
                                                                    (const float*)pinned_host + 4 * n4, (float*)device + 4 * n4, n_tail);
  return check_launch("svs_stage_in");
}

int svs_split_last(const float* z, int n_rays, int n, float* head, float* last, void* hip_stream) {
  if (!z || !head || !last || n_rays <= 0 || n < 2) { set_error("svs_split_last: null/invalid argument"); return SVS_EINVAL; }
  split_last_kernel<<<(n_rays * n + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(z, n_rays, n, head, last);
  return check_launch("svs_split_last");
}

int svs_eikonal_points(const float* uniform_points, const float* cam_loc, const float* z_eik, const float* ray_dirs,
                       int n_rays, float* points, void* hip_stream) {
  if (!uniform_points || !cam_loc || !z_eik || !ray_dirs || !points || n_rays <= 0) {
    set_error("svs_eikonal_points: null/invalid argument"); return SVS_EINVAL;
  }
  eikonal_points_kernel<<<(3 * n_rays + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(uniform_points, cam_loc, z_eik,
                                                                                      ray_dirs, n_rays, points);
  return check_launch("svs_eikonal_points");
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

// BG model: inverse-sphere samples of a batch of rays (ray_sampler.py:215-216, network_bg.py:79-86).
// jitter: (n_rays, n_bg) uniform draws in train mode or NULL.  -> z_bg (n_rays,n_bg) descending, pts (n_rays*n_bg,4),
// depth_real (n_rays,n_bg).
int svs_bg_points(const float* cam, int cam_stride, const float* dirs, int n_rays, int n_bg, const float* jitter,
                  float radius, float* z_bg, float* pts, float* depth_real, void* hip_stream) {
  if (!cam || !dirs || !z_bg || !pts || !depth_real || n_rays <= 0 || n_bg < 2 || n_bg > 64 || (cam_stride != 0 && cam_stride != 3)) {
    set_error("svs_bg_points: null/invalid argument"); return SVS_EINVAL;
  }
  BgPointsArgs a{n_rays, n_bg, cam, cam_stride, dirs, jitter, radius, z_bg, pts, depth_real};
  const int n = n_rays * n_bg;
  bg_points_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_bg_points");
}

// BG model: fg weights with the sphere exit as the last interval end, bg weights, composition (network_bg.py:76-125)
int svs_composite_bg(int n_rays, int n_samples, int n_bg, const float* z, const float* z_max, const float* sdf,
                     const float* rgb, const float* normals, const float* depth_scale, const float* beta_param,
                     float beta_min, const float* z_bg, const float* bg_out0, const float* bg_rgb, const float* bg_depth,
                     float* weights, float* bg_trans, float* bg_weights, float* rgb_values, float* depth_values,
                     float* depth_values_all, float* depth_vals, float* normal_map, void* hip_stream) {
  if (!z || !z_max || !sdf || !rgb || !depth_scale || !beta_param || !z_bg || !bg_out0 || !bg_rgb || !bg_depth || !weights ||
      !bg_trans || !bg_weights || !rgb_values || !depth_values || !depth_values_all || !depth_vals || n_rays <= 0) {
    set_error("svs_composite_bg: null/invalid argument"); return SVS_EINVAL;
  }
  if (n_samples < 2 || n_samples + 1 > kMaxS || n_bg < 2 || n_bg > 64) { set_error("svs_composite_bg: sample counts out of range"); return SVS_ESHAPE; }
  if (normal_map && !normals) { set_error("svs_composite_bg: normal_map needs normals"); return SVS_EINVAL; }
  CompositeBgArgs a{n_rays, n_samples, n_bg, z, z_max, sdf, rgb, normals, depth_scale, beta_param, beta_min, z_bg, bg_out0,
                    bg_rgb, bg_depth, weights, bg_trans, bg_weights, rgb_values, depth_values, depth_values_all, depth_vals,
                    normal_map};
  composite_bg_kernel<<<n_rays, 64, 0, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_composite_bg");
}

// backward of svs_composite_bg with respect to sdf, rgb, density.beta, the bg density logits and the bg colours
// (d_weights, d_depth_values may be NULL); d_beta_ray (n_rays) is workspace, *d_beta_param receives the sum.
int svs_composite_bg_bwd(int n_rays, int n_samples, int n_bg, const float* z, const float* z_max, const float* sdf,
                         const float* rgb, const float* depth_scale, const float* beta_param, float beta_min,
                         const float* z_bg, const float* bg_out0, const float* bg_rgb, const float* d_rgb_values,
                         const float* d_weights, const float* d_depth_values, const float* d_depth_all,
                         const float* bg_depth, float* d_sdf, float* d_rgb,
                         float* d_bg_out0, float* d_bg_rgb, float* d_beta_ray, float* d_beta_param, void* hip_stream) {
  if (!z || !z_max || !sdf || !rgb || !depth_scale || !beta_param || !z_bg || !bg_out0 || !bg_rgb || !d_rgb_values || !d_sdf ||
      !d_rgb || !d_bg_out0 || !d_bg_rgb || !d_beta_ray || !d_beta_param || n_rays <= 0 || (d_depth_all && !bg_depth)) {
    set_error("svs_composite_bg_bwd: null/invalid argument"); return SVS_EINVAL;
  }
  if (n_samples < 2 || n_samples + 1 > kMaxS || n_bg < 2 || n_bg > 64) { set_error("svs_composite_bg_bwd: sample counts out of range"); return SVS_ESHAPE; }
  CompositeBgBwdArgs a{n_rays, n_samples, n_bg, z, z_max, sdf, rgb, depth_scale, beta_param, beta_min, z_bg, bg_out0, bg_rgb,
                       d_rgb_values, d_weights, d_depth_values, d_depth_all, bg_depth, d_sdf, d_rgb, d_bg_out0, d_bg_rgb,
                       d_beta_ray};
  hipStream_t s = (hipStream_t)hip_stream;
  composite_bg_bwd_kernel<<<n_rays, 64, 0, s>>>(a);
  beta_reduce_kernel<<<1, 256, 0, s>>>(d_beta_ray, n_rays, beta_param, d_beta_param, 0);
  return check_launch("svs_composite_bg_bwd");
}

int svs_composite_bwd(int n_rays, int n_samples, const float* z, const float* sdf, const float* rgb,
                      const float* depth_scale, const float* beta_param, float beta_min, const float* d_rgb_values,
                      const float* d_weights, const float* d_depth_values, float* d_sdf, float* d_rgb,
                      float* d_beta_ray, float* d_beta_param, void* hip_stream) {
  if (!z || !sdf || !rgb || !depth_scale || !beta_param || !d_rgb_values || !d_sdf || !d_rgb || !d_beta_ray ||
      !d_beta_param || n_rays <= 0) { set_error("svs_composite_bwd: null/invalid argument"); return SVS_EINVAL; }
  if (n_samples < 2 || n_samples > kMaxS) { set_error("svs_composite_bwd: n_samples must be in [2,%d]", kMaxS); return SVS_ESHAPE; }
  CompositeBwdArgs a{n_rays, n_samples, z, sdf, rgb, depth_scale, beta_param, beta_min, d_rgb_values, d_weights,
                     d_depth_values, d_sdf, d_rgb, d_beta_ray};
  hipStream_t s = (hipStream_t)hip_stream;
  composite_bwd_kernel<<<n_rays, 64, 0, s>>>(a);
  beta_reduce_kernel<<<1, 256, 0, s>>>(d_beta_ray, n_rays, beta_param, d_beta_param, 0);
  return check_launch("svs_composite_bwd");
}

int svs_cost_lookup(const float* xyz, const float* cam, const float* dirs, const float* z, int S, int n_points,
                    int n_views, int same_view, int inverse_depth, float img_w, float img_h, const float* view_params,
                    const float* const* cost, const float* const* z_near, const float* const* z_far, const int* dims,
                    float* pj, float* pi, unsigned char* valid, const int* same_view_dev, void* hip_stream) {
  if ((!xyz && !(cam && dirs && z && S > 0)) || !view_params || !cost || !z_near || !z_far || !dims || !pj || !pi ||
      !valid || n_points <= 0) { set_error("svs_cost_lookup: null/invalid argument"); return SVS_EINVAL; }
  if (n_views < 1 || n_views > kMaxViews) { set_error("svs_cost_lookup: 1..%d views", kMaxViews); return SVS_ESHAPE; }
  LookupArgs a;
  a.xyz = xyz; a.cam = cam; a.dirs = dirs; a.z = z; a.S = S > 0 ? S : 1; a.P = n_points; a.n_views = n_views;
  a.same_view = same_view; a.same_view_dev = same_view_dev; a.inverse_depth = inverse_depth;
  a.half_w = (img_w - 1.0f) / 2.0f; a.half_h = (img_h - 1.0f) / 2.0f;
  a.pj = pj; a.pi = pi; a.valid = valid;
  for (int j = 0; j < n_views; ++j) {
    const float* vp = view_params + 17 * j;   // HOST array: fx, fy, cx, cy, sk, c2w[12]
    a.v[j].fx = vp[0]; a.v[j].fy = vp[1]; a.v[j].cx = vp[2]; a.v[j].cy = vp[3]; a.v[j].sk = vp[4];
    for (int k = 0; k < 12; ++k) a.v[j].c2w[k] = vp[5 + k];
    a.v[j].cost = cost[j]; a.v[j].z_near = z_near[j]; a.v[j].z_far = z_far[j];
    a.v[j].D = dims[3 * j]; a.v[j].H = dims[3 * j + 1]; a.v[j].W = dims[3 * j + 2];
    if (!a.v[j].cost || !a.v[j].z_near || !a.v[j].z_far || a.v[j].D < 1 || a.v[j].H < 1 || a.v[j].W < 1) {
      set_error("svs_cost_lookup: bad view %d", j); return SVS_EINVAL;
    }
  }
  cost_lookup_kernel<<<(n_points + 63) / 64, 64 * n_views, 0, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_cost_lookup");
}

int svs_loss(int n_rays, int n_samples, int n_eik, const float* rgb_values, const float* rgb_target,
             const float* grad_theta, const float* weights, const float* pi, const float* pj, const float* depth_values,
             float rgb_weight, float eikonal_weight, float mvs_weight, float sparse_weight, float gce, float confi,
             int annealed, float anneal_sparse, int n_rays_norm, int n_eik_norm, float* losses, float* d_rgb_values,
             float* d_grad_theta, float* d_weights, float* d_depth_values, double* workspace, const float* anneal_dev,
             void* hip_stream) {
  if (!rgb_values || !rgb_target || !weights || !depth_values || !losses || !d_rgb_values || !d_weights ||
      !d_depth_values || !workspace || n_rays <= 0 || n_samples <= 0 || n_samples > 256 || (n_eik > 0 && (!grad_theta || !d_grad_theta)) || (!pi != !pj)) {
    set_error("svs_loss: null/invalid argument"); return SVS_EINVAL;
  }
  LossArgs a{n_rays, n_samples, n_eik, n_rays_norm > 0 ? n_rays_norm : n_rays, n_eik_norm > 0 ? n_eik_norm : n_eik, rgb_values, rgb_target, grad_theta, weights, pi, pj, depth_values, rgb_weight,
             eikonal_weight, mvs_weight, sparse_weight, gce, confi, anneal_sparse, annealed, anneal_dev, losses, d_rgb_values,
             d_grad_theta, d_weights, d_depth_values};
  const int n_units = n_rays + (n_eik + 63) / 64;
  loss_rays_kernel<<<(n_units + 3) / 4, 256, 0, (hipStream_t)hip_stream>>>(a, workspace);
  loss_reduce_kernel<<<1, 256, 0, (hipStream_t)hip_stream>>>(a, workspace, n_units);
  return check_launch("svs_loss");
}

}  // extern "C"
