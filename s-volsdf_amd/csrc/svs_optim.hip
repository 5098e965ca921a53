// Fused optimiser step on a flat float32 parameter buffer: global-norm gradient clipping, NaN/Inf guard and Adam,
// with no host synchronisation.
//
// Reference: volsdf/vsdf.py:214-219 -- torch.nn.utils.clip_grad_norm_(params, 1.0), on_after_backward (:454-463: if any
// gradient entry is NaN/Inf the gradients are zeroed and the step still runs, as torch 1.9's zero_grad() does),
// torch.optim.Adam(lr=5e-4) (:101-102; betas (0.9,0.999), eps 1e-8, no weight decay).
#include "svs_common.h"

namespace svs {
namespace optim {

constexpr int kBlocks = 256;

// stage 1: per-block partial sum of squares (float64) and non-finite count
__global__ __launch_bounds__(256) void grad_stats_kernel(const float* __restrict__ g, long long n, double* __restrict__ part,
                                                         int* __restrict__ bad, int* __restrict__ step_counter) {
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;      // read by adam_kernel (next launch)
  __shared__ double sh[4];
  __shared__ int shb[4];
  double acc = 0.0;
  int nb = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = g[i];
    if (!(__builtin_fabsf(v) <= 3.4028234e38f)) nb = 1;     // NaN or Inf
    acc += (double)v * (double)v;
  }
  for (int d = 32; d >= 1; d >>= 1) { acc += __shfl_xor(acc, d); nb |= __shfl_xor(nb, d); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = acc; shb[threadIdx.x >> 6] = nb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    bad[blockIdx.x] = shb[0] | shb[1] | shb[2] | shb[3];
  }
}

// stage 2: every block reduces the partials in the same fixed order (reproducible), then updates its slice.
// Scalars arrive as the float32 values torch's kernels receive: torch keeps lr / betas / eps as Python floats
// (float64), forms 1 - beta, the bias corrections and lr / bias_correction1 in float64 and rounds once when the scalar
// enters a float32 kernel.  (1.0f - 0.999f is 4.7e-5 larger than float(1 - 0.999).)
struct AdamScalars {
  float max_norm, beta1, omb1, beta2, omb2, eps;
  double lr, beta1_d, beta2_d;
};

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, const double* __restrict__ part,
                                                   const int* __restrict__ bad, int n_part, AdamScalars sc, int step,
                                                   const int* __restrict__ step_dev, float* __restrict__ info) {
  // every wave reduces the partials itself, in one fixed order (lane l takes l, l + 64, ...; then a butterfly): the same
  // bits in every wave of every block, without the 256-term serial sum each thread used to walk through
  double tot = 0.0;
  int nb = 0;
  for (int i = threadIdx.x & 63; i < n_part; i += 64) { tot += part[i]; nb |= bad[i]; }
  for (int d = 32; d >= 1; d >>= 1) { tot += __shfl_xor(tot, d); nb |= __shfl_xor(nb, d); }
  if (step_dev) step = *step_dev;                      // graph replays: the counter lives on the device (grad_stats bumped it)
  const double bc1 = 1.0 - pow(sc.beta1_d, (double)step);
  const double bc2 = 1.0 - pow(sc.beta2_d, (double)step);
  const float step_size = (float)(sc.lr / bc1), bc2_sqrt = (float)__builtin_sqrt(bc2);
  const float total_norm = (float)__builtin_sqrt(tot);
  float coef = sc.max_norm > 0.0f ? sc.max_norm / (total_norm + 1e-6f) : 1.0f;     // clip_grad_norm_
  coef = coef > 1.0f ? 1.0f : coef;
  const bool drop = nb != 0 || !(total_norm <= 3.4028234e38f);
  if (blockIdx.x == 0 && threadIdx.x == 0 && info) { info[0] = total_norm; info[1] = drop ? 1.0f : 0.0f; }
  auto update = [&](float& pi, float& gi_io, float& mi_io, float& vi_io) {
    const float gi = drop ? 0.0f : gi_io * coef;
    gi_io = gi;                                         // the clipped (or zeroed) gradient stays visible, as in torch
    const float mi = sc.beta1 * mi_io + sc.omb1 * gi;
    const float vi = sc.beta2 * vi_io + (sc.omb2 * gi) * gi;     // addcmul_(grad, grad, value = 1 - beta2): value * g * g
    mi_io = mi; vi_io = vi;
    const float denom = __builtin_sqrtf(vi) / bc2_sqrt + sc.eps;
    pi = pi - step_size * (mi / denom);
  };
  // four consecutive entries per thread and pass (16-byte accesses; the buffers are torch allocations, 256-byte aligned);
  // the grid covers the buffer in one or two passes -- with 12 scalar passes per thread the launch took 18-28 us for
  // 0.8-1.4 M parameters, most of it the latency of a dozen dependent round trips
  const long long n4 = n >> 2;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float a = pp[c], b = gg[c], d = mm[c], e = vv[c];
      update(a, b, d, e);
      pp[c] = a; gg[c] = b; mm[c] = d; vv[c] = e;
    }
    p4[i] = pp; g4[i] = gg; m4[i] = mm; v4[i] = vv;
  }
  if (blockIdx.x == 0) {
    const long long i = (n4 << 2) + threadIdx.x;
    if (i < n) update(p[i], g[i], m[i], v[i]);
  }
}

}  // namespace optim
}  // namespace svs

using namespace svs;
using namespace svs::optim;

extern "C" {

size_t svs_adam_workspace_bytes(void) { return kBlocks * (sizeof(double) + sizeof(int)); }

// step: 1-based Adam step count, or -- when step_counter (device int) is given -- ignored: the launch increments the
// counter and uses the new value, so that a captured launch sequence (hipGraph) replays with an advancing step.
// lr / betas / eps are float64 like torch's hyper-parameters.  workspace: svs_adam_workspace_bytes().  info (2
// floats, may be NULL): the gradient norm before clipping and whether the gradient was dropped (non-finite).
int svs_clip_guard_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step,
                        int* step_counter, double max_norm, double lr, double beta1, double beta2, double eps,
                        void* workspace, float* info, void* hip_stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !workspace || n <= 0 || (!step_counter && step < 1)) {
    set_error("svs_clip_guard_adam: bad argument"); return SVS_EINVAL;
  }
  double* part = (double*)workspace;
  int* bad = (int*)(part + kBlocks);
  hipStream_t s = (hipStream_t)hip_stream;
  grad_stats_kernel<<<kBlocks, 256, 0, s>>>(grads, n, part, bad, step_counter);
  AdamScalars sc;
  sc.max_norm = (float)max_norm; sc.beta1 = (float)beta1; sc.omb1 = (float)(1.0 - beta1); sc.beta2 = (float)beta2;
  sc.omb2 = (float)(1.0 - beta2); sc.eps = (float)eps; sc.lr = lr; sc.beta1_d = beta1; sc.beta2_d = beta2;
  const bool aligned = (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
  if (!aligned) { set_error("svs_clip_guard_adam: the four buffers must be 16-byte aligned"); return SVS_EINVAL; }
  long long blocks = ((n >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  adam_kernel<<<(int)blocks, 256, 0, s>>>(params, grads, exp_avg, exp_avg_sq, n, part, bad, kBlocks, sc, step, step_counter, info);
  return check_launch("svs_clip_guard_adam");
}

}  // extern "C"
