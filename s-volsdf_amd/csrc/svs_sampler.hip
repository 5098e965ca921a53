// Error-bounded ray sampler (VolSDF Algorithm 1) for gfx950: one ray per wavefront, interval state in LDS.
//
// Reference: volsdf/model/ray_sampler.py:22-43 (UniformSampler), :67-219 (ErrorBoundSampler.get_z_vals),
// :221-229 (get_error_bound); volsdf/model/density.py:21-30; volsdf/utils/rend_util.py:200-216.
//
// Numeric contract: float32 IEEE ops in the reference's order with no fma contraction (this
// file is compiled with -ffp-contract=off); exp / expm1 = svs::sleef_expf / sleef_expm1f (the Sleef u10 routines torch-CPU
// evaluates, restated); the pdf's row sum = svs::aten_row_sum (ATen's cascade_sum order); cumsum accumulates in
// float64 and rounds every prefix, as torch-CPU cumsum does (blocked over the 64 lanes).  With that the kernels
// reproduce oracle/svs_oracle.py -- and through it the reference's CPU path -- bit for bit, indices included.
//
// Control flow: the reference's data-dependent `while` (batch-global `beta.max() > beta0`, :136) becomes
// device-side flags: round-A kernels OR the per-ray convergence test into conv_flag[i], round-B kernels read
// it, decide "up-sample or final" and publish active[i+1] for the next round's kernels (incl. the gated MLP
// launch).  All rounds are enqueued without a host sync.
#include "svs_common.h"

namespace svs {
namespace sampler {

constexpr int kCap = 768;     // max bins held per ray (N_samples_eval * (max_total_iters + 1) <= kCap)
constexpr int kMaxNew = 128;  // max new samples per round (N_samples_eval)

struct Ctl {            // device-side control block, one per group of rays (the reference decides "converged" per
                        // forward call = per chunk of split_n_pixels rays; a launch can cover many such groups)
  int conv_flag[8];     // OR over rays of (beta > beta0) in round i
  int active[8];        // round i runs (0/1)
  int final_round;      // round whose B kernel produced the final samples (-1: none yet)
};

__device__ __forceinline__ float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// torch.linspace(start,end,n)[i] in float32 (ATen: step=(end-start)/(n-1); fma(step,i,start) for i<n/2,
// fma(-step, n-1-i, end) otherwise)
__device__ __forceinline__ float linspace_at(float start, float end, int n, int i) {
  const float step = (end - start) / (float)(n - 1);
  return i < n / 2 ? fma32(step, (float)i, start) : fma32(-step, (float)(n - 1 - i), end);
}

// Canonical inclusive cumsum of m floats held in LDS (in -> out, may alias), float64 accumulation.
// Returns the total (prefix m-1) rounded to float32, in every lane.  Caller syncs before and after.
__device__ __forceinline__ float wave_cumsum(const float* in, float* out, int m, int lane) {
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
  const int owner = (m - 1) / c;
  return __shfl(last, owner);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, d));
  return v;
}

struct RayLds {
  float z[kCap];
  float sdf[kCap];
  float dists[kCap];
  float dstar[kCap];
  float t0[kCap];
  float t1[kCap];
};

// d* per interval (Theorem 1), ray_sampler.py:97-111
__device__ __forceinline__ void compute_dstar(RayLds& L, int n, int lane) {
  for (int i = lane; i < n - 1; i += 64) {
    const float a = L.z[i + 1] - L.z[i];
    const float d0 = L.sdf[i], d1 = L.sdf[i + 1];
    const float b = __builtin_fabsf(d0), c = __builtin_fabsf(d1);
    const float a2 = a * a, b2 = b * b, c2 = c * c;
    const bool first = (a2 + b2) <= c2;
    const bool second = (a2 + c2) <= b2;
    float ds = 0.0f;
    if (first) ds = b;
    if (second) ds = c;
    const float s = ((a + b) + c) / 2.0f;
    const float area = ((s * (s - a)) * (s - b)) * (s - c);
    if (!first && !second && ((b + c) - a > 0.0f)) ds = (2.0f * __builtin_sqrtf(area)) / a;
    const float sg0 = d0 > 0.0f ? 1.0f : (d0 < 0.0f ? -1.0f : 0.0f);
    const float sg1 = d1 > 0.0f ? 1.0f : (d1 < 0.0f ? -1.0f : 0.0f);
    const float same = (sg1 * sg0 == 1.0f) ? 1.0f : 0.0f;
    L.dists[i] = a;
    L.dstar[i] = same * ds;
  }
}

// get_error_bound (ray_sampler.py:221-229) for one ray, beta wave-uniform.  Uses t0/t1 as scratch.
__device__ __forceinline__ float error_bound(RayLds& L, int n, float beta, int lane) {
  const float four_b2 = 4.0f * (beta * beta);
  for (int i = lane; i < n; i += 64) {
    L.t0[i] = i == 0 ? 0.0f : L.dists[i - 1] * laplace_density(L.sdf[i - 1], beta);
    if (i < n - 1) L.t1[i] = (sleef_expf(-L.dstar[i] / beta) * (L.dists[i] * L.dists[i])) / four_b2;
  }
  __syncthreads();
  wave_cumsum(L.t0, L.t0, n, lane);
  wave_cumsum(L.t1, L.t1, n - 1, lane);
  __syncthreads();
  float m = -__builtin_inff();
  bool has_nan = false;
  for (int i = lane; i < n - 1; i += 64) {
    const float e = sleef_expf(L.t1[i]);
    const float bo = ((e > 1.0e6f ? 1.0e6f : e) - 1.0f) * sleef_expf(-L.t0[i]);
    has_nan |= (bo != bo);
    m = __builtin_fmaxf(m, bo);
  }
  __syncthreads();
  m = wave_max(m);
  if (__any(has_nan)) m = __builtin_nanf("");   // torch.max propagates NaN
  return m;
}

struct InitArgs {
  const float* cam; int cam_stride; const float* dirs;  // rays
  int R;
  int n_eval;                 // N_samples_eval
  float near, far;            // far used when !sphere_far
  int sphere_far;             // inverse_sphere_bg: far = sphere exit (rend_util.py:200-216)
  float sphere_radius;
  const float* jitter;        // (R, n_eval) uniform draws in train mode, or nullptr
  float inv_4log;             // 1/(4 log(1+eps)) computed by the host in float32 (ray_sampler.py:77)
  float* samples;             // out (R, kMaxNew): the uniform z (round-0 samples)
  float* beta;                // out (R): Lemma-2 bound
  float* far_out;             // out (R): far per ray (finalize needs it)
  Ctl* ctl;                   // [ceil(R / group_rays)]
  int group_rays;             // rays per convergence group
  int max_iters;
  int* err;                   // set to 1 on a bounding-sphere miss (the reference calls exit())
};

__global__ __launch_bounds__(64) void init_kernel(InitArgs a) {
  __shared__ float zs[kMaxNew];
  __shared__ float ds[kMaxNew];
  const int r = blockIdx.x, lane = threadIdx.x;
  if (r % a.group_rays == 0 && lane < 8) {
    Ctl* c = a.ctl + r / a.group_rays;
    c->conv_flag[lane] = 0;
    c->active[lane] = (lane == 0 && a.max_iters > 0) ? 1 : 0;
    if (lane == 0) c->final_round = -1;
  }
  float far = a.far;
  if (a.sphere_far) {
    const float* o = a.cam + (size_t)r * a.cam_stride;
    const float* d = a.dirs + 3 * (size_t)r;
    const float dot = (d[0] * o[0] + d[1] * o[1]) + d[2] * o[2];
    const float nrm = norm3(o[0], o[1], o[2]);            // cam_loc.norm(2, 1) ** 2: the square of the ROUNDED norm
    const float under = dot * dot - (nrm * nrm - a.sphere_radius * a.sphere_radius);
    if (under <= 0.0f) { if (lane == 0) *a.err = 1; }
    far = __builtin_fmaxf(__builtin_sqrtf(under) - dot, 0.0f);
  }
  if (lane == 0) a.far_out[r] = far;
  const int n = a.n_eval;
  for (int i = lane; i < n; i += 64) {
    const float t = linspace_at(0.0f, 1.0f, n, i);
    zs[i] = a.near * (1.0f - t) + far * t;
  }
  __syncthreads();
  if (a.jitter) {
    float znew[(kMaxNew + 63) / 64];
    int k = 0;
    for (int i = lane; i < n; i += 64, ++k) {
      const float lower = i == 0 ? zs[0] : 0.5f * (zs[i] + zs[i - 1]);
      const float upper = i == n - 1 ? zs[n - 1] : 0.5f * (zs[i + 1] + zs[i]);
      znew[k] = lower + (upper - lower) * a.jitter[(size_t)r * n + i];
    }
    __syncthreads();
    k = 0;
    for (int i = lane; i < n; i += 64, ++k) zs[i] = znew[k];
    __syncthreads();
  }
  for (int i = lane; i < n; i += 64) {
    a.samples[(size_t)r * kMaxNew + i] = zs[i];
    if (i < n - 1) { const float d = zs[i + 1] - zs[i]; ds[i] = d * d; }
  }
  __syncthreads();
  const float tot = aten_row_sum(ds, n - 1, lane);       // (dists ** 2).sum(-1), ray_sampler.py:77
  if (lane == 0) a.beta[r] = __builtin_sqrtf(a.inv_4log * tot);
}

struct RoundArgs {
  int R, round, max_iters;
  int n_eval, n_final, n_extra;
  const float* beta_param; float beta_min;   // beta0 = |*beta_param| + beta_min (density.py:28-30), read on the device
  float eps; int beta_iters; float add_tiny;
  float near;
  const float* far;            // (R)
  float* z;                    // (R, kCap) bins
  float* sdf;                  // (R, kCap)
  float* beta;                 // (R)
  float* samples;              // (R, kMaxNew) new sample positions (in: round A; out: round B when up-sampling)
  const float* samples_sdf;    // (R, kMaxNew) sdf of `samples` (round A)
  Ctl* ctl;                    // [ceil(R / group_rays)]
  int group_rays;
  // final sampling
  const float* u_final;        // (R, n_final) train-mode draws or nullptr (linspace)
  const int* extra_idx;        // (n_extra) train-mode bins (randperm[:n_extra]) or nullptr (linspace idx)
  const int* eik_idx;          // (R) train-mode pick of the eikonal sample or nullptr
  float* z_final;              // (R, n_out), n_out = n_final(or n_eval when max_iters==0) + n_extra + 2
  float* z_eik;                // (R)
  // optional parity/debug outputs
  int* dbg_samples_idx;        // (R, kCap) merged order of round A
  int* dbg_inds;               // (R, kMaxNew) searchsorted indices of round B
  float* dbg_cdf;              // (R, kCap)
  float* dbg_weights;          // (R, kCap)
};

// Round A: merge the new samples (stable, torch.sort semantics with ties to the lower index, :189-190 and
// the sdf gather of :90-93), d* (:97-111), beta line search (:114-123).
__global__ __launch_bounds__(64) void round_a_kernel(RoundArgs a) {
  __shared__ RayLds L;
  const int r = blockIdx.x, lane = threadIdx.x, i_round = a.round;
  Ctl* ctl = a.ctl + r / a.group_rays;
  if (!ctl->active[i_round]) return;
  const int n_old = a.n_eval * i_round, n_new = a.n_eval, n = n_old + n_new;
  float* zrow = a.z + (size_t)r * kCap;
  float* srow = a.sdf + (size_t)r * kCap;
  // stage old bins in t0/t1, new samples in dists/dstar
  for (int i = lane; i < n_old; i += 64) { L.t0[i] = zrow[i]; L.t1[i] = srow[i]; }
  for (int i = lane; i < n_new; i += 64) {
    L.dists[i] = a.samples[(size_t)r * kMaxNew + i];
    L.dstar[i] = a.samples_sdf[(size_t)r * kMaxNew + i];
  }
  __syncthreads();
  for (int e = lane; e < n; e += 64) {
    int pos;
    float zv, sv;
    if (e < n_old) {
      zv = L.t0[e]; sv = L.t1[e];
      int cnt = 0;
      for (int k = 0; k < n_new; ++k) cnt += (L.dists[k] < zv) ? 1 : 0;
      pos = e + cnt;
    } else {
      const int j = e - n_old;
      zv = L.dists[j]; sv = L.dstar[j];
      int lo = 0, hi = n_old;          // number of old bins <= zv
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.t0[mid] <= zv) lo = mid + 1; else hi = mid; }
      int cnt = lo;
      for (int k = 0; k < n_new; ++k) {
        const float o = L.dists[k];
        cnt += (o < zv || (o == zv && k < j)) ? 1 : 0;
      }
      pos = cnt;
    }
    L.z[pos] = zv; L.sdf[pos] = sv;
    if (a.dbg_samples_idx) a.dbg_samples_idx[(size_t)r * kCap + pos] = e;
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) { zrow[i] = L.z[i]; srow[i] = L.sdf[i]; }
  compute_dstar(L, n, lane);
  __syncthreads();

  float beta = a.beta[r];
  const float beta0 = __builtin_fabsf(*a.beta_param) + a.beta_min;
  float err = error_bound(L, n, beta0, lane);
  if (err <= a.eps) beta = beta0;
  float bmin = beta0, bmax = beta;
  for (int it = 0; it < a.beta_iters; ++it) {
    const float mid = (bmin + bmax) / 2.0f;
    err = error_bound(L, n, mid, lane);
    if (err <= a.eps) bmax = mid;
    if (err > a.eps) bmin = mid;
  }
  beta = bmax;
  if (lane == 0) {
    a.beta[r] = beta;
    if (beta > beta0) atomicOr(&ctl->conv_flag[i_round], 1);
  }
}

// extras + final sort + eikonal pick (ray_sampler.py:192-212).  samples in L.t0[0..ns), bins in L.z[0..n).
__device__ __forceinline__ void finalize(const RoundArgs& a, RayLds& L, int r, int n, int ns, int lane) {
  const int m = ns + a.n_extra + 2;
  for (int i = lane; i < a.n_extra + 2; i += 64) {
    float v;
    if (i == 0) v = a.near;
    else if (i == 1) v = a.far[r];
    else {
      int idx;
      if (a.extra_idx) idx = a.extra_idx[i - 2];
      else idx = (int)linspace_at(0.0f, (float)(n - 1), a.n_extra, i - 2);   // linspace(0,n-1,k).long()
      v = L.z[idx];
    }
    L.t0[ns + i] = v;
  }
  __syncthreads();
  for (int e = lane; e < m; e += 64) {
    const float v = L.t0[e];
    int cnt = 0;
    for (int k = 0; k < m; ++k) { const float o = L.t0[k]; cnt += (o < v || (o == v && k < e)) ? 1 : 0; }
    L.t1[cnt] = v;
  }
  __syncthreads();
  for (int i = lane; i < m; i += 64) a.z_final[(size_t)r * m + i] = L.t1[i];
  if (lane == 0) a.z_eik[r] = L.t1[a.eik_idx ? a.eik_idx[r] : 0];
}

// Round B: weights (:126-132), convergence decision (:135-138), pdf/cdf (:138-163), inverse CDF (:166-185);
// when this is the last round also the final sample set.
__global__ __launch_bounds__(64) void round_b_kernel(RoundArgs a) {
  __shared__ RayLds L;
  const int r = blockIdx.x, lane = threadIdx.x, i_round = a.round;
  Ctl* ctl = a.ctl + r / a.group_rays;
  if (!ctl->active[i_round]) return;
  const bool upsample = ctl->conv_flag[i_round] != 0 && (i_round + 1 < a.max_iters);
  if (r % a.group_rays == 0 && lane == 0) {
    ctl->active[i_round + 1] = upsample ? 1 : 0;
    if (!upsample) ctl->final_round = i_round;
  }
  const int n = a.n_eval * (i_round + 1);
  const float* zrow = a.z + (size_t)r * kCap;
  const float* srow = a.sdf + (size_t)r * kCap;
  for (int i = lane; i < n; i += 64) { L.z[i] = zrow[i]; L.sdf[i] = srow[i]; }
  __syncthreads();
  compute_dstar(L, n, lane);
  __syncthreads();
  const float beta = a.beta[r];
  // free energy, transmittance, weights
  for (int i = lane; i < n; i += 64) {
    L.t0[i] = i == 0 ? 0.0f : L.dists[i - 1] * laplace_density(L.sdf[i - 1], beta);   // shifted free energy
  }
  __syncthreads();
  wave_cumsum(L.t0, L.t0, n, lane);
  __syncthreads();
  for (int i = lane; i < n; i += 64) L.t0[i] = sleef_expf(-L.t0[i]);      // transmittance
  __syncthreads();
  const int N = upsample ? a.n_eval : a.n_final;
  if (upsample) {
    const float four_b2 = 4.0f * (beta * beta);
    for (int i = lane; i < n - 1; i += 64)
      L.t1[i] = (sleef_expf(-L.dstar[i] / beta) * (L.dists[i] * L.dists[i])) / four_b2;
    __syncthreads();
    wave_cumsum(L.t1, L.t1, n - 1, lane);
    __syncthreads();
    for (int i = lane; i < n - 1; i += 64) {
      const float e = sleef_expf(L.t1[i]);
      L.t1[i] = ((e > 1.0e6f ? 1.0e6f : e) - 1.0f) * L.t0[i] + a.add_tiny;
    }
  } else {
    for (int i = lane; i < n - 1; i += 64) {
      const float fe = L.dists[i] * laplace_density(L.sdf[i], beta);
      const float w = (1.0f - sleef_expf(-fe)) * L.t0[i];
      L.t1[i] = w + 1e-5f;
      if (a.dbg_weights) a.dbg_weights[(size_t)r * kCap + i] = w;
    }
  }
  __syncthreads();
  // pdf / sum, cdf = [0, cumsum(pdf)]  -> L.t1[0..n)
  const float tot = aten_row_sum(L.t1, n - 1, lane);          // torch.sum(pdf, -1), ray_sampler.py:149,161
  __syncthreads();
  for (int i = lane; i < n - 1; i += 64) L.t1[i] = L.t1[i] / tot;
  __syncthreads();
  wave_cumsum(L.t1, L.dstar, n - 1, lane);                    // dstar no longer needed: holds cumsum(pdf)
  __syncthreads();
  for (int i = lane; i < n; i += 64) {
    const float c = i == 0 ? 0.0f : L.dstar[i - 1];
    L.t1[i] = c;
    if (a.dbg_cdf) a.dbg_cdf[(size_t)r * kCap + i] = c;
  }
  __syncthreads();
  // inverse CDF
  for (int j = lane; j < N; j += 64) {
    float u;
    if (upsample || !a.u_final) u = linspace_at(0.0f, 1.0f, N, j);
    else u = a.u_final[(size_t)r * N + j];
    int lo = 0, hi = n;                 // searchsorted(right=True): #cdf <= u
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.t1[mid] <= u) lo = mid + 1; else hi = mid; }
    const int inds = lo;
    const int below = inds - 1 > 0 ? inds - 1 : 0;
    const int above = inds < n - 1 ? inds : n - 1;
    const float cb = L.t1[below], ca = L.t1[above];
    const float zb = L.z[below], za = L.z[above];
    float denom = ca - cb;
    if (denom < 1e-5f) denom = 1.0f;
    const float t = (u - cb) / denom;
    L.dstar[j] = zb + t * (za - zb);
    if (a.dbg_inds) a.dbg_inds[(size_t)r * kMaxNew + j] = inds;
  }
  __syncthreads();
  if (upsample) {
    for (int j = lane; j < N; j += 64) a.samples[(size_t)r * kMaxNew + j] = L.dstar[j];
  } else {
    for (int j = lane; j < N; j += 64) L.t0[j] = L.dstar[j];
    __syncthreads();
    finalize(a, L, r, n, N, lane);
  }
}

// max_total_iters == 0 (fast = 0): the uniform samples are the sample set (ray_sampler.py:192-208)
__global__ __launch_bounds__(64) void finalize0_kernel(RoundArgs a) {
  __shared__ RayLds L;
  const int r = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < a.n_eval; i += 64) {
    const float v = a.samples[(size_t)r * kMaxNew + i];
    L.z[i] = v; L.t0[i] = v;
  }
  __syncthreads();
  finalize(a, L, r, a.n_eval, a.n_eval, lane);
}

}  // namespace sampler
}  // namespace svs

using namespace svs;
using namespace svs::sampler;

extern "C" {

size_t svs_sampler_ctl_bytes(int n_rays, int group_rays) {
  const int g = group_rays > 0 ? group_rays : (n_rays > 0 ? n_rays : 1);
  return (size_t)((n_rays + g - 1) / g) * sizeof(Ctl);
}
int svs_sampler_ctl_stride(void) { return (int)(sizeof(Ctl) / sizeof(int)); }
int svs_sampler_cap(void) { return kCap; }
int svs_sampler_max_new(void) { return kMaxNew; }

int svs_sampler_init(const float* cam, int cam_stride, const float* dirs, int n_rays, int n_eval, float near_, float far_,
                     int sphere_far, float sphere_radius, const float* jitter, float inv_4log, int max_iters,
                     float* samples, float* beta, float* far_out, void* ctl, int group_rays, int* err_flag,
                     void* hip_stream) {
  if (group_rays <= 0) group_rays = n_rays;
  if (!cam || !dirs || !samples || !beta || !far_out || !ctl || !err_flag || n_rays <= 0) {
    set_error("svs_sampler_init: null/invalid argument"); return SVS_EINVAL;
  }
  if (n_eval < 2 || n_eval > kMaxNew || max_iters < 0 || max_iters > 7 || n_eval * (max_iters > 0 ? max_iters : 1) > kCap) {
    set_error("svs_sampler_init: n_eval=%d max_iters=%d exceed the kernel limits (%d new / %d bins)", n_eval, max_iters, kMaxNew, kCap);
    return SVS_ESHAPE;
  }
  InitArgs a{cam, cam_stride, dirs, n_rays, n_eval, near_, far_, sphere_far, sphere_radius, jitter, inv_4log,
             samples, beta, far_out, (Ctl*)ctl, group_rays, max_iters, err_flag};
  init_kernel<<<n_rays, 64, 0, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_sampler_init");
}

// phase: 0 = round A, 1 = round B, 2 = finalize for max_iters == 0
int svs_sampler_round(int phase, int n_rays, int round, int max_iters, int n_eval, int n_final, int n_extra,
                      const float* beta_param, float beta_min, float eps, int beta_iters, float add_tiny, float near_,
                      const float* far_,
                      float* z, float* sdf, float* beta, float* samples, const float* samples_sdf, void* ctl,
                      int group_rays, const float* u_final, const int* extra_idx, const int* eik_idx, float* z_final, float* z_eik,
                      int* dbg_samples_idx, int* dbg_inds, float* dbg_cdf, float* dbg_weights, void* hip_stream) {
  if (!beta_param || !far_ || !z || !sdf || !beta || !samples || !ctl || !z_final || !z_eik || n_rays <= 0 || round < 0 || round > 6) {
    set_error("svs_sampler_round: null/invalid argument"); return SVS_EINVAL;
  }
  if (group_rays <= 0) group_rays = n_rays;
  if (n_final > kMaxNew || n_final + n_extra + 2 > kCap || n_eval + n_extra + 2 > kCap) {
    set_error("svs_sampler_round: sample counts exceed kernel limits"); return SVS_ESHAPE;
  }
  RoundArgs a{n_rays, round, max_iters, n_eval, n_final, n_extra, beta_param, beta_min, eps, beta_iters, add_tiny, near_, far_,
              z, sdf, beta, samples, samples_sdf, (Ctl*)ctl, group_rays, u_final, extra_idx, eik_idx, z_final, z_eik,
              dbg_samples_idx, dbg_inds, dbg_cdf, dbg_weights};
  hipStream_t s = (hipStream_t)hip_stream;
  if (phase == 0) {
    if (!samples_sdf) { set_error("svs_sampler_round: round A needs samples_sdf"); return SVS_EINVAL; }
    round_a_kernel<<<n_rays, 64, 0, s>>>(a);
  } else if (phase == 1) {
    round_b_kernel<<<n_rays, 64, 0, s>>>(a);
  } else {
    finalize0_kernel<<<n_rays, 64, 0, s>>>(a);
  }
  return check_launch("svs_sampler_round");
}

// bit-exactness self-test hooks for the numeric contract (tests/test_gpu_numeric_contract.py)
__global__ void selftest_exp_kernel(const float* x, float* y_exp, float* y_expm1, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { y_exp[i] = sleef_expf(x[i]); y_expm1[i] = sleef_expm1f(x[i]); }
}
__global__ void selftest_arith_kernel(const float* a, const float* b, float* q, float* s, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { q[i] = a[i] / b[i]; s[i] = __builtin_sqrtf(__builtin_fabsf(a[i])); }
}
__global__ __launch_bounds__(64) void selftest_cumsum_kernel(const float* x, float* y, float* tot, int m) {
  __shared__ float buf[kCap];
  const int lane = threadIdx.x;
  for (int i = lane; i < m; i += 64) buf[i] = x[(size_t)blockIdx.x * m + i];
  __syncthreads();
  const float t = wave_cumsum(buf, buf, m, lane);
  __syncthreads();
  for (int i = lane; i < m; i += 64) y[(size_t)blockIdx.x * m + i] = buf[i];
  if (lane == 0) tot[blockIdx.x] = t;
}

__global__ __launch_bounds__(64) void selftest_rowsum_kernel(const float* x, float* tot, int m) {
  extern __shared__ float rowbuf[];
  const int lane = threadIdx.x;
  for (int i = lane; i < m; i += 64) rowbuf[i] = x[(size_t)blockIdx.x * m + i];
  __syncthreads();
  const float t = aten_row_sum(rowbuf, m, lane);
  if (lane == 0) tot[blockIdx.x] = t;
}

int svs_selftest_rowsum(const float* x, float* tot, int rows, int m, void* hip_stream) {
  if (m < 1 || m > 16000) { set_error("svs_selftest_rowsum: m out of range (1..16000)"); return SVS_ESHAPE; }
  selftest_rowsum_kernel<<<rows, 64, (size_t)m * sizeof(float), (hipStream_t)hip_stream>>>(x, tot, m);
  return check_launch("svs_selftest_rowsum");
}
int svs_selftest_exp(const float* x, float* y_exp, float* y_expm1, int n, void* hip_stream) {
  selftest_exp_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(x, y_exp, y_expm1, n);
  return check_launch("svs_selftest_exp");
}
int svs_selftest_arith(const float* a, const float* b, float* q, float* s, int n, void* hip_stream) {
  selftest_arith_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(a, b, q, s, n);
  return check_launch("svs_selftest_arith");
}
int svs_selftest_cumsum(const float* x, float* y, float* tot, int rows, int m, void* hip_stream) {
  if (m < 1 || m > kCap) { set_error("svs_selftest_cumsum: m out of range"); return SVS_ESHAPE; }
  selftest_cumsum_kernel<<<rows, 64, 0, (hipStream_t)hip_stream>>>(x, y, tot, m);
  return check_launch("svs_selftest_cumsum");
}

}  // extern "C"
