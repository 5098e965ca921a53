// Weight packing for the fused MLP kernels: weight-norm materialisation (w = g * v / ||v||_row,
// volsdf/model/network.py:64-65) and the permutation of every layer into the order in which the MFMA
// k-loops of svs_mlp.hip consume it (chunk format: svs_mlp_layout.h).  Runs once per optimisation step.
#include "svs_common.h"
#include "svs_mlp_layout.h"

namespace svs {
namespace mlp {

struct LayerPtrs {
  const float* v[9];   // weight_v (or plain weight) [out][in] row-major
  const float* g[9];   // weight_g [out] or nullptr (no weight-norm)
  const float* b[9];   // bias [out]
};

constexpr int kScaleStride = 264;

__device__ __host__ constexpr int sdf_rows(int l) { return l == 3 ? 217 : (l == 8 ? 257 : 256); }
__device__ __host__ constexpr int sdf_cols(int l) { return l == 0 ? 39 : 256; }
__device__ __host__ constexpr int rgb_rows(int l) { return l == 4 ? 3 : 256; }
__device__ __host__ constexpr int rgb_cols(int l) { return l == 0 ? 271 : 256; }

// scale[l][o] = g[o] / ||v[o,:]||  (1 when the layer has no weight-norm); one wave per row
__global__ void rownorm_kernel(LayerPtrs w, int n_layers, int is_rgb, float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int row_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int l = row_global / kScaleStride, o = row_global % kScaleStride;
  if (l >= n_layers) return;
  const int rows = is_rgb ? rgb_rows(l) : sdf_rows(l);
  const int cols = is_rgb ? rgb_cols(l) : sdf_cols(l);
  if (o >= rows) return;
  float s = 1.0f;
  if (w.g[l]) {
    double acc = 0.0;
    for (int c = lane; c < cols; c += 64) {
      const double x = (double)w.v[l][(size_t)o * cols + c];
      acc += x * x;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    s = w.g[l][o] / (float)__builtin_sqrt(acc);
  }
  if (lane == 0) scale[l * kScaleStride + o] = s;
}

__device__ __forceinline__ float weff(const LayerPtrs& w, const float* scale, int l, int o, int c, int cols) {
  return w.v[l][(size_t)o * cols + c] * scale[l * kScaleStride + o];
}

// column of lin4's weight that multiplies accumulator row i of the spliced layer-4 input
// (rows 0..216 = h, 217..223 = PE[32..38], 224..255 = PE[0..31]; reference order is cat[h(217), PE(39)])
__device__ __forceinline__ int l4_col(int i) { return i < 217 ? i : (i < 224 ? 217 + 32 + (i - 217) : 217 + (i - 224)); }

// PE index carried by output row (tile, local) of the reverse layer-0 product (matches the splice layout)
__device__ __forceinline__ int rev0_pe(int tile, int local) {
  if (tile == 0) return local;             // PE[0..31]
  return local >= 25 ? 32 + (local - 25) : -1;  // tile 1: local rows 25..31 -> PE[32..38]
}

__global__ void pack_sdf_kernel(LayerPtrs w, const float* __restrict__ scale, float* __restrict__ out, int full) {
  const size_t total = (full ? kSdfFullF4 : kSdfFwdF4) * 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const float inv_sqrt2 = 0.70710678118654752f;
  const size_t base1 = 8 * (size_t)kChunk0F4 * 4;
  float val = 0.0f;
  if (idx < base1) {
    // ---- layer 0, tile t
    const int t = (int)(idx / (kChunk0F4 * 4));
    const int wi = (int)(idx % (kChunk0F4 * 4));
    if (wi < kHdrF4 * 4) {
      const int r = 4 * (wi / 256) + (wi & 3), lane = (wi & 255) >> 2;
      val = w.b[0][32 * t + rho(r) + 4 * (lane >> 5)];
    } else {
      const int wb = wi - kHdrF4 * 4;
      const int s = 4 * (wb / 256) + (wb & 3), lane = (wb & 255) >> 2;
      const int q = 2 * s + (lane >> 5);
      if (q < 39) val = weff(w, scale, 0, 32 * t + (lane & 31), q, 39);
    }
    out[idx] = val;
    return;
  }
  const size_t rel = idx - base1;
  const int j = (int)(rel / (kChunkF4 * 4));
  const int wi = (int)(rel % (kChunkF4 * 4));
  const bool hdr = wi < kHdrF4 * 4;
  const int wb = hdr ? wi : wi - kHdrF4 * 4;
  const int lane = (wb & 255) >> 2, half = lane >> 5, col32 = lane & 31;
  const int sr = 4 * (wb / 256) + (wb & 3);              // hdr: accumulator register r; body: k-step s
  const int crow = 32 * (sr / 16) + rho(sr % 16) + 4 * half;  // body: C-layout row addressed by k-step s

  // decode chunk j
  int kind, l, t;  // kind 0 FWD, 1 VEC, 2 FEAT, 3 REV, 4 REV0
  if (j < 55) {
    kind = 0;
    if (j < 16) { l = 1 + j / 8; t = j % 8; }
    else if (j < 23) { l = 3; t = j - 16; }
    else { l = 4 + (j - 23) / 8; t = (j - 23) % 8; }
  } else if (j == 55) { kind = 1; l = 8; t = 0; }
  else if (j < 64) { kind = 2; l = 8; t = j - 56; }
  else if (j < 120) { kind = 3; l = 7 - (j - 64) / 8; t = (j - 64) % 8; }
  else { kind = 4; l = 0; t = j - 120; }

  if (kind == 0 || kind == 2) {
    const int rows = sdf_rows(l);
    if (hdr) {
      const int o = (kind == 2 ? 1 : 0) + 32 * t + rho(sr) + 4 * half;
      if (o < rows) val = w.b[l][o];
    } else {
      const int o = (kind == 2 ? 1 : 0) + 32 * t + col32;
      if (o < rows) {
        if (l == 4) val = weff(w, scale, 4, o, l4_col(crow), 256) * inv_sqrt2;
        else val = weff(w, scale, l, o, crow, 256);
      }
    }
  } else if (kind == 1) {
    if (hdr) val = w.b[8][0];
    else val = weff(w, scale, 8, 0, crow, 256);
  } else if (kind == 3) {
    // reverse of layer l: out row = input feature i_out (C-layout row of the layer's input), k = output feature
    if (!hdr) {
      const int i_out = 32 * t + col32;
      const int k = crow;
      if (k < sdf_rows(l)) {
        if (l == 4) val = weff(w, scale, 4, k, l4_col(i_out), 256) * inv_sqrt2;
        else val = weff(w, scale, l, k, i_out, 256);
      }
    }
  } else {
    if (!hdr) {
      const int q = rev0_pe(t, col32);
      if (q >= 0) val = weff(w, scale, 0, crow, q, 39);
    }
  }
  out[idx] = val;
}

// radiance stream (mode 'idr', network.py:174-176: cat[points(3), PE1(view)(9), normals(3), feature(256)])
__global__ void pack_rgb_kernel(LayerPtrs w, const float* __restrict__ scale, float* __restrict__ out) {
  const size_t total = kRgbF4 * 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t base1 = 8 * (size_t)kRgbChunk0F4 * 4;
  float val = 0.0f;
  if (idx < base1) {
    const int t = (int)(idx / (kRgbChunk0F4 * 4));
    const int wi = (int)(idx % (kRgbChunk0F4 * 4));
    if (wi < kHdrF4 * 4) {
      const int r = 4 * (wi / 256) + (wi & 3), lane = (wi & 255) >> 2;
      val = w.b[0][32 * t + rho(r) + 4 * (lane >> 5)];
    } else {
      const int wb = wi - kHdrF4 * 4;
      const int s = 4 * (wb / 256) + (wb & 3), lane = (wb & 255) >> 2, half = lane >> 5;
      const int o = 32 * t + (lane & 31);
      if (s < 128) {
        const int crow = 32 * (s / 16) + rho(s % 16) + 4 * half;   // feature row
        val = weff(w, scale, 0, o, 15 + crow, 271);
      } else {
        const int e = rho(s - 128) + 4 * half;                     // extra row 0..15
        if (e < 15) val = weff(w, scale, 0, o, e, 271);
      }
    }
    out[idx] = val;
    return;
  }
  const size_t rel = idx - base1;
  const int j = (int)(rel / (kChunkF4 * 4));
  const int wi = (int)(rel % (kChunkF4 * 4));
  const bool hdr = wi < kHdrF4 * 4;
  const int wb = hdr ? wi : wi - kHdrF4 * 4;
  const int lane = (wb & 255) >> 2, half = lane >> 5, col32 = lane & 31;
  const int sr = 4 * (wb / 256) + (wb & 3);
  const int crow = 32 * (sr / 16) + rho(sr % 16) + 4 * half;
  const int l = j < 24 ? 1 + j / 8 : 4;
  const int t = j < 24 ? j % 8 : 0;
  const int rows = rgb_rows(l);
  if (hdr) {
    const int o = 32 * t + rho(sr) + 4 * half;
    if (o < rows) val = w.b[l][o];
  } else {
    const int o = 32 * t + col32;
    if (o < rows) val = weff(w, scale, l, o, crow, 256);
  }
  out[idx] = val;
}

}  // namespace mlp
}  // namespace svs

using namespace svs;
using namespace svs::mlp;

extern "C" {

size_t svs_sdf_stream_bytes(int full) { return (full ? kSdfFullF4 : kSdfFwdF4) * 16 ; }
size_t svs_rgb_stream_bytes(void) { return kRgbF4 * 16; }
size_t svs_pack_workspace_bytes(void) { return 9 * kScaleStride * sizeof(float); }

int svs_sdf_pack(const float* const* weight_v, const float* const* weight_g, const float* const* bias,
                 float* workspace, float* stream_out, int full, void* hip_stream) {
  if (!weight_v || !bias || !workspace || !stream_out) { set_error("svs_sdf_pack: null argument"); return SVS_EINVAL; }
  LayerPtrs w;
  for (int l = 0; l < 9; ++l) {
    w.v[l] = weight_v[l]; w.g[l] = weight_g ? weight_g[l] : nullptr; w.b[l] = bias[l];
    if (!w.v[l] || !w.b[l]) { set_error("svs_sdf_pack: null layer %d", l); return SVS_EINVAL; }
  }
  hipStream_t s = (hipStream_t)hip_stream;
  rownorm_kernel<<<(9 * kScaleStride + 3) / 4, 256, 0, s>>>(w, 9, 0, workspace);
  const size_t total = (full ? kSdfFullF4 : kSdfFwdF4) * 4;
  pack_sdf_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(w, workspace, stream_out, full);
  return check_launch("svs_sdf_pack");
}

int svs_rgb_pack(const float* const* weight_v, const float* const* weight_g, const float* const* bias,
                 float* workspace, float* stream_out, void* hip_stream) {
  if (!weight_v || !bias || !workspace || !stream_out) { set_error("svs_rgb_pack: null argument"); return SVS_EINVAL; }
  LayerPtrs w = {};
  for (int l = 0; l < 5; ++l) {
    w.v[l] = weight_v[l]; w.g[l] = weight_g ? weight_g[l] : nullptr; w.b[l] = bias[l];
    if (!w.v[l] || !w.b[l]) { set_error("svs_rgb_pack: null layer %d", l); return SVS_EINVAL; }
  }
  hipStream_t s = (hipStream_t)hip_stream;
  rownorm_kernel<<<(5 * kScaleStride + 3) / 4, 256, 0, s>>>(w, 5, 1, workspace);
  pack_rgb_kernel<<<(unsigned)((kRgbF4 * 4 + 255) / 256), 256, 0, s>>>(w, workspace, stream_out);
  return check_launch("svs_rgb_pack");
}

}  // extern "C"
