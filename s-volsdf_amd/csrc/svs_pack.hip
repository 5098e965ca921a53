// Weight packing for the fused MLP kernels: weight-norm materialisation (w = g * v / ||v||_row,
// volsdf/model/network.py:64-65) and the permutation of every layer into the order in which the MFMA
// k-loops of svs_mlp.hip / svs_mlp_bwd.hip consume it (chunk format: svs_mlp_layout.h).  Runs once per
// optimisation step.  Streams are described by a table of chunk descriptors, so forward, gradient-pass and
// training-backward streams share one pack kernel.
#include "svs_common.h"
#include "svs_mlp_layout.h"

#include <vector>

namespace svs {
namespace mlp {

struct LayerPtrs {
  const float* v[9];   // weight_v (or plain weight) [out][in] row-major
  const float* g[9];   // weight_g [out] or nullptr (no weight-norm)
  const float* b[9];   // bias [out]
};

constexpr int kScaleStride = 264;

// layer shapes; net: 0 = foreground networks, 1 = inverted-sphere background networks (network_bg.py:31-35)
__device__ __host__ constexpr int sdf_rows(int l, int net = 0) { return l == 3 ? (net ? 172 : 217) : (l == 8 ? 257 : 256); }
__device__ __host__ constexpr int sdf_cols(int l, int net = 0) { return l == 0 ? (net ? 84 : 39) : 256; }
__device__ __host__ constexpr int rgb_rows(int l, int net = 0) { return net ? (l == 0 ? 128 : 3) : (l == 4 ? 3 : 256); }
__device__ __host__ constexpr int rgb_cols(int l, int net = 0) { return net ? (l == 0 ? 283 : 128) : (l == 0 ? 271 : 256); }

// scale[l][o] = g[o] / ||v[o,:]||  (1 when the layer has no weight-norm); one wave per row
__global__ void rownorm_kernel(LayerPtrs w, int n_layers, int is_rgb, int net, float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int row_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int l = row_global / kScaleStride, o = row_global % kScaleStride;
  if (l >= n_layers) return;
  const int rows = is_rgb ? rgb_rows(l, net) : sdf_rows(l, net);
  const int cols = is_rgb ? rgb_cols(l, net) : sdf_cols(l, net);
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
__device__ __host__ inline int l4_col(int i, int net = 0) {
  if (net == 0) return i < 217 ? i : (i < 224 ? 217 + 32 + (i - 217) : 217 + (i - 224));
  // bg: rows 0..171 = h, 172..191 = PE[64..83], 192..223 = PE[32..63], 224..255 = PE[0..31]; reference order cat[h(172), PE(84)]
  return i < 172 ? i : (i < 192 ? 172 + 64 + (i - 172) : (i < 224 ? 172 + 32 + (i - 192) : 172 + (i - 224)));
}

// PE index carried by output row (tile, local) of the reverse layer-0 product (matches the splice layout)
__device__ __forceinline__ int rev0_pe(int tile, int local) {
  if (tile == 0) return local;                  // PE[0..31]
  return local >= 25 ? 32 + (local - 25) : -1;  // tile 1: local rows 25..31 -> PE[32..38]
}

struct ChunkDesc { int kind, layer, tile, off_f4; };   // off_f4: offset of the chunk in the stream, in float4
// The chunk table of a stream travels BY VALUE in the kernel arguments (<= 136 descriptors = 2176 bytes of the 4 KiB a
// launch may carry): the library allocates nothing and copies nothing on its own -- the boundary's "no allocation" rule
// holds for svs_pack_stream too (until round 3 its first call per stream kind did a hipMalloc + hipMemcpy of the table).
constexpr int kMaxChunks = 136;
struct ChunkTable { ChunkDesc d[kMaxChunks]; };

// Weight that multiplies input row `crow` (C-layout row of the layer input) in output row 32*t + col32 of a chunk.
// q0: PE index of the element (SDF layer 0); e: index into the 16 extra input rows (radiance layer 0) or -1.
__device__ __forceinline__ float body_value(const LayerPtrs& w, const float* scale, int kind, int l, int t, int col32,
                                            int crow, int q0, int e, int net) {
  const float inv_sqrt2 = 0.70710678118654752f;
  switch (kind) {
    case kSdfFwd0:
    case kSdfFwd0W:
      return q0 < 39 ? weff(w, scale, 0, 32 * t + col32, q0, 39) : 0.0f;
    case kBgFwd0:
      return q0 < kBgPeDim ? weff(w, scale, 0, 32 * t + col32, q0, kBgPeDim) : 0.0f;
    case kBgRgbFwd0: {
      // nerf mode (network.py:176): input = cat[PE4(view)(27), feature(256)]
      const int o = 32 * t + col32;
      if (o >= 128) return 0.0f;
      if (e < 0) return weff(w, scale, 0, o, 27 + crow, 283);
      return e < 27 ? weff(w, scale, 0, o, e, 283) : 0.0f;
    }
    case kBgRgbFwd1: {
      const int o = 32 * t + col32;
      return (o < 3 && crow < 128) ? weff(w, scale, 1, o, crow, 128) : 0.0f;
    }
    case kBgRgbRev0:
      return crow < 128 ? weff(w, scale, 0, crow, 27 + 32 * t + col32, 283) : 0.0f;
    case kSdfFwd:
    case kSdfFeat: {
      const int o = (kind == kSdfFeat ? 1 : 0) + 32 * t + col32;
      if (o >= sdf_rows(l, net)) return 0.0f;
      return l == 4 ? weff(w, scale, 4, o, l4_col(crow, net), 256) * inv_sqrt2 : weff(w, scale, l, o, crow, 256);
    }
    case kSdfVec:
      return weff(w, scale, 8, 0, crow, 256);
    case kSdfRev: {
      // reverse of layer l: out row = input feature i_out (C-layout row of the layer's input), k = output feature
      if (crow >= sdf_rows(l, net)) return 0.0f;
      const int i_out = 32 * t + col32;
      return l == 4 ? weff(w, scale, 4, crow, l4_col(i_out, net), 256) * inv_sqrt2 : weff(w, scale, l, crow, i_out, 256);
    }
    case kSdfRev0: {
      const int q = rev0_pe(t, col32);
      return q >= 0 ? weff(w, scale, 0, crow, q, 39) : 0.0f;
    }
    case kSdfFeatT:
      // h_bar_8 += W8[1:,:]^T f_bar : out row = h_8 feature (32t+col), k = feature-vector index
      return weff(w, scale, 8, 1 + crow, 32 * t + col32, 256);
    case kRgbFwd0: {
      const int o = 32 * t + col32;
      if (e < 0) return weff(w, scale, 0, o, 15 + crow, 271);
      return e < 15 ? weff(w, scale, 0, o, e, 271) : 0.0f;
    }
    case kRgbFwd: {
      const int o = 32 * t + col32;
      return o < rgb_rows(l) ? weff(w, scale, l, o, crow, 256) : 0.0f;
    }
    case kRgbRev:
      return weff(w, scale, l, crow, 32 * t + col32, 256);
    case kRgbRev0: {
      // out row = layer-0 input in kernel order: tiles 0..7 feature rows (param col 15+row), tile 8 the 16 extras
      const int pc = t < 8 ? 15 + 32 * t + col32 : (col32 < 15 ? col32 : -1);
      return pc >= 0 ? weff(w, scale, 0, crow, pc, 271) : 0.0f;
    }
    default:
      return 0.0f;
  }
}

// bias of output row o_local = rho(r) + 4*half of tile t (header register r)
__device__ __forceinline__ float header_value(const LayerPtrs& w, int kind, int l, int t, int o_local, int net) {
  switch (kind) {
    case kSdfFwd0: case kSdfFwd0W: case kBgFwd0: return w.b[0][32 * t + o_local];
    case kBgRgbFwd0: { const int o = 32 * t + o_local; return o < 128 ? w.b[0][o] : 0.0f; }
    case kBgRgbFwd1: { const int o = 32 * t + o_local; return o < 3 ? w.b[1][o] : 0.0f; }
    case kSdfFwd: { const int o = 32 * t + o_local; return o < sdf_rows(l, net) ? w.b[l][o] : 0.0f; }
    case kSdfFeat: { const int o = 1 + 32 * t + o_local; return o < sdf_rows(l) ? w.b[l][o] : 0.0f; }
    case kSdfVec: return w.b[8][0];
    case kRgbFwd0: return w.b[0][32 * t + o_local];
    case kRgbFwd: { const int o = 32 * t + o_local; return o < rgb_rows(l) ? w.b[l][o] : 0.0f; }
    default: return 0.0f;     // transposed (backward) chunks carry no bias
  }
}

// kPackParts workgroups per chunk (one per chunk left the launch latency-bound: 128 workgroups of 8 serial gather rounds,
// 34 us on the critical path of every optimiser step)
constexpr int kPackParts = 4;
constexpr int kPackStride = 256 * kPackParts;
__global__ __launch_bounds__(256) void pack_stream_kernel(LayerPtrs w, const float* __restrict__ scale,
                                                          const ChunkTable table, int fmt, int net,
                                                          float* __restrict__ out) {
  const ChunkDesc d = table.d[blockIdx.x / kPackParts];
  const int tid = (blockIdx.x % kPackParts) * 256 + threadIdx.x;
  const int kind = d.kind & 0xff;
  const bool nobias = (d.kind & kNoBias) != 0;
  const int l = d.layer, t = d.tile;
  float* dst = out + (size_t)d.off_f4 * 4;
  if (kind == kRgbW4T) {
    // [tile 8][lane 64][4 k-steps]: A = W4^T rows (input feature 32*tile + lane&31), k = rho(s) + 4*half < 3
    for (int wi = tid; wi < kW4TF4 * 4; wi += kPackStride) {
      const int tt = wi / 256, lane = (wi & 255) >> 2, s = wi & 3;
      const int k = rho(s) + 4 * (lane >> 5);
      dst[wi] = k < 3 ? weff(w, scale, 4, k, 32 * tt + (lane & 31), 256) : 0.0f;
    }
    return;
  }
  if (kind == kBgRgbW1T) {
    // [tile 4][lane 64][4 k-steps]: A = W1^T rows (hidden unit 32*tile + lane&31), k = rho(s) + 4*half < 3
    for (int wi = tid; wi < kW1TF4 * 4; wi += kPackStride) {
      const int tt = wi / 256, lane = (wi & 255) >> 2, s = wi & 3;
      const int k = rho(s) + 4 * (lane >> 5);
      dst[wi] = k < 3 ? weff(w, scale, 1, k, 32 * tt + (lane & 31), 128) : 0.0f;
    }
    return;
  }
  if (fmt == kFmtF16x2W) {
    // ---- the 16-point-wave encoding (svs_mlp_layout.h): header float4 u * 64 + lane = bias of rows 16 u + 4 g .. + 3
    for (int wi = tid; wi < kHdrF4 * 4; wi += kPackStride) {
      const int f4 = wi >> 2, c = wi & 3, u = f4 >> 6, lane = f4 & 63, g = lane >> 4;
      dst[wi] = (nobias || u > 1) ? 0.0f : header_value(w, kind, l, t, 16 * u + 4 * g + c, net);
    }
    float* bodyw = dst + kHdrF4 * 4;
    const int ks = kind == kSdfFwd0W ? 2 : 8;
    uint4* fragw = reinterpret_cast<uint4*>(bodyw);
    for (int f = tid; f < 2 * ks * 2 * 64; f += kPackStride) {
      const int lane = f & 63, piece = (f >> 6) & 1, s = (f >> 7) % ks, u = (f >> 7) / ks, g = lane >> 4, row = lane & 15;
      unsigned short ebits[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int crow = 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3);        // input feature (kSdfFwd0W: PE index q0)
        const float wv = body_value(w, scale, kind, l, t, 16 * u + row, crow, 32 * s + 8 * g + j, -1, net);
        const _Float16 hi = (_Float16)wv;
        const _Float16 mid = (_Float16)(wv - (float)hi);
        ebits[j] = __builtin_bit_cast(unsigned short, piece == 0 ? hi : mid);
      }
      uint4 v;
      v.x = ebits[0] | ((unsigned)ebits[1] << 16); v.y = ebits[2] | ((unsigned)ebits[3] << 16);
      v.z = ebits[4] | ((unsigned)ebits[5] << 16); v.w = ebits[6] | ((unsigned)ebits[7] << 16);
      fragw[f] = v;
    }
    return;
  }
  // ---- header: [r/4][lane][4] bias block in accumulator layout
  for (int wi = tid; wi < kHdrF4 * 4; wi += kPackStride) {
    const int lane = (wi & 255) >> 2, r = 4 * (wi / 256) + (wi & 3);
    dst[wi] = nobias ? 0.0f : header_value(w, kind, l, t, rho(r) + 4 * (lane >> 5), net);
  }
  float* body = dst + kHdrF4 * 4;
  const int body_f4 = chunk_f4(kind) - kHdrF4;
  if (fmt == kFmtF32 || kind == kSdfVec) {
    for (int wb = tid; wb < body_f4 * 4; wb += kPackStride) {
      const int lane = (wb & 255) >> 2, half = lane >> 5, col32 = lane & 31;
      const int sr = 4 * (wb / 256) + (wb & 3);                   // k-step: input rows 2*sr, 2*sr+1 in K order
      const int crow = 32 * (sr / 16) + rho(sr % 16) + 4 * half;  // C-layout row addressed by k-step sr
      const int e = sr >= 128 ? rho(sr - 128) + 4 * half : -1;
      body[wb] = body_value(w, scale, kind, l, t, col32, crow, 2 * sr + half, e, net);
    }
    return;
  }
  // ---- fp16x2: one 16-byte fragment per (k-step, piece, lane)
  uint4* frag = reinterpret_cast<uint4*>(body);
  for (int f = tid; f < body_f4; f += kPackStride) {
    const int s = f >> 7, piece = (f >> 6) & 1, lane = f & 63, half = lane >> 5, col32 = lane & 31;
    unsigned short ebits[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int crow = 32 * (s >> 1) + rho(8 * (s & 1) + j) + 4 * half;
      const int e = s >= 16 ? 16 * (s - 16) + rho(j) + 4 * half : -1;
      const float wv = body_value(w, scale, kind, l, t, col32, crow, 16 * s + 8 * half + j, e, net);
      const _Float16 hi = (_Float16)wv;
      const _Float16 mid = (_Float16)(wv - (float)hi);
      ebits[j] = __builtin_bit_cast(unsigned short, piece == 0 ? hi : mid);
    }
    uint4 v;
    v.x = ebits[0] | ((unsigned)ebits[1] << 16); v.y = ebits[2] | ((unsigned)ebits[3] << 16);
    v.z = ebits[4] | ((unsigned)ebits[5] << 16); v.w = ebits[6] | ((unsigned)ebits[7] << 16);
    frag[f] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------
// stream tables (host)
// ------------------------------------------------------------------------------------------------------------
struct StreamTable {
  std::vector<ChunkDesc> host;
  ChunkTable arg;                     // the same descriptors in the form the kernel takes them
  size_t total_f4 = 0;
  void add(int kind, int layer, int tile) {
    host.push_back(ChunkDesc{kind, layer, tile, (int)total_f4});
    total_f4 += chunk_f4(kind & 0xff);
  }
};

static void sdf_forward_part(StreamTable& t, int flags) {
  for (int i = 0; i < 8; ++i) t.add(kSdfFwd0 | flags, 0, i);
  for (int l = 1; l < 8; ++l)
    for (int i = 0; i < (l == 3 ? 7 : 8); ++i) t.add(kSdfFwd | flags, l, i);
}

static StreamTable& table_for(int which) {
  static StreamTable tabs[kNumStreams];
  StreamTable& t = tabs[which];
  if (!t.host.empty()) return t;
  switch (which) {
    case kStreamSdfFwd:
    case kStreamSdfFull:
      sdf_forward_part(t, 0);
      t.add(kSdfVec, 8, 0);
      if (which == kStreamSdfFull) {
        for (int i = 0; i < 8; ++i) t.add(kSdfFeat, 8, i);
        for (int l = 7; l >= 1; --l) for (int i = 0; i < 8; ++i) t.add(kSdfRev, l, i);
        for (int i = 0; i < 2; ++i) t.add(kSdfRev0, 0, i);
      }
      break;
    case kStreamSdfTrain:
      // pass A (second-order sweep): the forward trunk without biases; pass B: transposed feature head, then the
      // transposed trunk
      sdf_forward_part(t, kNoBias);
      for (int i = 0; i < 8; ++i) t.add(kSdfFeatT, 8, i);
      for (int l = 7; l >= 1; --l) for (int i = 0; i < 8; ++i) t.add(kSdfRev, l, i);
      break;
    case kStreamRgbFwd:
      for (int i = 0; i < 8; ++i) t.add(kRgbFwd0, 0, i);
      for (int l = 1; l < 4; ++l) for (int i = 0; i < 8; ++i) t.add(kRgbFwd, l, i);
      t.add(kRgbFwd, 4, 0);
      break;
    case kStreamRgbBwd:
      t.add(kRgbW4T, 4, 0);
      for (int l = 3; l >= 1; --l) for (int i = 0; i < 8; ++i) t.add(kRgbRev, l, i);
      for (int i = 0; i < 9; ++i) t.add(kRgbRev0, 0, i);
      break;
    case kStreamBgFwd:
      for (int i = 0; i < 8; ++i) t.add(kBgFwd0, 0, i);
      for (int l = 1; l < 8; ++l)
        for (int i = 0; i < (l == 3 ? 6 : 8); ++i) t.add(kSdfFwd, l, i);     // lin3: 172 rows = 6 tiles
      t.add(kSdfVec, 8, 0);
      for (int i = 0; i < 8; ++i) t.add(kSdfFeat, 8, i);
      break;
    case kStreamBgTrain:
      for (int i = 0; i < 8; ++i) t.add(kSdfFeatT, 8, i);
      for (int l = 7; l >= 1; --l) for (int i = 0; i < 8; ++i) t.add(kSdfRev, l, i);
      break;
    case kStreamBgRgbFwd:
      for (int i = 0; i < 4; ++i) t.add(kBgRgbFwd0, 0, i);
      t.add(kBgRgbFwd1, 1, 0);
      break;
    case kStreamBgRgbBwd:
      t.add(kBgRgbW1T, 1, 0);
      for (int i = 0; i < 8; ++i) t.add(kBgRgbRev0, 0, i);
      break;
    case kStreamSdfFwdW:
      for (int i = 0; i < 8; ++i) t.add(kSdfFwd0W, 0, i);
      for (int l = 1; l < 8; ++l)
        for (int i = 0; i < (l == 3 ? 7 : 8); ++i) t.add(kSdfFwd, l, i);
      t.add(kSdfFwd, 8, 0);     // the head: rows 0..31 of lin8 as one more tile (row 0 = sdf; the kernel runs its first sub-tile)
      break;
  }
  if (t.host.size() > (size_t)kMaxChunks) { t.host.clear(); t.total_f4 = 0; return t; }   // (cannot happen: static layouts)
  t.arg = ChunkTable{};
  for (size_t i = 0; i < t.host.size(); ++i) t.arg.d[i] = t.host[i];
  return t;
}

}  // namespace mlp
}  // namespace svs

using namespace svs;
using namespace svs::mlp;

extern "C" {

// which: 0 SDF forward, 1 SDF full (forward + feature head + gradient pass), 2 SDF training backward,
//        3 radiance forward, 4 radiance backward; background networks: 5 bg implicit forward,
//        6 bg implicit backward, 7 bg radiance forward, 8 bg radiance backward; 9 SDF forward for the 16-point-wave kernel
//        (svs_sdf_vals16, fp16x2 only).  precision: body encoding of the MFMA chunks, 0 float32, 1 fp16x2.
// Stream sizes do not depend on the precision.
size_t svs_stream_bytes(int which) {
  if (which < 0 || which >= kNumStreams) return 0;
  return table_for(which).total_f4 * 16;
}
size_t svs_pack_workspace_bytes(void) { return 9 * kScaleStride * sizeof(float); }

// weight_v / weight_g / bias: HOST arrays of 9 (SDF) or 5 (radiance) device pointers; weight_g may be NULL.
int svs_pack_stream(int which, int precision, const float* const* weight_v, const float* const* weight_g,
                    const float* const* bias, float* workspace, float* stream_out, void* hip_stream) {
  if (which < 0 || which >= kNumStreams || !weight_v || !bias || !workspace || !stream_out ||
      (precision != kFmtF32 && !is_h2(precision))) {
    set_error("svs_pack_stream: bad argument"); return SVS_EINVAL;
  }
  const bool is_rgb = which == kStreamRgbFwd || which == kStreamRgbBwd || which == kStreamBgRgbFwd || which == kStreamBgRgbBwd;
  const int net = stream_is_bg(which) ? 1 : 0;
  const int nl = is_rgb ? (net ? 2 : 5) : 9;
  if (is_h2(precision)) precision = kFmtF16x2;     // one body encoding for both fp16x2 block formats
  if (which == kStreamSdfFwdW) {
    if (precision != kFmtF16x2) { set_error("svs_pack_stream: stream 9 (16-point-wave SDF forward) is fp16x2 only"); return SVS_EINVAL; }
    precision = kFmtF16x2W;
  }
  LayerPtrs w = {};
  for (int l = 0; l < nl; ++l) {
    w.v[l] = weight_v[l]; w.g[l] = weight_g ? weight_g[l] : nullptr; w.b[l] = bias[l];
    if (!w.v[l] || !w.b[l]) { set_error("svs_pack_stream: null layer %d", l); return SVS_EINVAL; }
  }
  StreamTable& t = table_for(which);
  if (t.host.empty()) { set_error("svs_pack_stream: stream %d has no chunk table", which); return SVS_EINVAL; }
  hipStream_t s = (hipStream_t)hip_stream;
  rownorm_kernel<<<(nl * kScaleStride + 3) / 4, 256, 0, s>>>(w, nl, is_rgb ? 1 : 0, net, workspace);
  pack_stream_kernel<<<(unsigned)t.host.size() * kPackParts, 256, 0, s>>>(w, workspace, t.arg, precision, net, stream_out);
  return check_launch("svs_pack_stream");
}

}  // extern "C"
