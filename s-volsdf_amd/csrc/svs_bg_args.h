// Kernel argument blocks of the background networks' kernels, shared by the fp16x2 (svs_bg_h2.hip) and the float32
// (svs_bg_f32.hip) forms.
#pragma once
#include "svs_mlp_h2_trunk.h"
#include "svs_mlp_bwd_args.h"

namespace svs {
namespace mlp {

struct BgSdfArgs {
  const float* pts;      // (P,4) inverted-sphere points (unit direction, 1/r)
  int P;
  const f32x4* stream;   // kStreamBgFwd
  float* out0;           // (P) raw output[:, 0] (the density is its absolute value, AbsDensity)
  float* feat_tiles;     // [wave tiles][kBlockF]
  float* hbuf;           // training: [wave tiles][8][kBlockF] h_1..h_8, else nullptr
  float* ghat7;          // training: [wave tiles][kBlockF] W8[0,:] * softplus'(a_7) (pass B's seed), else nullptr
  float* pebuf;          // training: [wave tiles][kBlockF] the 84 PE inputs in PE order (first 3 tiles), B operand of dW_0
};

struct BgRgbArgs {
  int P;
  const float* view;       // view directions: (R,3) if view_S > 0 (one per ray) else (P,3)
  int view_S;
  const float* feat_tiles; // [wave tiles][kBlockF]
  const f32x4* stream;     // kStreamBgRgbFwd
  float* rgb;              // (P,3)
  float* rbuf;             // training: [wave tiles][kBgRbufF]: r_1 (post-ReLU, tiles 0..3) and the 32 view-PE rows, else nullptr
};
constexpr int kBgRbufF = kBlockF + 1024;
constexpr int kBgRgbBufF4 = kBgRgbChunk0F4;

typedef StreamT<kBgRgbBufF4> BgRgbStream;

struct BgRgbBwdArgs {
  int P;
  const float* d_rgb;      // (P,3)
  const float* rgb;        // (P,3) forward output
  const float* rbuf;       // forward activations [wave tiles][kBgRbufF]
  const f32x4* stream;     // kStreamBgRgbBwd
  float* zbuf;             // out [wave tiles][2][kBlockF]: zbar_0 (tiles 0..3), zbar_1 (rows 0..2 of tile 0); the other
                           // tiles stay zero (ZERO-INITIALISED by the caller once)
  float* feat_bar;         // out [wave tiles][kBlockF]
  float* absmax;           // [3]: [1] = max |zbar|, [2] = max |feat_bar|
};

// float32-MFMA forms (svs_bg_f32.hip; buffers of the same sizes and arrangement, every block in the float32 layout of svs_mlp_dev.h)
int launch_bg_sdf_f32(const BgSdfArgs& a, hipStream_t s);
int launch_bg_rgb_f32(const BgRgbArgs& a, hipStream_t s);
int launch_bg_rgb_bwd_f32(const BgRgbBwdArgs& a, hipStream_t s);
int launch_bg_bwd_b_f32(const SdfBwdBArgs& a, hipStream_t s);

}  // namespace mlp
}  // namespace svs
