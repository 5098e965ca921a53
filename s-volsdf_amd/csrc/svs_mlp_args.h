// Kernel argument blocks shared by the float32 (svs_mlp.hip) and fp16x2 (svs_mlp_h2.hip) MLP kernels.
#pragma once
#include "svs_mlp_dev.h"

namespace svs {
namespace mlp {

struct SdfOnlyArgs {
  PointSrc src;
  const f32x4* stream;   // packed forward stream
  float* sdf;            // (P)
  float sphere_radius;   // <= 0: no clamp (network.py:128)
  float sphere_scale;
  int clamp_n;           // the clamp applies to points [0, clamp_n)
  const int* gate;       // optional device flags: the workgroup is a no-op when gate[(its first point / gate_points) *
                         // gate_stride] == 0 (sampler rounds; one flag per convergence group of rays)
  int gate_points;       // points per gate group (a multiple of the 128 points of a workgroup)
  int gate_stride;       // ints between the flags of consecutive groups
};

struct SdfFullArgs {
  PointSrc src;
  const f32x4* stream;   // full stream
  float* sdf;            // (P)
  float* grad;           // (P,3)
  float* feat_tiles;     // [wave tiles][128*64] or nullptr
  float* hbuf;           // [wave tiles][8][128*64] activations h_1..h_8
  float* gbuf;           // optional [wave tiles][8][128*64]: ghat_l = g(h_{l+1}) * softplus'(a_l) of the gradient pass,
                         // l = 0..7 (training; rows >= 217 of block 3 are not meaningful)
  unsigned char* clamp_mask;  // optional (P): 1 where the sphere term of the clamp is active (training)
  float sphere_radius;   // > 0: min(sdf, scale*(R-|x|)) inside the differentiated graph (network.py:110-112)
  float sphere_scale;
  int clamp_n;           // ... for points [0, clamp_n); the rest differentiate the raw output (network.py:90-103)
};

struct RgbArgs {
  PointSrc src;            // sample positions (same source as the SDF kernel)
  const float* normals;    // (P,3) = d sdf / dx, not normalised (network.py:234)
  const float* view;       // view directions, (R,3) if view_S > 0 (one per ray) else (P,3)
  int view_S;
  const float* feat_tiles; // [wave tiles][128*64]
  const f32x4* stream;
  float* rgb;              // (P,3)
  float* rbuf;             // optional [wave tiles][4*8192 + 1024]: r_1..r_4 (post-ReLU) and the 16 extra input rows (training)
};
constexpr int kRbufF = 4 * kBlockF + 1024;

constexpr int kRgbBufF4 = kRgbChunk0F4;   // LDS buffer size for the radiance kernel (38 KiB)

typedef StreamT<kRgbBufF4> RgbStream;

// fp16x2 launches (svs_mlp_h2.hip)
int launch_sdf_only_h2(const SdfOnlyArgs& a, hipStream_t s);
int launch_sdf_only_kp(const SdfOnlyArgs& a, hipStream_t s);       // K-split pairs, two waves per SIMD (svs_mlp_h2p.hip)
int launch_sdf_full_h2(const SdfFullArgs& a, bool grad_pair, hipStream_t s);   // grad_pair: ghat blocks as both pieces
int launch_rgb_h2(const RgbArgs& a, bool grad_pair, hipStream_t s);           // grad_pair: r_l stored with both pieces

}  // namespace mlp
}  // namespace svs
