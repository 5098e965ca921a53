// Kernel argument blocks shared by the float32 (svs_mlp_bwd.hip) and fp16x2 (svs_mlp_bwd_h2.hip) backward kernels.
#pragma once
#include "svs_mlp_dev.h"

namespace svs {
namespace mlp {

struct RgbBwdArgs {
  int P;
  const float* d_rgb;      // (P,3) d loss / d rgb (after the sigmoid)
  const float* rgb;        // (P,3) forward output
  const float* rbuf;       // forward activations [wave tiles][kRbufF]
  const f32x4* stream;     // radiance backward stream
  float* zbuf;             // out [wave tiles][5][kBlockF]: zbar_0..zbar_3, zbar_4 (first tile only; rest stays zero)
  float* feat_bar;         // out [wave tiles][kBlockF]: d loss / d feature vector
  float* d_normals;        // out (P,3): d loss / d normals (the rendering network's normal input)
  float* absmax;           // fp16x2 only, [3]: [1] = max |zbar|, [2] = max |feat_bar| (atomic max; caller zeroes)
};
constexpr int kRbufFb = 4 * kBlockF + 1024;

struct SdfBwdAArgs {
  PointSrc src;
  const float* d_grad;        // (P,3) nbar = d loss / d (d sdf/dx)
  const unsigned char* clamp_mask;  // (P) or nullptr: clamped points contribute no nbar
  const float* hbuf;          // [wave tiles][8][kBlockF] forward activations
  const float* gbuf;          // [wave tiles][8][kBlockF] ghat_l = g(h_{l+1}) * s'(a_l) (svs_sdf_outputs); float32 kernels only
  const f32x4* stream;        // SDF training stream (pass A part at offset 0)
  float* ubuf;                // out [wave tiles][9][kBlockF]: block 0 = u_0 (PE order, first 2 tiles), blocks 1..8 = u_1..u_8
  float* a2buf;               // out [wave tiles][8][kBlockF]; float32 kernels only (fp16x2: pass B re-forms a2 from ubuf and gbuf)
  float* pebuf;               // out [wave tiles][kBlockF]: h_0 = PE(x) in PE order (first 2 tiles), B operand of dW_0
  // fp16x2 only:
  float* absmax;              // [3]: [0] = max |u| (atomic max; caller zeroes)
};

struct SdfBwdBArgs {
  int P;
  const float* d_sdf;         // (P) sbar, or nullptr
  const unsigned char* clamp_mask;
  const float* feat_bar;      // [wave tiles][kBlockF] fbar (or nullptr: zero)
  int n_feat_tiles;           // wave tiles that have a feat_bar block (ray samples); later tiles (eikonal points) have none
  const float* hbuf; const float* gbuf;
  const float* a2buf;         // float32 kernels: the second-order blocks pass A stored
  const float* ubuf;          // fp16x2 foreground network: pass A's u blocks WITH their records (pass B re-forms a2 from them, gbuf
                              // and gbuf's records)
  const f32x4* stream;        // SDF training stream, pass B part
  float* abuf;                // out [wave tiles][8][kBlockF] abar_0..abar_7
  float* sbar_out;            // out (padded P): the effective sbar (clamp applied), for the lin8 row-0 gradient; may be null
  // fp16x2 only:
  float* absmax;              // [3]: [0] = max |abar| (atomic max)
  // the block holding ghat_7 = W8[0,:] s'(a_7) of a tile is w0 + tile * w0_stride (fg: gbuf block 7, stride 8 blocks)
  const float* w0; size_t w0_stride;
  // experiment (SVS_BWD_B_REVERSE=1): the workgroups walk the point tiles in DESCENDING order -- pass A, which ran just before,
  // wrote the highest tiles last (they may still sit in the 256 MB Infinity Cache), and the weight-gradient GEMM that follows
  // starts at tile 0, which this sweep then finishes last
  int reverse = 0;
};
// gp: the launch's scaled-block format (svs_blocks_h2.h): both fp16 pieces (true) or the hi piece only
int launch_bg_bwd_b_h2(const SdfBwdBArgs& a, bool gp, hipStream_t s);   // background implicit network: no a2, bg splice rows

int launch_rgb_bwd_h2(const RgbBwdArgs& a, bool gp, hipStream_t s);
int launch_lin8_row0_h2(const float* hbuf, const float* ubuf, const float* sbar, int n_points, int n_tiles_pad, float* out257,
                        bool gp, hipStream_t s);
int launch_sdf_bwd_a_h2(const SdfBwdAArgs& a, bool gp, hipStream_t s);
int launch_sdf_bwd_b_h2(const SdfBwdBArgs& a, bool gp, hipStream_t s);

}  // namespace mlp
}  // namespace svs
