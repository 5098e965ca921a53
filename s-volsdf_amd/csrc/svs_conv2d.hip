// 2-D convolutions of the FeatureNet pyramid for gfx950.
//
// Reference: models/CasMVSNet.py:24-55 (Conv2d block: conv + BatchNorm2d + ReLU) and :338-439 (FeatureNet, 'fpn':
// 3x3 / 5x5 / 1x1 kernels, strides 1 / 2, nearest x2 up-sampling added to a 1x1 lateral convolution).  BatchNorm in
// eval mode is folded into the weights by the caller.  The whole pyramid is ~5 GFLOP per 512x640 image and runs three
// times per scan (the stage loop caches the features): direct float32 convolution on the vector units, one thread
// = one output pixel x 8 output channels, the 8-channel weight slice in LDS, input rows shared through L1.
#include "svs_common.h"

namespace svs {
namespace conv2d {

constexpr int kCT = 8;
constexpr int kMaxW = 64 * 25 * kCT;          // Cin * k * k * 8 floats of LDS (51 KiB) at Cin = 64, k = 5

struct Args {
  const float* in;      // (Cin, H, W)
  const float* w;       // [Cout][Cin][k][k]
  const float* bias;    // [Cout] or nullptr
  const float* add;     // (Cout, Ho, Wo), or (Cout, Ho/2, Wo/2) with add_up2: added after the activation
  float* out;           // (Cout, Ho, Wo)
  int Cin, Cout, H, W, Ho, Wo, k, stride, pad, relu, add_up2;
};

__global__ __launch_bounds__(256) void conv2d_kernel(Args a) {
  extern __shared__ float wl[];                 // [ci][ky][kx][c]
  const int co0 = blockIdx.y * kCT;
  const int kk = a.k * a.k;
  for (int i = threadIdx.x; i < a.Cin * kk * kCT; i += 256) {
    const int c = i % kCT, r = i / kCT;         // r = ci * kk + tap
    wl[i] = (co0 + c < a.Cout) ? a.w[(size_t)(co0 + c) * a.Cin * kk + r] : 0.0f;
  }
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.Ho * a.Wo) return;
  const int yo = p / a.Wo, xo = p - yo * a.Wo;
  const int y0 = yo * a.stride - a.pad, x0 = xo * a.stride - a.pad;
  float acc[kCT];
#pragma unroll
  for (int c = 0; c < kCT; ++c) acc[c] = 0.0f;
  const size_t plane = (size_t)a.H * a.W;
  for (int ci = 0; ci < a.Cin; ++ci) {
    const float* ip = a.in + ci * plane;
    const float* wp = wl + ci * kk * kCT;
    for (int ky = 0; ky < a.k; ++ky) {
      const int iy = y0 + ky;
      if ((unsigned)iy >= (unsigned)a.H) continue;
      for (int kx = 0; kx < a.k; ++kx) {
        const int ix = x0 + kx;
        if ((unsigned)ix >= (unsigned)a.W) continue;
        const float v = ip[(size_t)iy * a.W + ix];
        const float* wv = wp + (ky * a.k + kx) * kCT;
#pragma unroll
        for (int c = 0; c < kCT; ++c) acc[c] = __builtin_fmaf(wv[c], v, acc[c]);
      }
    }
  }
  const size_t oplane = (size_t)a.Ho * a.Wo;
#pragma unroll
  for (int c = 0; c < kCT; ++c) {
    const int co = co0 + c;
    if (co >= a.Cout) break;
    float r = acc[c] + (a.bias ? a.bias[co] : 0.0f);
    if (a.relu) r = __builtin_fmaxf(r, 0.0f);
    if (a.add) {
      if (a.add_up2) r += a.add[(size_t)co * (a.Ho / 2) * (a.Wo / 2) + (size_t)(yo >> 1) * (a.Wo / 2) + (xo >> 1)];   // nearest x2
      else r += a.add[co * oplane + p];
    }
    a.out[co * oplane + p] = r;
  }
}

}  // namespace conv2d
}  // namespace svs

using namespace svs;
using namespace svs::conv2d;

extern "C" {

int svs_conv2d(const float* in, const float* weight, const float* bias, const float* add, int add_upsample2, float* out,
               int Cin, int Cout, int H, int W, int k, int stride, int relu, void* hip_stream) {
  if (!in || !weight || !out || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (k != 1 && k != 3 && k != 5) || (stride != 1 && stride != 2)) {
    set_error("svs_conv2d: bad argument (k in {1,3,5}, stride in {1,2})"); return SVS_EINVAL;
  }
  if (Cin * k * k * kCT > kMaxW) { set_error("svs_conv2d: Cin * k * k too large for the LDS weight slice"); return SVS_ESHAPE; }
  Args a;
  a.in = in; a.w = weight; a.bias = bias; a.add = add; a.out = out; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.k = k; a.stride = stride; a.pad = k / 2; a.relu = relu; a.add_up2 = add_upsample2;
  a.Ho = (H + 2 * a.pad - k) / stride + 1; a.Wo = (W + 2 * a.pad - k) / stride + 1;
  if (add && add_upsample2 && ((a.Ho & 1) || (a.Wo & 1))) { set_error("svs_conv2d: x2 up-sampled addend needs even output sizes"); return SVS_ESHAPE; }
  const size_t lds = (size_t)Cin * k * k * kCT * sizeof(float);
  hipStream_t s = (hipStream_t)hip_stream;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("svs_conv2d: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  dim3 grid((a.Ho * a.Wo + 255) / 256, (Cout + kCT - 1) / kCT);
  conv2d_kernel<<<grid, 256, lds, s>>>(a);
  return check_launch("svs_conv2d");
}

}  // extern "C"
