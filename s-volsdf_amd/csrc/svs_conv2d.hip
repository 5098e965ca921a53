// 2-D convolutions of the FeatureNet pyramid for gfx950.
//
// Reference: models/CasMVSNet.py:24-55 (Conv2d block: conv + BatchNorm2d + ReLU) and :338-439 (FeatureNet, 'fpn':
// 3x3 / 5x5 / 1x1 kernels, strides 1 / 2, nearest x2 up-sampling added to a 1x1 lateral convolution).  BatchNorm in
// eval mode is folded into the weights by the caller.  The whole pyramid is ~5.6 GFLOP per 512x640 image in 13 launches
// and runs three times per scan (the stage loop caches the features), so it is a vector-unit job: direct float32
// convolution, one thread = PX adjacent output pixels x 8 output channels (PX * 8 accumulators), kernel size and stride
// compile-time.  Per input channel and kernel row a thread loads the (PX - 1) * stride + K input values its pixels
// share once into registers; a tap's 8 weights are the same for the whole wave, so they come through the SCALAR cache
// into SGPRs (s_load_dwordx8 from the packed layout [Cout / 8][Cin][k][k][8], which the caller prepares once per weight
// tensor) and feed the FMAs as scalar operands: no LDS, no vector memory traffic for weights.  (Reading them as
// broadcast ds_read_b128 made the kernel LDS-issue bound: 2 reads per 8 FMAs at PX = 1.)
// That direct kernel now serves the 1x1 lateral convolutions only (memory-bound); 3x3 / 5x5 run on the LDS-tiled kernel
// below.  svs_featurenet_fpn enqueues the whole pyramid (13 launches) from one call.
#include "svs_common.h"
#include <cstdlib>

namespace svs {
namespace conv2d {

constexpr int kCT = 8;
constexpr int kThreads = 128;

struct Args {
  const float* in;      // (Cin, H, W)
  const float* w;       // packed [ceil(Cout / 8)][Cin][k][k][8] (zero beyond Cout)
  const float* bias;    // [Cout] or nullptr
  const float* add;     // (Cout, Ho, Wo), or (Cout, Ho/2, Wo/2) with add_up2: added after the activation
  float* out;           // (Cout, Ho, Wo)
  int Cin, Cout, H, W, Ho, Wo, relu, add_up2;
  int groups_per_row;   // direct kernel: ceil(Wo / PX)
  int chunk;            // tiled kernel: input channels per LDS tile
};

template <int K, int S, int PX>
__global__ __launch_bounds__(kThreads) void conv2d_kernel(Args a) {
  constexpr int KK = K * K, PAD = K / 2, SPAN = (PX - 1) * S + K, kRowUnroll = K == 5 ? 1 : K;
  const int co0 = blockIdx.y * kCT;
  const float* __restrict__ wg = a.w + (size_t)blockIdx.y * a.Cin * KK * kCT;      // wave-uniform: scalar loads
  const int g = blockIdx.x * kThreads + threadIdx.x;
  if (g >= a.Ho * a.groups_per_row) return;
  const int yo = g / a.groups_per_row, xo = (g - yo * a.groups_per_row) * PX;
  const int y0 = yo * S - PAD, x0 = xo * S - PAD;
  // column offsets and validity of the input span (the same for every channel and row)
  int xoff[SPAN];
  bool xok[SPAN];
#pragma unroll
  for (int j = 0; j < SPAN; ++j) {
    const int ix = x0 + j;
    xok[j] = (unsigned)ix < (unsigned)a.W;
    xoff[j] = xok[j] ? ix : 0;
  }
  float acc[PX][kCT];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int c = 0; c < kCT; ++c) acc[p][c] = 0.0f;
  const size_t plane = (size_t)a.H * a.W;
  for (int ci = 0; ci < a.Cin; ++ci) {
    const float* ip = a.in + ci * plane;
    const float4* wp = reinterpret_cast<const float4*>(wg + ci * KK * kCT);
    // (K = 5: one kernel row at a time -- 25 taps x 8 weights would not fit the SGPR file)
#pragma unroll kRowUnroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = y0 + ky;
      const bool yok = (unsigned)iy < (unsigned)a.H;
      const float* row = ip + (size_t)(yok ? iy : 0) * a.W;
      float v[SPAN];
#pragma unroll
      for (int j = 0; j < SPAN; ++j) { const float t = row[xoff[j]]; v[j] = (yok && xok[j]) ? t : 0.0f; }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float4 w0 = wp[(ky * K + kx) * 2], w1 = wp[(ky * K + kx) * 2 + 1];
        const float wv[kCT] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int p = 0; p < PX; ++p)
#pragma unroll
          for (int c = 0; c < kCT; ++c) acc[p][c] = __builtin_fmaf(wv[c], v[p * S + kx], acc[p][c]);
      }
    }
  }
  const size_t oplane = (size_t)a.Ho * a.Wo;
  const bool whole = (xo + PX <= a.Wo);
#pragma unroll
  for (int c = 0; c < kCT; ++c) {
    const int co = co0 + c;
    if (co >= a.Cout) break;
    const float b = a.bias ? a.bias[co] : 0.0f;
    float r[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      r[p] = acc[p][c] + b;
      if (a.relu) r[p] = __builtin_fmaxf(r[p], 0.0f);
    }
    if (a.add) {
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        if (xo + p >= a.Wo) break;
        if (a.add_up2) r[p] += a.add[(size_t)co * (a.Ho / 2) * (a.Wo / 2) + (size_t)(yo >> 1) * (a.Wo / 2) + ((xo + p) >> 1)];   // nearest x2
        else r[p] += a.add[co * oplane + (size_t)yo * a.Wo + xo + p];
      }
    }
    float* op = a.out + co * oplane + (size_t)yo * a.Wo + xo;
    if constexpr (PX == 4) {
      if (whole && (a.Wo & 3) == 0) { *reinterpret_cast<float4*>(op) = make_float4(r[0], r[1], r[2], r[3]); continue; }
    }
    if constexpr (PX == 2) {
      if (whole && (a.Wo & 1) == 0) { *reinterpret_cast<float2*>(op) = make_float2(r[0], r[1]); continue; }
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) if (xo + p < a.Wo) op[p] = r[p];
  }
}

template <int K, int S, int PX>
int launch(Args a, hipStream_t s) {
  a.groups_per_row = (a.Wo + PX - 1) / PX;
  dim3 grid((a.Ho * a.groups_per_row + kThreads - 1) / kThreads, (a.Cout + kCT - 1) / kCT);
  conv2d_kernel<K, S, PX><<<grid, kThreads, 0, s>>>(a);
  return check_launch("svs_conv2d");
}

// ---- k = 3 / 5: output tile 64 x (4 PY) per block, input tile (+ halo) of a few channels at a time in LDS ----------
// Thread (column tx = lane, wave wv) owns PY vertically adjacent output pixels x 8 channels.  Per input channel and
// kernel column it reads the (PY - 1) S + K input values of its column from LDS (lanes side by side: conflict-free at
// stride 1, two-way at stride 2) for K * PY * 8 FMAs; global memory is read once per tile (coalesced, zero-filled halo).
constexpr int kTileW = 64;
constexpr int kTileThreads = 256;
constexpr int kTileLdsBudget = 40 * 1024;

template <int K, int S, int PY> struct TileDims {
  static constexpr int TH = 4 * PY, ROWS = (TH - 1) * S + K, COLS = (kTileW - 1) * S + K, PER_CH = ROWS * COLS;
};

template <int K, int S, int PY>
__global__ __launch_bounds__(kTileThreads) void conv2d_tile_kernel(Args a) {
  extern __shared__ float tile[];               // [channel of the chunk][ROWS][COLS]
  using D = TileDims<K, S, PY>;
  constexpr int KK = K * K, PAD = K / 2, SPAN = (PY - 1) * S + K;
  const int tx = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_x = (a.Wo + kTileW - 1) / kTileW;
  const int by = blockIdx.x / tiles_x, bx = blockIdx.x - by * tiles_x;
  const int xo0 = bx * kTileW, yo0 = by * D::TH;
  const int ix0 = xo0 * S - PAD, iy0 = yo0 * S - PAD;
  const int co0 = blockIdx.y * kCT;
  const float* __restrict__ wg = a.w + (size_t)blockIdx.y * a.Cin * KK * kCT;      // wave-uniform: scalar loads
  const size_t plane = (size_t)a.H * a.W;
  float acc[PY][kCT];
#pragma unroll
  for (int p = 0; p < PY; ++p)
#pragma unroll
    for (int c = 0; c < kCT; ++c) acc[p][c] = 0.0f;
  for (int c0 = 0; c0 < a.Cin; c0 += a.chunk) {
    const int nc = a.Cin - c0 < a.chunk ? a.Cin - c0 : a.chunk;
    __syncthreads();                            // the previous chunk has been consumed
    // one tile row (channel, input row) per wave and pass: the row index math is wave-uniform (scalar unit), a lane only adds
    // its column -- the element-indexed form spent as many instructions on divisions as the block spends on FMAs
    constexpr int kXP = (D::COLS + 63) / 64, kRB = 8;      // kRB rows of a wave in flight at a time
    const int n_rows = nc * D::ROWS;
    for (int rb = wv; rb < n_rows; rb += 4 * kRB) {
      float v[kRB][kXP];
#pragma unroll
      for (int q = 0; q < kRB; ++q) {
        const int r = rb + 4 * q;
        const int c = r / D::ROWS, ry = r - c * D::ROWS, iy = iy0 + ry;
        const bool yok = r < n_rows && (unsigned)iy < (unsigned)a.H;
        const float* __restrict__ src = a.in + (c0 + c) * plane + (size_t)(yok ? iy : 0) * a.W;
#pragma unroll
        for (int j = 0; j < kXP; ++j) {
          const int rx = 64 * j + tx, ix = ix0 + rx;
          v[q][j] = (yok && rx < D::COLS && (unsigned)ix < (unsigned)a.W) ? src[ix] : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < kRB; ++q) {
        const int r = rb + 4 * q;
        if (r < n_rows) {
#pragma unroll
          for (int j = 0; j < kXP; ++j) {
            const int rx = 64 * j + tx;
            if (64 * (j + 1) <= D::COLS || rx < D::COLS) tile[r * D::COLS + rx] = v[q][j];
          }
        }
      }
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
      const float* tp = tile + c * D::PER_CH + (wv * PY * S) * D::COLS + tx * S;
      const float4* wp = reinterpret_cast<const float4*>(wg + (size_t)(c0 + c) * KK * kCT);
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        float v[SPAN];
#pragma unroll
        for (int j = 0; j < SPAN; ++j) v[j] = tp[j * D::COLS + kx];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const float4 w0 = wp[(ky * K + kx) * 2], w1 = wp[(ky * K + kx) * 2 + 1];
          const float wv8[kCT] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int p = 0; p < PY; ++p)
#pragma unroll
            for (int cc = 0; cc < kCT; ++cc) acc[p][cc] = __builtin_fmaf(wv8[cc], v[p * S + ky], acc[p][cc]);
        }
      }
    }
  }
  const int xo = xo0 + tx;
  if (xo >= a.Wo) return;
  const size_t oplane = (size_t)a.Ho * a.Wo;
#pragma unroll
  for (int p = 0; p < PY; ++p) {
    const int yo = yo0 + wv * PY + p;
    if (yo >= a.Ho) break;
#pragma unroll
    for (int c = 0; c < kCT; ++c) {
      const int co = co0 + c;
      if (co >= a.Cout) break;
      float r = acc[p][c] + (a.bias ? a.bias[co] : 0.0f);
      if (a.relu) r = __builtin_fmaxf(r, 0.0f);
      if (a.add) {
        if (a.add_up2) r += a.add[(size_t)co * (a.Ho / 2) * (a.Wo / 2) + (size_t)(yo >> 1) * (a.Wo / 2) + (xo >> 1)];   // nearest x2
        else r += a.add[co * oplane + (size_t)yo * a.Wo + xo];
      }
      a.out[co * oplane + (size_t)yo * a.Wo + xo] = r;
    }
  }
}

template <int K, int S, int PY>
int launch_tile(Args a, hipStream_t s) {
  using D = TileDims<K, S, PY>;
  int chunk = kTileLdsBudget / (D::PER_CH * (int)sizeof(float));
  chunk = chunk < 1 ? 1 : (chunk > 8 ? 8 : chunk);
  a.chunk = chunk < a.Cin ? chunk : a.Cin;
  const size_t lds = (size_t)a.chunk * D::PER_CH * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_tile_kernel<K, S, PY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("svs_conv2d: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  dim3 grid(((a.Wo + kTileW - 1) / kTileW) * ((a.Ho + D::TH - 1) / D::TH), (a.Cout + kCT - 1) / kCT);
  conv2d_tile_kernel<K, S, PY><<<grid, kTileThreads, lds, s>>>(a);
  return check_launch("svs_conv2d");
}

// PY = 1 (64 x 4 tiles, ~13 KiB of LDS per block) everywhere: taller tiles lose more to the un-overlapped load phase of a
// block than they gain in halo re-reads -- several small blocks per CU overlap each other's loads and FMAs (measured on
// the 512 x 640 pyramid: 0.35 ms of kernel time at PY = 1, 0.37 at PY = 2 where it still gives 512 blocks, 0.58 at PY = 4).
// Also measured and dropped: the four waves of a block sharing one output row and splitting the input channels (four
// times the blocks for the quarter-resolution layers, partial sums added through LDS): 35 us instead of 28 for the
// 32 -> 32 layers, slower on every layer.
template <int K, int S>
int launch_tiled(const Args& a, hipStream_t s) { return launch_tile<K, S, 1>(a, s); }

// the largest PX that still gives the launch kMinWaves waves (256 CUs x 4 SIMDs; the accumulators give each wave plenty of
// independent work, so a few waves per SIMD hide the load latency)
constexpr int kMinWaves = 2048;
template <int K, int S>
int launch_px(const Args& a, hipStream_t s) {
  const long cgroups = (a.Cout + kCT - 1) / kCT;
  auto waves = [&](int px) { return (long)a.Ho * ((a.Wo + px - 1) / px) / 64 * cgroups; };
  if (waves(4) >= kMinWaves) return launch<K, S, 4>(a, s);
  if (waves(2) >= kMinWaves) return launch<K, S, 2>(a, s);
  return launch<K, S, 1>(a, s);
}

}  // namespace conv2d
}  // namespace svs

using namespace svs;
using namespace svs::conv2d;

namespace svs {
namespace conv2dmfma {      // csrc/svs_conv2d_mfma.hip
bool supported(int Cin, int Cout, int k, int stride);
int run(const float* in, const void* wfrag, const float* bias, float* out, int Cin, int Cout, int H, int W, int k, int stride,
        int relu, hipStream_t s);
int run_lateral(const float* lat_in, const float* lat_w, const float* lat_b, const float* lat_add, const void* wfrag,
                const float* bias, float* out, int Cout, int H, int W, int relu, hipStream_t s);
}  // namespace conv2dmfma
}  // namespace svs

namespace {

int run_conv(const float* in, const float* weight, const float* bias, const float* add, int add_upsample2, float* out,
             int Cin, int Cout, int H, int W, int k, int stride, int relu, hipStream_t s) {
  Args a;
  a.in = in; a.w = weight; a.bias = bias; a.add = add; a.out = out; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.relu = relu; a.add_up2 = add_upsample2; a.groups_per_row = 0; a.chunk = 0;
  const int pad = k / 2;
  a.Ho = (H + 2 * pad - k) / stride + 1; a.Wo = (W + 2 * pad - k) / stride + 1;
  if (add && add_upsample2 && ((a.Ho & 1) || (a.Wo & 1))) { set_error("svs_conv2d: x2 up-sampled addend needs even output sizes"); return SVS_ESHAPE; }
  if (k == 1) return stride == 1 ? launch_px<1, 1>(a, s) : launch_px<1, 2>(a, s);
  if (stride == 1) return k == 3 ? launch_tiled<3, 1>(a, s) : launch_tiled<5, 1>(a, s);
  return k == 3 ? launch_tiled<3, 2>(a, s) : launch_tiled<5, 2>(a, s);
}

// workspace layout of svs_featurenet_fpn (floats), P = H * W, b = base channels
struct FpnBuffers {
  size_t c0a, c0, c1a, c1b, c1, c2a, c2b, c2, f1, f2, total;
  FpnBuffers(int b, int H, int W) {
    const size_t P = (size_t)H * W;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    c0a = take(b * P); c0 = take(b * P);
    c1a = take(2 * b * P / 4); c1b = take(2 * b * P / 4); c1 = take(2 * b * P / 4);
    c2a = take(4 * b * P / 16); c2b = take(4 * b * P / 16); c2 = take(4 * b * P / 16);
    f1 = take(4 * b * P / 4); f2 = take(4 * b * P);
    total = o;
  }
};

}  // namespace

extern "C" {

int svs_conv2d(const float* in, const float* weight, const float* bias, const float* add, int add_upsample2, float* out,
               int Cin, int Cout, int H, int W, int k, int stride, int relu, void* hip_stream) {
  if (!in || !weight || !out || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (k != 1 && k != 3 && k != 5) || (stride != 1 && stride != 2)) {
    set_error("svs_conv2d: bad argument (k in {1,3,5}, stride in {1,2})"); return SVS_EINVAL;
  }
  return run_conv(in, weight, bias, add, add_upsample2, out, Cin, Cout, H, W, k, stride, relu, (hipStream_t)hip_stream);
}

size_t svs_featurenet_fpn_workspace_bytes(int base_channels, int H, int W) {
  if (base_channels < 1 || H < 4 || W < 4) return 0;
  return FpnBuffers(base_channels, H, W).total * sizeof(float);
}

int svs_featurenet_fpn2(const float* image, int H, int W, int base_channels, const float* const* weights, const float* const* biases,
                        const void* const* wfrags, float* workspace, float* stage1, float* stage2, float* stage3, void* hip_stream);

int svs_featurenet_fpn(const float* image, int H, int W, int base_channels, const float* const* weights, const float* const* biases,
                       float* workspace, float* stage1, float* stage2, float* stage3, void* hip_stream) {
  return svs_featurenet_fpn2(image, H, W, base_channels, weights, biases, nullptr, workspace, stage1, stage2, stage3, hip_stream);
}

// wfrags: null, or 13 entries: wfrags[i] != null -> layer i runs on the matrix cores (svs_conv2d_mfma: the caller packed its
// weights with svs_conv2d_mfma_pack and checked svs_conv2d_mfma_supported); null entries run on the float32 kernels
int svs_featurenet_fpn2(const float* image, int H, int W, int base_channels, const float* const* weights, const float* const* biases,
                        const void* const* wfrags, float* workspace, float* stage1, float* stage2, float* stage3, void* hip_stream) {
  if (!image || !weights || !biases || !workspace || !stage1 || !stage2 || !stage3 || base_channels < 1) {
    set_error("svs_featurenet_fpn: null argument"); return SVS_EINVAL;
  }
  if (H < 4 || W < 4 || (H & 3) || (W & 3)) { set_error("svs_featurenet_fpn: image height and width must be multiples of 4"); return SVS_ESHAPE; }
  for (int i = 0; i < 13; ++i) if (!weights[i]) { set_error("svs_featurenet_fpn: weights[%d] is null", i); return SVS_EINVAL; }
  const int b = base_channels, H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
  const FpnBuffers B(b, H, W);
  float* ws = workspace;
  hipStream_t s = (hipStream_t)hip_stream;
  int rc;
  // layer i on the matrix cores where the caller handed in its fragments (no addend there), else on the float32 kernels
  auto conv = [&](int i, const float* in, const float* add, int add_up2, float* out, int Cin, int Cout, int h, int w, int k, int stride,
                  int relu) -> int {
    if (wfrags && wfrags[i] && !add && svs::conv2dmfma::supported(Cin, Cout, k, stride))
      return svs::conv2dmfma::run(in, wfrags[i], biases[i], out, Cin, Cout, h, w, k, stride, relu, s);
    return run_conv(in, weights[i], biases[i], add, add_up2, out, Cin, Cout, h, w, k, stride, relu, s);
  };
#define SVS_FPN(...) if ((rc = conv(__VA_ARGS__)) != SVS_OK) return rc
  // bottom-up path (models/CasMVSNet.py:343-361): conv + folded BatchNorm + ReLU
  SVS_FPN(0, image, nullptr, 0, ws + B.c0a, 3, b, H, W, 3, 1, 1);
  SVS_FPN(1, ws + B.c0a, nullptr, 0, ws + B.c0, b, b, H, W, 3, 1, 1);
  SVS_FPN(2, ws + B.c0, nullptr, 0, ws + B.c1a, b, 2 * b, H, W, 5, 2, 1);
  SVS_FPN(3, ws + B.c1a, nullptr, 0, ws + B.c1b, 2 * b, 2 * b, H2, W2, 3, 1, 1);
  SVS_FPN(4, ws + B.c1b, nullptr, 0, ws + B.c1, 2 * b, 2 * b, H2, W2, 3, 1, 1);
  SVS_FPN(5, ws + B.c1, nullptr, 0, ws + B.c2a, 2 * b, 4 * b, H2, W2, 5, 2, 1);
  SVS_FPN(6, ws + B.c2a, nullptr, 0, ws + B.c2b, 4 * b, 4 * b, H4, W4, 3, 1, 1);
  SVS_FPN(7, ws + B.c2b, nullptr, 0, ws + B.c2, 4 * b, 4 * b, H4, W4, 3, 1, 1);
  // top-down path (:413-431): the nearest x2 up-sampling is an index shift in the lateral convolution's epilogue
  SVS_FPN(8, ws + B.c2, nullptr, 0, stage1, 4 * b, 4 * b, H4, W4, 1, 1, 0);               // out1
  SVS_FPN(9, ws + B.c1, ws + B.c2, 1, ws + B.f1, 2 * b, 4 * b, H2, W2, 1, 1, 0);          // inner1 + up(c2)
  SVS_FPN(10, ws + B.f1, nullptr, 0, stage2, 4 * b, 2 * b, H2, W2, 3, 1, 0);              // out2
  // inner2 + up(f1), then out3.  With out3 on the matrix cores (b = 8) the lateral step is formed while out3 converts its
  // input window and its 4b-channel full-resolution result never exists in memory (42 MB written and read back at 512 x 640:
  // 22.7 + 30.8 us as two launches, NOTES/r06.md); SVS_FPN_FUSE_LATERAL=0: two launches
  static const bool fuse_env = [] { const char* e = getenv("SVS_FPN_FUSE_LATERAL"); return !(e && e[0] == '0'); }();
  if (fuse_env && b == 8 && wfrags && wfrags[12] && svs::conv2dmfma::supported(4 * b, b, 3, 1)) {
    if ((rc = svs::conv2dmfma::run_lateral(ws + B.c0, weights[11], biases[11], ws + B.f1, wfrags[12], biases[12], stage3, b, H, W, 0, s)) != SVS_OK)
      return rc;
  } else {
    SVS_FPN(11, ws + B.c0, ws + B.f1, 1, ws + B.f2, b, 4 * b, H, W, 1, 1, 0);             // inner2 + up(f1)
    SVS_FPN(12, ws + B.f2, nullptr, 0, stage3, 4 * b, b, H, W, 3, 1, 0);                  // out3
  }
#undef SVS_FPN
  return SVS_OK;
}

}  // extern "C"
