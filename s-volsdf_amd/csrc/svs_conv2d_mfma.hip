// FeatureNet's 3x3 / 5x5 convolutions on the fp16 matrix cores (models/CasMVSNet.py:24-55,338-439).
//
// The pyramid's layers with 8 / 16 / 32 input channels carry 5 of its 5.6 GFLOP; on the float32 vector ALUs
// (csrc/svs_conv2d.hip) they are chains of latencies (17-39 us per launch at 512 x 640).  Here they are implicit GEMMs on
// v_mfma_f32_16x16x32_f16 with two-piece fp16 operands (fp16x2: hi * hi + hi * mid + mid * hi, float32 accumulation -- the
// float32 accuracy class of the MLP kernels, 2e-7 relative per layer), the 2-D sibling of csrc/svs_conv_mfma.hip:
//   D[cout][pixel] = sum_k W[cout][k] * P[k][pixel],   k = tap * Cin + cin  (cin fastest),  M = 16 output channels per tile
//   (MT = 1 or 2 tiles: Cout <= 32), N = 16 consecutive output x, K = 32.
// A workgroup (4 waves) owns 4 (y) x 32 (x) output windows and walks over its share of the windows (the weights -- A fragments
// of ALL k-steps -- are read once per wave and stay in registers); per window it converts the input halo (float32, channel
// first in global memory) into channel-last fp16 hi / mid pieces in LDS, then every wave computes one output row: 2 N tiles
// x MT M tiles x KS k-steps x 3 MFMAs, the B fragments (16-byte channel vectors of one input pixel) read from LDS one k-step
// ahead.  Pixel pitch in LDS: an odd multiple of 16 bytes (conflict-free 16-byte reads at pixel stride 1; stride 2: two-way).
#include "svs_common.h"

namespace svs {
namespace conv2dmfma {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4, kTX = 32;            // a workgroup's output window: RW * 4 rows (RW rows per wave) x 32 columns

struct Args {
  const float* in;      // (Cin, H, W)
  const uint4* wfrag;   // [MT][KS][2 pieces][64 lanes] 16-byte A fragments (svs_conv2d_mfma_pack)
  const float* bias;    // [Cout] or nullptr
  float* out;           // (Cout, Ho, Wo)
  int Cout, H, W, Ho, Wo, relu;
  int tiles_x, tiles;   // windows per output row of windows, windows in all
  // LAT instances: `in` is not read; the layer's input is formed while the halo is converted (the FPN's lateral step, models/
  // CasMVSNet.py:425-431: intra = inner(conv_k) + nearest-up2(intra_coarser)) and never exists in memory:
  //   input[c][y][x] = sum_ci lat_w[c][ci] * lat_in[ci][y][x] + lat_b[c] + lat_add[c][y / 2][x / 2]
  const float* lat_in;  // (8, H, W)
  const float* lat_w;   // the 1x1 weights in svs_conv2d's packed layout [Cin / 8][8][1][1][8]: W[c][ci] = lat_w[(c / 8) * 64 + ci * 8 + c % 8]
  const float* lat_b;   // [Cin] or nullptr
  const float* lat_add; // (Cin, H / 2, W / 2)
};

template <int CIN, int K, int S> constexpr int k_steps() { return (K * K * CIN + 31) / 32; }
template <int CIN> constexpr int pitch() { return CIN == 8 ? 16 : (CIN == 16 ? 48 : 80); }
template <int K, int S, int RW> constexpr int halo_y() { return (kWaves * RW - 1) * S + K; }
template <int K, int S> constexpr int halo_x() { return (kTX - 1) * S + K; }
template <int CIN, int K, int S, int RW> constexpr int lds_bytes() {
  return 2 * halo_y<K, S, RW>() * halo_x<K, S>() * pitch<CIN>() + (CIN * 8 + CIN) * 4;      // + the lateral weights of LAT instances
}

// RW: output rows per wave (2: an 8 x 32 window -- 1.33 instead of 1.6 input pixels converted per output pixel at 3x3 --
// for the layers at full image resolution, where the conversion of the halo is most of the kernel)
template <int CIN, int K, int S, int MT, int RW, bool LAT = false>
__global__ __launch_bounds__(256, 1) void conv2d_mfma_kernel(Args a) {
  static_assert(!LAT || CIN == 32, "the fused lateral step feeds a 32-channel layer from 8 channels");
  constexpr int KS = k_steps<CIN, K, S>();
  constexpr int PITCH = pitch<CIN>();
  constexpr int kTY = kWaves * RW;
  constexpr int HY = halo_y<K, S, RW>(), HX = halo_x<K, S>(), PAD = K / 2;
  constexpr int PIECE = HY * HX * PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int vox = lane & 15, kg = lane >> 4;
  const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;

  // ---- weights: all A fragments of this lane, for the whole kernel
  f16x8 wh[MT][KS], wm[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      wh[mt][s] = __builtin_bit_cast(f16x8, a.wfrag[((mt * KS + s) * 2) * 64 + lane]);
      wm[mt][s] = __builtin_bit_cast(f16x8, a.wfrag[((mt * KS + s) * 2 + 1) * 64 + lane]);
    }
  // ---- LAT: the lateral step's weights [c][ci] and biases behind the window in LDS (read back as broadcasts: held in registers
  // they cost the kernel two thirds of its occupancy)
  float* latw = reinterpret_cast<float*>(smem + 2 * PIECE);
  if (LAT) {
    for (int i = tid; i < CIN * 8; i += 256) { const int c = i >> 3, ci = i & 7; latw[i] = a.lat_w[(c >> 3) * 64 + ci * 8 + (c & 7)]; }
    for (int i = tid; i < CIN; i += 256) latw[CIN * 8 + i] = a.lat_b ? a.lat_b[i] : 0.0f;
    __syncthreads();
  }
  // ---- the B-fragment address of k-step s inside the halo window: (tap, channel group) of this lane's 8 elements
  int boff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int kk = 32 * s + 8 * kg;
    int tap = kk / CIN;
    const int ci0 = kk % CIN;
    if (tap > K * K - 1) tap = K * K - 1;        // padded k: the weights are zero there
    const int kh = tap / K, kw = tap % K;
    boff[s] = ((wave * RW * S + kh) * HX + (vox * S + kw)) * PITCH + 2 * ci0;
  }

  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int xo0 = tx * kTX, yo0 = ty * kTY;
    const int xi0 = xo0 * S - PAD, yi0 = yo0 * S - PAD;
    // ---- the input halo window -> channel-last fp16 hi / mid pieces (zero outside the image)
#pragma unroll 1
    for (int p = tid; p < HY * HX; p += 256) {
      const int ly = p / HX, lx = p - ly * HX;
      const int gy = yi0 + ly, gx = xi0 + lx;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      unsigned char* ph = smem + p * PITCH;
      auto put = [&](int c8, const float (&v)[8]) {
        f16x8 h, m;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const _Float16 hh = (_Float16)v[j];
          h[j] = hh;
          m[j] = (_Float16)(v[j] - (float)hh);
        }
        *reinterpret_cast<f16x8*>(ph + 16 * c8) = h;
        *reinterpret_cast<f16x8*>(ph + PIECE + 16 * c8) = m;
      };
      if (LAT) {
        // the same operations in the same order as the float32 1x1 kernel it replaces (svs_conv2d.hip: fma chain over the
        // input channels from 0, + bias, + addend): the values are the materialised tensor's bit for bit.  Eight output
        // channels at a time (their addends requested together), so that a pixel's 40 loads are not all live at once
        const float* g = a.lat_in + (ok ? (size_t)gy * a.W + gx : 0);
        const float* ad = a.lat_add + (ok ? (size_t)(gy >> 1) * (a.W >> 1) + (gx >> 1) : 0);
        const size_t HW4 = (size_t)(a.H >> 1) * (a.W >> 1);
        float c8[8];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) c8[ci] = ok ? g[(size_t)ci * HW] : 0.0f;
#pragma unroll 1
        for (int q = 0; q < CIN / 8; ++q) {
          float add8[8], v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) add8[j] = ok ? ad[(size_t)(8 * q + j) * HW4] : 0.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = 8 * q + j;
            const f32x4v w0 = *reinterpret_cast<const f32x4v*>(latw + 8 * c), w1 = *reinterpret_cast<const f32x4v*>(latw + 8 * c + 4);
            float acc = 0.0f;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) acc = __builtin_fmaf(w0[ci], c8[ci], acc);
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) acc = __builtin_fmaf(w1[ci], c8[4 + ci], acc);
            float r = acc + latw[CIN * 8 + c];
            r += add8[j];
            v[j] = ok ? r : 0.0f;                    // zero padding outside the image
          }
          put(q, v);
        }
      } else {
        const float* g = a.in + (ok ? (size_t)gy * a.W + gx : 0);
        float v[CIN];
#pragma unroll
        for (int c = 0; c < CIN; ++c) v[c] = ok ? g[(size_t)c * HW] : 0.0f;
#pragma unroll
        for (int c8 = 0; c8 < CIN / 8; ++c8) {
          const float w8[8] = {v[8 * c8], v[8 * c8 + 1], v[8 * c8 + 2], v[8 * c8 + 3], v[8 * c8 + 4], v[8 * c8 + 5], v[8 * c8 + 6], v[8 * c8 + 7]};
          put(c8, w8);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
    const int rowb = rw * S * HX * PITCH;        // this wave's row rw inside the window
    f32x4v acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4v)(0.0f); acc[mt][1] = (f32x4v)(0.0f); }
    // two-stage pipeline: the fragments of k-step s + 1 are read behind the first MFMA of k-step s
    f16x8 bh[2], bm[2];
    {
      const unsigned char* p = smem + rowb + boff[0];
#pragma unroll
      for (int xt = 0; xt < 2; ++xt) {
        bh[xt] = *reinterpret_cast<const f16x8*>(p + xt * 16 * S * PITCH);
        bm[xt] = *reinterpret_cast<const f16x8*>(p + PIECE + xt * 16 * S * PITCH);
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f16x8 nh[2], nm[2];
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[0][s], bh[0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < KS) {
        const unsigned char* p = smem + rowb + boff[s + 1];
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
          nh[xt] = *reinterpret_cast<const f16x8*>(p + xt * 16 * S * PITCH);
          nm[xt] = *reinterpret_cast<const f16x8*>(p + PIECE + xt * 16 * S * PITCH);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[0][s], bh[1], acc[0][1], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
          if (mt > 0) acc[mt][xt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[mt][s], bh[xt], acc[mt][xt], 0, 0, 0);
          acc[mt][xt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt][s], bm[xt], acc[mt][xt], 0, 0, 0);
          acc[mt][xt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt][s], bh[xt], acc[mt][xt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < KS) {
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) { bh[xt] = nh[xt]; bm[xt] = nm[xt]; }
      }
    }
    // accumulator: row = 4 kg + r (output channel of the M tile), column = vox
    const int yo = yo0 + wave * RW + rw;
    if (yo < a.Ho) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
          const int xo = xo0 + 16 * xt + vox;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 16 * mt + 4 * kg + r;
            if (co < a.Cout && xo < a.Wo) {
              float v = acc[mt][xt][r] + (a.bias ? a.bias[co] : 0.0f);
              if (a.relu) v = __builtin_fmaxf(v, 0.0f);
              a.out[(size_t)co * HWo + (size_t)yo * a.Wo + xo] = v;
            }
          }
        }
    }
    }
    __syncthreads();                  // everyone is done reading the window before the next one is converted
  }
}

// (Cout, Cin, K, K) float32 -> A fragments: element j of fragment [mt][s][piece][lane] is the hi / mid part of
// W[16 mt + (lane & 15)][ci][kh][kw] with k = 32 s + 8 (lane >> 4) + j = (kh * K + kw) * Cin + ci; zero beyond K * K taps and Cout
__global__ void pack_kernel(const float* __restrict__ w, int Cout, int Cin, int K, int KS, int MT, _Float16* __restrict__ frag) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= MT * KS * 64 * 8) return;
  const int j = e & 7, lane = (e >> 3) & 63, s = (e >> 9) % KS, mt = (e >> 9) / KS;
  const int co = 16 * mt + (lane & 15), k = 32 * s + 8 * (lane >> 4) + j;
  const int tap = k / Cin, ci = k - tap * Cin;
  float v = 0.0f;
  if (tap < K * K && co < Cout) v = w[((size_t)co * Cin + ci) * K * K + tap];
  const _Float16 h = (_Float16)v;
  const size_t base = (((size_t)(mt * KS + s) * 2) * 64 + lane) * 8 + j;
  frag[base] = h;
  frag[base + 64 * 8] = (_Float16)(v - (float)h);
}

template <int CIN, int K, int S, int MT, int RW, bool LAT = false>
int launch(Args a, hipStream_t s) {
  constexpr int lds = lds_bytes<CIN, K, S, RW>();
  a.tiles = a.tiles_x * ((a.Ho + kWaves * RW - 1) / (kWaves * RW));
  static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_mfma_kernel<CIN, K, S, MT, RW, LAT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { set_error("svs_conv2d_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  // two workgroups per CU where the registers allow it; a workgroup walks its windows with the weights in registers
  const int grid = a.tiles < 512 ? a.tiles : 512;
  conv2d_mfma_kernel<CIN, K, S, MT, RW, LAT><<<grid, 256, lds, s>>>(a);
  return check_launch("svs_conv2d_mfma");
}

bool supported(int Cin, int Cout, int k, int stride) {
  if (Cout < 1 || Cout > 32) return false;
  if (k == 3 && stride == 1) return Cin == 8 || Cin == 16 || Cin == 32;
  if (k == 5 && stride == 2) return Cin == 8 || Cin == 16;
  return false;
}

// the 3x3 layer (32 -> Cout <= 16) behind a lateral step whose result is formed on the fly (Args::lat_*): H, W even
int run_lateral(const float* lat_in, const float* lat_w, const float* lat_b, const float* lat_add, const void* wfrag,
                const float* bias, float* out, int Cout, int H, int W, int relu, hipStream_t s) {
  if (Cout < 1 || Cout > 16 || (H & 1) || (W & 1)) { set_error("svs_conv2d_mfma: lateral fusion needs Cout <= 16 and even H, W"); return SVS_ESHAPE; }
  Args a;
  a.in = nullptr; a.wfrag = reinterpret_cast<const uint4*>(wfrag); a.bias = bias; a.out = out; a.Cout = Cout; a.H = H; a.W = W;
  a.relu = relu; a.Ho = H; a.Wo = W; a.tiles_x = (W + kTX - 1) / kTX; a.tiles = 0;
  a.lat_in = lat_in; a.lat_w = lat_w; a.lat_b = lat_b; a.lat_add = lat_add;
  const bool tall = (long long)a.tiles_x * ((a.Ho + 3) / 4) > 1024;
  return tall ? launch<32, 3, 1, 1, 2, true>(a, s) : launch<32, 3, 1, 1, 1, true>(a, s);
}

int run(const float* in, const void* wfrag, const float* bias, float* out, int Cin, int Cout, int H, int W, int k, int stride,
        int relu, hipStream_t s) {
  Args a;
  a.lat_in = a.lat_w = a.lat_b = a.lat_add = nullptr;
  a.in = in; a.wfrag = reinterpret_cast<const uint4*>(wfrag); a.bias = bias; a.out = out; a.Cout = Cout; a.H = H; a.W = W;
  a.relu = relu;
  const int pad = k / 2;
  a.Ho = (H + 2 * pad - k) / stride + 1; a.Wo = (W + 2 * pad - k) / stride + 1;
  a.tiles_x = (a.Wo + kTX - 1) / kTX;
  a.tiles = 0;
  const bool two = Cout > 16;
  // 8 x 32 windows once the 4 x 32 windows would be more than two rounds of the launch's 512 workgroups
  const bool tall = (long long)a.tiles_x * ((a.Ho + 3) / 4) > 1024 && stride == 1;
#define SVS_C2M(CIN, K, S) return two ? launch<CIN, K, S, 2, 1>(a, s) : (tall ? launch<CIN, K, S, 1, 2>(a, s) : launch<CIN, K, S, 1, 1>(a, s))
  if (k == 3 && stride == 1) {
    if (Cin == 8) SVS_C2M(8, 3, 1);
    if (Cin == 16) SVS_C2M(16, 3, 1);
    if (Cin == 32) SVS_C2M(32, 3, 1);
  }
  if (k == 5 && stride == 2) {
    if (Cin == 8) SVS_C2M(8, 5, 2);
    if (Cin == 16) SVS_C2M(16, 5, 2);
  }
#undef SVS_C2M
  set_error("svs_conv2d_mfma: unsupported shape (3x3 stride 1 with Cin in {8,16,32}; 5x5 stride 2 with Cin in {8,16}; Cout <= 32)");
  return SVS_ESHAPE;
}

}  // namespace conv2dmfma
}  // namespace svs

using namespace svs;

extern "C" {

int svs_conv2d_mfma_supported(int Cin, int Cout, int k, int stride) { return conv2dmfma::supported(Cin, Cout, k, stride) ? 1 : 0; }

// bytes of the packed A fragments of a layer: [Cout <= 16 ? 1 : 2][ceil(k k Cin / 32)][2][64][16 B]
size_t svs_conv2d_mfma_wfrag_bytes(int Cin, int Cout, int k) {
  return (size_t)(Cout > 16 ? 2 : 1) * ((k * k * Cin + 31) / 32) * 2 * 64 * 16;
}

// weight (Cout,Cin,k,k) float32 on the device (BatchNorm folded) -> wfrag (svs_conv2d_mfma_wfrag_bytes)
int svs_conv2d_mfma_pack(const float* weight, int Cin, int Cout, int k, void* wfrag, void* hip_stream) {
  if (!weight || !wfrag || Cin < 1 || Cout < 1 || Cout > 32 || (k != 3 && k != 5)) { set_error("svs_conv2d_mfma_pack: bad argument"); return SVS_EINVAL; }
  const int KS = (k * k * Cin + 31) / 32, MT = Cout > 16 ? 2 : 1, n = MT * KS * 64 * 8;
  conv2dmfma::pack_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(weight, Cout, Cin, k, KS, MT, static_cast<_Float16*>(wfrag));
  return check_launch("svs_conv2d_mfma_pack");
}

// out (Cout,Ho,Wo) = relu?(conv2d(in (Cin,H,W), k x k, padding k / 2, stride) + bias) on the fp16x2 matrix-core path:
// 3x3 stride 1 with Cin in {8,16,32}, 5x5 stride 2 with Cin in {8,16}; Cout <= 32 (svs_conv2d_mfma_supported)
int svs_conv2d_mfma(const float* in, const void* wfrag, const float* bias, float* out, int Cin, int Cout, int H, int W, int k,
                    int stride, int relu, void* hip_stream) {
  if (!in || !wfrag || !out || H < 1 || W < 1) { set_error("svs_conv2d_mfma: bad argument"); return SVS_EINVAL; }
  return conv2dmfma::run(in, wfrag, bias, out, Cin, Cout, H, W, k, stride, relu, (hipStream_t)hip_stream);
}

// The FPN's lateral step fused into the 3x3 layer behind it: out (Cout <= 16, H, W) = relu?(conv3x3(X) + bias) with
// X[c][y][x] = sum_ci W1[c][ci] lat_in[ci][y][x] + lat_bias[c] + lat_add[c][y / 2][x / 2], c < 32, ci < 8 -- X is formed while
// the kernel converts its input windows and never exists in memory.  lat_weight: the 1x1 weights (32, 8, 1, 1) in svs_conv2d's
// packed layout; wfrag: the 3x3 weights (Cout, 32, 3, 3) packed by svs_conv2d_mfma_pack.  H and W even.
int svs_conv2d_mfma_lateral(const float* lat_in, const float* lat_weight, const float* lat_bias, const float* lat_add,
                            const void* wfrag, const float* bias, float* out, int Cout, int H, int W, int relu, void* hip_stream) {
  if (!lat_in || !lat_weight || !lat_add || !wfrag || !out || H < 2 || W < 2) { set_error("svs_conv2d_mfma_lateral: bad argument"); return SVS_EINVAL; }
  return conv2dmfma::run_lateral(lat_in, lat_weight, lat_bias, lat_add, wfrag, bias, out, Cout, H, W, relu, (hipStream_t)hip_stream);
}

}  // extern "C"
