// 3x3x3 convolutions of the cost-regularisation U-Net as implicit GEMMs on the fp16 matrix cores, general form:
// stride 1 or 2, and ConvTranspose3d(k3, s2, p1, op1), Cin in {8,16,32,64}, Cout <= 64.
//
// Reference: models/CasMVSNet.py:107-186 (Conv3d / Deconv3d blocks, BatchNorm folded by the caller), :441-472.
// svs_conv_mfma.hip covers the three big stride-1 layers with a register-resident weight set and an LDS ring; this
// kernel covers the rest: the stride-2 and transposed layers and the coarse levels, whose volumes are small
// (2-8 MB, cache-resident) and which were bound by per-thread latency as direct convolutions (one output voxel and a
// 1728-term serial sum per thread at the 64-channel level).
//
// One wave = one tile of 16 output voxels along x times all output channels:
//   v_mfma_f32_16x16x32_f16, M = 16 output channels per M-tile, N = 16 voxels, K = 32 of the (tap, cin) axis.
// Both operands are split a = hi + mid into fp16 pieces (hi*hi + hi*mid + mid*hi, float32 accumulation: float32-class
// accuracy).  The weights arrive pre-split as A fragments (svs_conv3d_gemm_pack); the B fragment
// of a lane is 8 consecutive input channels of one tap of its voxel, read straight from the channel-first volume
// (16 consecutive x per load instruction) and split in registers, one k-step ahead of the MFMAs that use it.
// The transposed convolution is a set of independent GEMMs, one per output parity class: a class uses only the taps
// that meet non-zero input, so no multiplications by the zeros of the up-sampled input are issued.  With Cout <= 32
// the two x-parities of a (pz,py) class share one GEMM: output rows [0,Cout) are the even-x outputs, rows
// [Cout,2Cout) the odd-x ones, over the same B fragments (input voxels xi and xi+1) -- one third fewer input loads,
// full M tiles for Cout = 8, and the even/odd outputs of a channel leave in the same store instruction.
#include "svs_common.h"
#include "svs_split_volume.h"
#include <cstdlib>

namespace svs {
namespace convgemm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Args {
  const float* in;      // (Cin, Di, Hi, Wi)
  const f16x8* wfrag;   // [class][k-step][m-tile][piece][lane]
  const float* bias;    // [Cout] or nullptr
  const float* skip;    // (Cout, Do, Ho, Wo), added after the ReLU, or nullptr
  float* out;           // (Cout, Do, Ho, Wo)
  int Cout, Di, Hi, Wi, Do, Ho, Wo, stride, relu;
  int xtiles, rows;     // tiles per output row (per class row for the transposed form), rows = Dz * Hy
  uint4* split;         // conv_s2c8_kernel only: not null = the output leaves as a split volume (svs_split_volume.h), not to `out`
};

// taps of class (pz,py,px) of the transposed convolution, per dimension: parity 0 -> kernel index 1, input offset 0;
// parity 1 -> kernel index 0 with input offset +1, then kernel index 2 with offset 0   (o = 2 i - 1 + k)
__host__ __device__ inline void deconv_tap(int parity, int idx, int* k, int* d) {
  if (parity == 0) { *k = 1; *d = 0; }
  else if (idx == 0) { *k = 0; *d = 1; }
  else { *k = 2; *d = 0; }
}

__host__ __device__ constexpr int ksteps(int cin, int ntaps) { return (ntaps * cin + 31) / 32; }
__host__ __device__ constexpr int ksteps_max(int cin, bool transposed) { return ksteps(cin, transposed ? 8 : 27); }

// ---- weight packing: [Cin][27][Cout] float32 -> fp16 hi / mid A fragments ------------------------------------------------------
// transposed: 0 convolution, 1 transposed (8 classes), 2 transposed with the x-parities paired (4 classes, M = 2 Cout)
__global__ void pack_kernel(const float* __restrict__ w, int Cin, int Cout, int transposed, int MT, f16x8* __restrict__ frag) {
  const int KS = ksteps_max(Cin, transposed != 0);
  const int n_class = transposed == 1 ? 8 : (transposed >= 2 ? 4 : 1);
  const long long total = (long long)n_class * KS * MT * 64;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i % 64), mt = (int)((i / 64) % MT), s = (int)((i / (64 * MT)) % KS), c = (int)(i / ((long long)64 * MT * KS));
  int co = 16 * mt + (lane & 15);
  const int k0 = 32 * s + 8 * (lane >> 4);
  const int t = k0 / Cin, ci0 = k0 % Cin;
  int tap = -1;
  if (!transposed) {
    if (t < 27) tap = t;
  } else if (transposed == 3) {
    // all classes share the B operand: slot t = (dz*2 + dy)*2 + dxs of the 8 input voxels a 2x2x2 output cell touches;
    // class c = (pz,py) takes offset d along an axis with kernel index {parity 0: d=0 -> 1; parity 1: d=1 -> 0, d=0 -> 2}
    const int pz = (c >> 1) & 1, py = c & 1;
    // row r of the 16: channel 2 (r >> 2) + ((r >> 1) & 1), x parity r & 1 -- a lane's four accumulator rows are then the
    // even / odd outputs of two channels: both parities of a channel leave (and their skip values arrive) as one float2
    const int px = co & 1;
    co = 2 * (co >> 2) + ((co >> 1) & 1);
    const int dz = t >> 2, dy = (t >> 1) & 1, dxs = t & 1;
    const int kz = pz == 0 ? (dz == 0 ? 1 : -1) : (dz == 1 ? 0 : 2);
    const int ky = py == 0 ? (dy == 0 ? 1 : -1) : (dy == 1 ? 0 : 2);
    const int kx = px == 0 ? (dxs == 0 ? 1 : -1) : (dxs == 0 ? 2 : 0);
    if (t < 8 && kz >= 0 && ky >= 0 && kx >= 0) tap = (kz * 3 + ky) * 3 + kx;
  } else if (transposed == 2) {
    // slot t = (a * ny + b) * 2 + dxs: input offset (dz(a), dy(b), dxs); row block px = co / Cout
    const int pz = (c >> 1) & 1, py = c & 1;
    const int ny = 1 + py, nz = 1 + pz;
    const int px = co >= Cout ? 1 : 0;
    co -= px * Cout;
    if (px == 1 && co >= Cout) co = 1 << 20;                 // rows beyond 2 * Cout: zero
    if (t < nz * ny * 2) {
      const int dxs = t & 1, ab = t >> 1;
      int kz, ky, d;
      deconv_tap(pz, ab / ny, &kz, &d); deconv_tap(py, ab % ny, &ky, &d);
      // even x (px = 0): kernel index 1 on input xi; odd x: index 2 on xi, index 0 on xi + 1
      const int kx = px == 0 ? (dxs == 0 ? 1 : -1) : (dxs == 0 ? 2 : 0);
      if (kx >= 0) tap = (kz * 3 + ky) * 3 + kx;
    }
  } else {
    const int pz = (c >> 2) & 1, py = (c >> 1) & 1, px = c & 1;
    const int nx = 1 + px, ny = 1 + py, nz = 1 + pz;
    if (t < nx * ny * nz) {
      int kz, ky, kx, d;
      deconv_tap(pz, t / (nx * ny), &kz, &d); deconv_tap(py, (t / nx) % ny, &ky, &d); deconv_tap(px, t % nx, &kx, &d);
      tap = (kz * 3 + ky) * 3 + kx;
    }
  }
  f16x8 hi, mid;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (tap >= 0 && co < Cout) ? w[((size_t)(ci0 + j) * 27 + tap) * Cout + co] : 0.0f;
    const _Float16 h = (_Float16)v;
    hi[j] = h; mid[j] = (_Float16)(v - (float)h);
  }
  const size_t o = (((size_t)c * KS + s) * MT + mt) * 2 * 64 + lane;
  frag[o] = hi; frag[o + 64] = mid;
}

// MODE 0: convolution (stride 1 or 2); MODE 1: transposed convolution, 8 classes; MODE 2: transposed, x-parities paired
// KSPLIT: the four waves of a workgroup share ONE tile, wave w takes the k-steps s = w (mod 4) and the partial sums meet in
// LDS -- for the coarse levels (fewer tiles than SIMDs x 2), where a wave's k-steps are a serial chain of latencies and three
// quarters of the chip would otherwise idle; every wave then finishes the M tiles m = w (mod 4).
template <int CIN, int MT, int MODE, bool KSPLIT>
__global__ __launch_bounds__(256) void conv_gemm_kernel(Args a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  long long tile = KSPLIT ? (long long)blockIdx.x : (long long)blockIdx.x * 4 + wave;
  const long long per_class = (long long)a.rows * a.xtiles;
  int cls = 0;
  if (MODE >= 1) { cls = (int)(tile / per_class); tile -= (long long)cls * per_class; if (cls >= (MODE == 1 ? 8 : 4)) return; }
  else if (tile >= per_class) return;
  const int xt = (int)(tile % a.xtiles), row = (int)(tile / a.xtiles);
  const int n = lane & 15, g = lane >> 4;
  const int pz = MODE == 2 ? (cls >> 1) & 1 : (cls >> 2) & 1, py = MODE == 2 ? cls & 1 : (cls >> 1) & 1, px = MODE == 2 ? 0 : cls & 1;
  // output voxel of this lane's column and the base input coordinate of tap (0,0,0)
  int zo, yo, xo, zb, yb, xb, ntaps;
  bool col_ok;
  if (MODE == 0) {
    zo = row / a.Ho; yo = row - zo * a.Ho; xo = 16 * xt + n;
    col_ok = xo < a.Wo;
    zb = zo * a.stride - 1; yb = yo * a.stride - 1; xb = xo * a.stride - 1;
    ntaps = 27;
  } else {
    const int zi = row / a.Hi, yi = row - zi * a.Hi, xi = 16 * xt + n;
    col_ok = xi < a.Wi;
    zo = 2 * zi + pz; yo = 2 * yi + py; xo = 2 * xi + px;
    zb = zi; yb = yi; xb = xi;
    ntaps = MODE == 2 ? (1 + pz) * (1 + py) * 2 : (1 + pz) * (1 + py) * (1 + px);
  }
  const int KS = ksteps(CIN, ntaps);
  const size_t chan_in = (size_t)a.Di * a.Hi * a.Wi;
  const f16x8* __restrict__ wf = a.wfrag + (size_t)cls * ksteps_max(CIN, MODE >= 1) * MT * 128 + lane;

  // B operand of k-step s for this lane: 8 consecutive input channels of one tap at this lane's voxel
  auto load_b = [&](int s, float* x) {
    const int k0 = 32 * s + 8 * g;
    const int t = k0 / CIN, ci0 = k0 % CIN;
    int iz, iy, ix;
    if (MODE == 0) {
      const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
      iz = zb + kz; iy = yb + ky; ix = xb + kx;
    } else if (MODE == 1) {
      const int nx = 1 + px, ny = 1 + py;
      int k_, dz, dy, dx;
      deconv_tap(pz, t / (nx * ny), &k_, &dz); deconv_tap(py, (t / nx) % ny, &k_, &dy); deconv_tap(px, t % nx, &k_, &dx);
      iz = zb + dz; iy = yb + dy; ix = xb + dx;
    } else {
      const int ny = 1 + py, ab = t >> 1;
      int k_, dz, dy;
      deconv_tap(pz, ab / ny, &k_, &dz); deconv_tap(py, ab % ny, &k_, &dy);
      iz = zb + dz; iy = yb + dy; ix = xb + (t & 1);
    }
    const bool ok = col_ok && t < ntaps && (unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
    const float* p = a.in + (size_t)ci0 * chan_in + ((size_t)(ok ? iz : 0) * a.Hi + (ok ? iy : 0)) * a.Wi + (ok ? ix : 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = p[(size_t)j * chan_in]; x[j] = ok ? v : 0.0f; }
  };

  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // Both operands of k-step s+1 are requested before the MFMAs of k-step s: the B fragment (input voxels) and the A
  // fragments (weights, from L2).  The coarse levels of the U-Net run well under one wave per SIMD (768 waves at
  // 24 x 16 x 20), so a wave's k-steps are a serial chain; with the weights read at their point of use every k-step
  // waited out an L2 round trip (conv6: 54 of them).
  float xc[8], xn[8];
  f16x8 ahc[MT], amc[MT], ahn[MT], amn[MT];
  constexpr int SS = KSPLIT ? 4 : 1;                 // stride between this wave's k-steps
  const int s0 = KSPLIT ? wave : 0;
  if (s0 < KS) {
    load_b(s0, xc);
#pragma unroll
#ifdef CONV_ABL_W0
    for (int m = 0; m < MT; ++m) { ahc[m] = wf[0]; amc[m] = wf[64]; }
#else
    for (int m = 0; m < MT; ++m) { ahc[m] = wf[((size_t)s0 * MT + m) * 128]; amc[m] = wf[((size_t)s0 * MT + m) * 128 + 64]; }
#endif
  }
  for (int s = s0; s < KS; s += SS) {
    if (s + SS < KS) {
      load_b(s + SS, xn);
      const f16x8* wn = wf + (size_t)(s + SS) * MT * 128;
#pragma unroll
#ifdef CONV_ABL_W0
      for (int m = 0; m < MT; ++m) { ahn[m] = wf[0]; amn[m] = wf[64]; }
#else
      for (int m = 0; m < MT; ++m) { ahn[m] = wn[m * 128]; amn[m] = wn[m * 128 + 64]; }
#endif
    }
    f16x8 bh, bm;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 h = (_Float16)xc[j];
      bh[j] = h; bm[j] = (_Float16)(xc[j] - (float)h);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(amc[m], bh, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahc[m], bm, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahc[m], bh, acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) xc[j] = xn[j];
#pragma unroll
    for (int m = 0; m < MT; ++m) { ahc[m] = ahn[m]; amc[m] = amn[m]; }
  }
  if (KSPLIT) {
    __shared__ f32x4 part[4][MT][64];
#pragma unroll
    for (int m = 0; m < MT; ++m) part[wave][m][lane] = acc[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
      if ((m & 3) == wave) acc[m] = (part[0][m][lane] + part[1][m][lane]) + (part[2][m][lane] + part[3][m][lane]);
  }
  if (!col_ok) return;
  const size_t chan_out = (size_t)a.Do * a.Ho * a.Wo;
  size_t vox = ((size_t)zo * a.Ho + yo) * a.Wo + xo;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KSPLIT && (m & 3) != wave) continue;
      int co = 16 * m + 4 * g + j;                 // accumulator row of v_mfma_f32_16x16x32: 4 * (lane >> 4) + j
      if (MODE == 2) {                             // rows [Cout, 2 Cout): the odd-x outputs (Cout is a multiple of 4)
        const int odd = co >= a.Cout ? 1 : 0;
        co -= odd * a.Cout;
        vox = ((size_t)zo * a.Ho + yo) * a.Wo + xo + odd;
      }
      if (co >= a.Cout) continue;
      float r = acc[m][j] + (a.bias ? a.bias[co] : 0.0f);
      if (a.relu) r = __builtin_fmaxf(r, 0.0f);
      const size_t o = (size_t)co * chan_out + vox;
      if (a.skip) r += a.skip[o];
      a.out[o] = r;
    }
}

// MODE 3 (Cin = 16, Cout = 8: conv11, the transposed convolution back to full resolution): one wave computes the whole
// 2x2x2 output cell of each of its 16 input voxels -- the four (pz,py) classes as four M tiles (rows = 8 channels x
// 2 x-parities) over ONE set of B fragments, the 8 input voxels (zi + dz, yi + dy, xi + dxs) x 16 channels = 4 k-steps;
// a class skips the k-steps whose (dz,dy) it does not touch (9 of 16 blocks remain: the MFMA count of MODE 2).
// The layer is HBM traffic (283 MB at stage 1: skip in, output out) behind a handful of MFMAs; as four MODE-2 waves per
// cell each wave had ~3 KB in flight and the latency of its skip read at the very end (0.123 ms = 2.3 TB/s).  Here
// every load of the wave -- 4 k-steps of B and the 16 skip values per lane -- is issued before the first MFMA.
// MTC = M tiles per class: 1 (conv11, Cin 16 -> Cout 8) or 2 (conv9, 32 -> 16); M tile mt of a class holds channels 8 mt ..
// 8 mt + 7 in the (channel, parity) row order.  One k-step = 32 / CIN of the 8 input voxels of the cell.
template <int CIN, int MTC>
__global__ __launch_bounds__(256) void deconv_cell_kernel(Args a) {
  static_assert(CIN == 16 || CIN == 32, "k-steps cover whole input voxels");
  constexpr int KS = 8 * CIN / 32;
  constexpr int VPS = 32 / CIN;                       // input voxels (offsets) per k-step: 2 or 1
  const int lane = threadIdx.x & 63;
  const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= (long long)a.rows * a.xtiles) return;
  const int xt = (int)(tile % a.xtiles), row = (int)(tile / a.xtiles);
  const int n = lane & 15, g = lane >> 4;
  const int zi = row / a.Hi, yi = row - zi * a.Hi, xi = 16 * xt + n;
  const bool col_ok = xi < a.Wi;
  const size_t chan_in = (size_t)a.Di * a.Hi * a.Wi, chan_out = (size_t)a.Do * a.Ho * a.Wo;
  // accumulator rows 4 g + j of a lane in M tile mt: (channel 8 mt + 2 g + (j >> 1), x = 2 xi + (j & 1)) -- pack_kernel, mode 3
  // ---- skip values of the cell (8 MTC float2 per lane), requested first
  float2 sk[4][MTC][2];
  size_t obase[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    obase[m] = (size_t)(2 * g) * chan_out + ((size_t)(2 * zi + (m >> 1)) * a.Ho + (2 * yi + (m & 1))) * a.Wo + 2 * xi;
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        sk[m][mt][q] = (a.skip && col_ok) ? *reinterpret_cast<const float2*>(a.skip + obase[m] + (size_t)(8 * mt + q) * chan_out)
                                          : float2{0.0f, 0.0f};
  }
  // ---- B operand: offset t = (dz*2 + dy)*2 + dxs of the cell's 8 input voxels.  CIN = 16: k-step s = offsets 2s (lane groups
  // 0,1) and 2s+1 (groups 2,3), 8 channels per group; CIN = 32: k-step s = offset s, lane group g = channels 8g..8g+7
  float x[KS][8];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int t = VPS * s + (VPS == 2 ? (g >> 1) : 0);
    const int c0 = VPS == 2 ? 8 * (g & 1) : 8 * g;
    const int iz = zi + (t >> 2), iy = yi + ((t >> 1) & 1), ix = xi + (t & 1);
    const bool ok = col_ok && iz < a.Di && iy < a.Hi && ix < a.Wi;
    const float* p = a.in + (size_t)c0 * chan_in + ((size_t)(ok ? iz : 0) * a.Hi + (ok ? iy : 0)) * a.Wi + (ok ? ix : 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = p[(size_t)j * chan_in]; x[s][j] = ok ? v : 0.0f; }
  }
  const f16x8* __restrict__ wf = a.wfrag + lane;      // [class][k-step][M tile][piece][lane]
  f32x4 acc[4][MTC];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) acc[m][mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f16x8 bh, bm;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 h = (_Float16)x[s][j];
      bh[j] = h; bm[j] = (_Float16)(x[s][j] - (float)h);
    }
    const int dz = (VPS * s) >> 2, dy = ((VPS * s) >> 1) & 1;     // (the offsets of a k-step share dz, dy)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (dz > (m >> 1) || dy > (m & 1)) continue;                // the class does not reach these input rows
#pragma unroll
      for (int mt = 0; mt < MTC; ++mt) {
#ifdef CONV_ABL_W0      // diagnostic: every fragment is fragment 0 (no weight traffic beyond one line per lane)
        const f16x8 ah = wf[0], am = wf[64];
#else
        const f16x8 ah = wf[((m * KS + s) * MTC + mt) * 128], am = wf[((m * KS + s) * MTC + mt) * 128 + 64];
#endif
        acc[m][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bh, acc[m][mt], 0, 0, 0);
        acc[m][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bm, acc[m][mt], 0, 0, 0);
        acc[m][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[m][mt], 0, 0, 0);
      }
    }
  }
  if (!col_ok) return;
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) {
    float bias[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) bias[q] = a.bias ? a.bias[8 * mt + 2 * g + q] : 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float r0 = acc[m][mt][2 * q] + bias[q], r1 = acc[m][mt][2 * q + 1] + bias[q];
        if (a.relu) { r0 = __builtin_fmaxf(r0, 0.0f); r1 = __builtin_fmaxf(r1, 0.0f); }
        *reinterpret_cast<float2*>(a.out + obase[m] + (size_t)(8 * mt + q) * chan_out) =
            float2{r0 + sk[m][mt][q].x, r1 + sk[m][mt][q].y};
      }
  }
}

// conv1 of the U-Net (8 -> 16 channels, stride 2, full resolution in: models/CasMVSNet.py:445,461): the general kernel
// gathers every tap with 4-byte loads at stride 2 (56 load instructions per lane, three per input row touching the
// same lines) and was bound by the vector L1 (0.082 ms at stage 1 for 126 MB of input).  Here a lane group owns whole
// (kz,ky) input rows: one float2 load per (row, channel) brings the columns 2xo, 2xo+1 = taps kx 1 and 2, and tap
// kx 0 (column 2xo-1) is the neighbour lane's second value (DPP row shift).  A wave's 16 columns are 15 outputs plus,
// in lane 0, the column to their left, which only feeds lane 1 (loading the edge column separately cost as much L1
// time as a full gather: the cost is per line touched).  K order: k-step s = kx*3 + q carries rows g + 4q (g = lane
// group; rows 9..11: zero weights), 8 channels each: 9 k-steps instead of 7, 24 loads per lane instead of 56, all
// issued up front.
__global__ __launch_bounds__(256) void conv_s2c8_kernel(Args a) {
  const int lane = threadIdx.x & 63;
  const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= (long long)a.rows * a.xtiles) return;
  const int xt = (int)(tile % a.xtiles), row = (int)(tile / a.xtiles);
  const int n = lane & 15, g = lane >> 4;
  const int zo = row / a.Ho, yo = row - zo * a.Ho, xo = 15 * xt + n - 1;
  const bool col_in = xo >= 0 && xo < a.Wo;           // lane 0 of the first tile: the zero padding left of the volume
  const bool col_ok = col_in && n > 0;                // lane 0 computes nothing of its own
  const size_t chan_in = (size_t)a.Di * a.Hi * a.Wi;
  const int xc = col_in ? 2 * xo : 0;                 // even column of this lane's pair (Wi is even: 2xo+1 < Wi)
  float x0[3][8], x1[3][8], x2[3][8];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int r = g + 4 * q;
    const int iz = 2 * zo - 1 + r / 3, iy = 2 * yo - 1 + r % 3;
    const bool ok = col_in && r < 9 && (unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi;
    const float* p = a.in + ((size_t)(ok ? iz : 0) * a.Hi + (ok ? iy : 0)) * a.Wi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2 v = *reinterpret_cast<const float2*>(p + (size_t)j * chan_in + xc);
      const float odd = ok ? v.y : 0.0f;
      // lane n takes lane n-1's odd column (s_nop: DPP source hazard; inline assembly: see the note in svs_costvol.hip,
      // conv3d_c1_flat_kernel)
      float e = 0.0f;
      asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(e) : "v"(odd));   // (not volatile: a
      // volatile asm is a barrier to the loads around it and the 24 loads would be waited for one by one)
      x0[q][j] = e;
      x1[q][j] = ok ? v.x : 0.0f;
      x2[q][j] = odd;
    }
  }
  const f16x8* __restrict__ wf = a.wfrag + lane;      // [k-step][piece][lane]
  f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  auto kstep = [&](int s, const float* x) {
    f16x8 bh, bm;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 h = (_Float16)x[j];
      bh[j] = h; bm[j] = (_Float16)(x[j] - (float)h);
    }
    const f16x8 ah = wf[s * 128], am = wf[s * 128 + 64];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  };
#pragma unroll
  for (int q = 0; q < 3; ++q) { kstep(q, x0[q]); kstep(3 + q, x1[q]); kstep(6 + q, x2[q]); }
  if (!col_ok) return;
  const size_t chan_out = (size_t)a.Do * a.Ho * a.Wo;
  const size_t vox = ((size_t)zo * a.Ho + yo) * a.Wo + xo;
  if (a.split) {
    // the next layer (conv2, svs_conv3d_rows) reads fp16 hi / mid pieces, 8 channels per 16-byte unit: this lane's four
    // channels 4g .. 4g+3 are one half of unit g >> 1 (Cout = 16: rows beyond Cout hold zeros, their weights being zero)
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 h, m;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r = acc[j] + ((a.bias && 4 * g + j < a.Cout) ? a.bias[4 * g + j] : 0.0f);
      if (a.relu) r = __builtin_fmaxf(r, 0.0f);
      const _Float16 hh = (_Float16)r;
      h[j] = hh; m[j] = (_Float16)(r - (float)hh);
    }
    const int Gs = 2, Hp = splitvol::padded_h(a.Ho), Wp = splitvol::padded_w(a.Wo);
    uint2* u = reinterpret_cast<uint2*>(a.split + splitvol::unit(zo, yo, 0, g >> 1, xo, Gs, Hp, Wp)) + (g & 1);
    u[0] = __builtin_bit_cast(uint2, h);
    u[(size_t)Gs * Wp * 2] = __builtin_bit_cast(uint2, m);
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = 4 * g + j;
    if (co >= a.Cout) continue;
    float r = acc[j] + (a.bias ? a.bias[co] : 0.0f);
    if (a.relu) r = __builtin_fmaxf(r, 0.0f);
    const size_t o = (size_t)co * chan_out + vox;
    if (a.skip) r += a.skip[o];
    a.out[o] = r;
  }
}

constexpr long long kKsplitTiles = 2048;   // measured: pays below ~2 tiles per SIMD-pair (conv5-conv7 of every stage), costs above

template <int CIN, int MODE>
static void launch_mt(const Args& a, int MT, long long tiles, hipStream_t s) {
  // few tiles (the coarse levels): four waves per tile; SVS_GEMM_KSPLIT=0/1 forces the choice (measurements)
  static const char* env = getenv("SVS_GEMM_KSPLIT");
  const bool ksplit = (env && env[0]) ? env[0] == '1' : tiles <= kKsplitTiles;
  if (ksplit) {
    const unsigned grid = (unsigned)tiles;
    if (MT == 1) conv_gemm_kernel<CIN, 1, MODE, true><<<grid, 256, 0, s>>>(a);
    else if (MT == 2) conv_gemm_kernel<CIN, 2, MODE, true><<<grid, 256, 0, s>>>(a);
    else conv_gemm_kernel<CIN, 4, MODE, true><<<grid, 256, 0, s>>>(a);
    return;
  }
  const unsigned grid = (unsigned)((tiles + 3) / 4);
  if (MT == 1) conv_gemm_kernel<CIN, 1, MODE, false><<<grid, 256, 0, s>>>(a);
  else if (MT == 2) conv_gemm_kernel<CIN, 2, MODE, false><<<grid, 256, 0, s>>>(a);
  else conv_gemm_kernel<CIN, 4, MODE, false><<<grid, 256, 0, s>>>(a);
}

template <int MODE>
static bool launch_cin(const Args& a, int Cin, int MT, long long tiles, hipStream_t s) {
  switch (Cin) {
    case 8: launch_mt<8, MODE>(a, MT, tiles, s); return true;
    case 16: launch_mt<16, MODE>(a, MT, tiles, s); return true;
    case 32: launch_mt<32, MODE>(a, MT, tiles, s); return true;
    case 64: launch_mt<64, MODE>(a, MT, tiles, s); return true;
    default: return false;
  }
}

static int m_tiles(int rows) { return rows <= 16 ? 1 : (rows <= 32 ? 2 : 4); }
// the transposed form pairs the x-parities when both fit the 64 output rows of a wave and rows split on lanes' groups of 4
static int deconv_mode(int Cin, int Cout) {
  if ((Cin == 16 && Cout == 8) || (Cin == 32 && Cout == 16)) return 3;     // whole 2x2x2 cells per wave (deconv_cell_kernel)
  return (Cout <= 32 && Cout % 4 == 0) ? 2 : 1;
}

}  // namespace convgemm
}  // namespace svs

using namespace svs;
using namespace svs::convgemm;

extern "C" {

int svs_conv3d_gemm_supported(int Cin, int Cout) { return (Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64) && Cout >= 1 && Cout <= 64; }

size_t svs_conv3d_gemm_wfrag_bytes(int Cin, int Cout, int transposed) {
  if (!svs_conv3d_gemm_supported(Cin, Cout)) return 0;
  const int mode = transposed ? deconv_mode(Cin, Cout) : 0;
  return (size_t)(mode == 1 ? 8 : (mode >= 2 ? 4 : 1)) * ksteps_max(Cin, mode != 0) * m_tiles(mode >= 2 ? 2 * Cout : Cout) * 128 * sizeof(f16x8);
}

int svs_conv3d_gemm_pack(const float* weight, int Cin, int Cout, int transposed, void* wfrag, void* hip_stream) {
  if (!weight || !wfrag || !svs_conv3d_gemm_supported(Cin, Cout)) { set_error("svs_conv3d_gemm_pack: bad argument"); return SVS_EINVAL; }
  const int mode = transposed ? deconv_mode(Cin, Cout) : 0;
  const int MT = m_tiles(mode >= 2 ? 2 * Cout : Cout);
  const long long total = (long long)(mode == 1 ? 8 : (mode >= 2 ? 4 : 1)) * ksteps_max(Cin, mode != 0) * MT * 64;
  pack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)hip_stream>>>(weight, Cin, Cout, mode, MT, (f16x8*)wfrag);
  return check_launch("svs_conv3d_gemm_pack");
}

int svs_conv3d_gemm(const float* in, const void* wfrag, const float* bias, const float* skip, float* out, int Cin, int Cout,
                    int Di, int Hi, int Wi, int stride, int transposed, int relu, void* hip_stream) {
  if (!in || !wfrag || !out || Di < 1 || Hi < 1 || Wi < 1 || (stride != 1 && stride != 2)) { set_error("svs_conv3d_gemm: bad argument"); return SVS_EINVAL; }
  if (!svs_conv3d_gemm_supported(Cin, Cout)) { set_error("svs_conv3d_gemm: Cin must be 8/16/32/64 and Cout <= 64"); return SVS_ESHAPE; }
  Args a;
  a.in = in; a.wfrag = (const f16x8*)wfrag; a.bias = bias; a.skip = skip; a.out = out; a.Cout = Cout;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.stride = stride; a.relu = relu; a.split = nullptr;
  hipStream_t s = (hipStream_t)hip_stream;
  int MT = m_tiles(Cout);
  if (transposed) {
    a.Do = 2 * Di; a.Ho = 2 * Hi; a.Wo = 2 * Wi;
    a.xtiles = (Wi + 15) / 16; a.rows = Di * Hi;
    const int mode = deconv_mode(Cin, Cout);
    if (mode == 3) {
      const long long tiles = (long long)a.rows * a.xtiles;
      if (Cin == 16) deconv_cell_kernel<16, 1><<<(unsigned)((tiles + 3) / 4), 256, 0, s>>>(a);
      else deconv_cell_kernel<32, 2><<<(unsigned)((tiles + 3) / 4), 256, 0, s>>>(a);
    } else if (mode == 2) { MT = m_tiles(2 * Cout); launch_cin<2>(a, Cin, MT, 4LL * a.rows * a.xtiles, s); }
    else launch_cin<1>(a, Cin, MT, 8LL * a.rows * a.xtiles, s);
  } else {
    a.Do = (Di - 1) / stride + 1; a.Ho = (Hi - 1) / stride + 1; a.Wo = (Wi - 1) / stride + 1;
    a.xtiles = (a.Wo + 15) / 16; a.rows = a.Do * a.Ho;
    launch_cin<0>(a, Cin, MT, (long long)a.rows * a.xtiles, s);
  }
  return check_launch("svs_conv3d_gemm");
}

// The stride-2 convolution from 8 channels (conv1) with whole input rows per lane group (conv_s2c8_kernel): Cout <= 16,
// Wi even.  wfrag: 9 k-steps x 2 pieces x 64 lanes x 16 B: row = lane & 15 (output channel), k = 32 s + 8 g + j with
// s = kx*3 + q, g = lane >> 4: the folded weight of tap ((g + 4q) / 3, (g + 4q) % 3, kx) and input channel j (zero for
// g + 4q > 8).
size_t svs_conv3d_s2c8_wfrag_bytes(void) { return (size_t)9 * 2 * 64 * 16; }
// split_out: not null = the output (Cout = 16 channels) leaves as a split volume (svs_split_volume_dims(16, Do, Ho, Wo)
// bytes, zero-filled by the caller before its first use) instead of float32 to `out` -- the form svs_conv3d_rows reads.
int svs_conv3d_s2c8(const float* in, const void* wfrag, const float* bias, const float* skip, float* out, void* split_out,
                    int Cout, int Di, int Hi, int Wi, int relu, void* hip_stream) {
  if (!in || !wfrag || (!out && !split_out) || Cout < 1 || Cout > 16 || Di < 1 || Hi < 1 || Wi < 2 || (Wi & 1) ||
      (split_out && (Cout != 16 || skip))) {
    set_error("svs_conv3d_s2c8: bad argument (Cout <= 16, Wi even; split_out: Cout = 16, no skip)"); return SVS_EINVAL;
  }
  Args a;
  a.in = in; a.wfrag = (const f16x8*)wfrag; a.bias = bias; a.skip = skip; a.out = out; a.Cout = Cout;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.stride = 2; a.relu = relu; a.split = reinterpret_cast<uint4*>(split_out);
  a.Do = (Di - 1) / 2 + 1; a.Ho = (Hi - 1) / 2 + 1; a.Wo = (Wi - 1) / 2 + 1;
  a.xtiles = (a.Wo + 14) / 15; a.rows = a.Do * a.Ho;
  const long long tiles = (long long)a.rows * a.xtiles;
  conv_s2c8_kernel<<<(unsigned)((tiles + 3) / 4), 256, 0, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_conv3d_s2c8");
}

}  // extern "C"
