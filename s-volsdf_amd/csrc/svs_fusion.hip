// Depth-map fusion for gfx950: geometric + photometric consistency filter of one
// reference view against its source views, depth averaging, and ordered back-projection of the surviving pixels
// to a coloured world-space point list.
//
// Reference: helpers/utils.py:75-132 (reproject_with_depth, check_geometric_consistency) and runner.py:301-386
// (filter_depth).  The reference evaluates the geometry in float64 with numpy (integer pixel grids times float32
// depth promote to float64) from 3x3 / 4x4 matrices that it forms in float32; the host side forms the same
// matrices the same way and hands them over as float64, the kernel follows the per-pixel operation order.
// One thread per reference pixel, all source views in registers: 4 B read per (pixel, view) for the reference
// side plus one bilinear fetch of the source depth -- HBM/latency-bound integer-and-double work, no matrix cores.
#include "svs_common.h"
#include "svs_scan.h"

namespace svs {
namespace fusion {

constexpr int kMaxSrc = 16;
constexpr int kMatsPerSrc = 68;      // Kri(9) Trs(16) Ks(9) Ksi(9) Tsr(16) Kr(9), row-major float64

struct FuseArgs {
  const float* ref_depth;            // (H,W)
  const float* confidence;           // (H,W)
  const float* src_depth[kMaxSrc];   // (H,W) each
  const double* mats;                // device, n_src * kMatsPerSrc
  const uint8_t* extra_mask;         // optional (H,W) evaluation mask (runner.py:349-368), nonzero = keep
  int n_src, H, W, thres_view;
  float conf, filter_diff;
  double filter_dist;
  double* depth_avg;                 // (H,W)
  uint8_t *photo_mask, *geo_mask, *final_mask;
  // optional per-source outputs of check_geometric_consistency, (n_src,H,W) each
  uint8_t* src_mask;
  float *src_depth_reproj, *src_x, *src_y;
};

// cv2.remap(src, mapx, mapy, INTER_LINEAR) with the default BORDER_CONSTANT(0), float32 image: OpenCV converts the
// float maps to fixed point with 5 fractional bits (INTER_BITS = 5: cvRound(x * 32)), looks the four weights up in
// its float bilinear table and blends in float32, left to right; taps outside the image read the border value 0.
__device__ __forceinline__ float remap_linear(const float* __restrict__ img, int H, int W, float mx, float my) {
  const float fx32 = mx * 32.0f, fy32 = my * 32.0f;
  // cvRound: round half to even; NaN / out-of-int-range map to INT_MIN on x86 (cvtss2si), i.e. far outside
  const bool bad = !(fx32 > -2.1e9f && fx32 < 2.1e9f) || !(fy32 > -2.1e9f && fy32 < 2.1e9f);
  if (bad) return 0.0f;
  const int sx = (int)__builtin_rintf(fx32), sy = (int)__builtin_rintf(fy32);
  int ix = sx >> 5, iy = sy >> 5;
  ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);            // saturate_cast<short>
  iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
  const float ax = (float)(sx & 31) * (1.0f / 32.0f), ay = (float)(sy & 31) * (1.0f / 32.0f);
  const float w0 = (1.0f - ay) * (1.0f - ax), w1 = (1.0f - ay) * ax, w2 = ay * (1.0f - ax), w3 = ay * ax;
  const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W;
  const bool y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
  const float s0 = (x0 && y0) ? img[(size_t)iy * W + ix] : 0.0f;
  const float s1 = (x1 && y0) ? img[(size_t)iy * W + ix + 1] : 0.0f;
  const float s2 = (x0 && y1) ? img[(size_t)(iy + 1) * W + ix] : 0.0f;
  const float s3 = (x1 && y1) ? img[(size_t)(iy + 1) * W + ix + 1] : 0.0f;
  return s0 * w0 + s1 * w1 + s2 * w2 + s3 * w3;
}

__device__ __forceinline__ void mat3(const double* M, double a, double b, double c, double* o) {
  o[0] = M[0] * a + M[1] * b + M[2] * c;
  o[1] = M[3] * a + M[4] * b + M[5] * c;
  o[2] = M[6] * a + M[7] * b + M[8] * c;
}

// rows 0..2 of a 4x4 times [a,b,c,1]
__device__ __forceinline__ void mat4(const double* M, double a, double b, double c, double* o) {
  o[0] = M[0] * a + M[1] * b + M[2] * c + M[3];
  o[1] = M[4] * a + M[5] * b + M[6] * c + M[7];
  o[2] = M[8] * a + M[9] * b + M[10] * c + M[11];
}

__global__ __launch_bounds__(256) void fuse_view_kernel(FuseArgs a) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int HW = a.H * a.W;
  if (pix >= HW) return;
  const int y = pix / a.W, x = pix - y * a.W;
  const float dref = a.ref_depth[pix];
  const double xd = (double)x, yd = (double)y, dd = (double)dref;
  int geo_sum = 0;
  float dsum = 0.0f;                                     // python sum() of float32 arrays, in view order
  for (int v = 0; v < a.n_src; ++v) {
    const double* M = a.mats + (size_t)v * kMatsPerSrc;
    double p[3], q[3], k[3];
    // step 1: reference pixel -> source view (helpers/utils.py:80-91)
    mat3(M, xd * dd, yd * dd, 1.0 * dd, p);
    mat4(M + 9, p[0], p[1], p[2], q);
    mat3(M + 25, q[0], q[1], q[2], k);
    const double xs = k[0] / k[2], ys = k[1] / k[2];
    const float xsf = (float)xs, ysf = (float)ys;
    // step 2: sample the source depth there and project back (:93-113)
    const float ds = remap_linear(a.src_depth[v], a.H, a.W, xsf, ysf);
    const double dsd = (double)ds;
    mat3(M + 34, xs * dsd, ys * dsd, 1.0 * dsd, p);
    mat4(M + 43, p[0], p[1], p[2], q);
    float drep = (float)q[2];
    mat3(M + 59, q[0], q[1], q[2], k);
    const float xr = (float)(k[0] / k[2]), yr = (float)(k[1] / k[2]);
    // check_geometric_consistency (:116-132): float32 coordinates minus the integer grid promote to float64
    const double ex = (double)xr - xd, ey = (double)yr - yd;
    const double dist = __builtin_sqrt(ex * ex + ey * ey);
    const float rel = __builtin_fabsf(drep - dref) / dref;          // x/0 -> inf, 0/0 -> nan: both fail the test
    const bool ok = dist < a.filter_dist && rel < a.filter_diff;
    if (!ok) drep = 0.0f;
    geo_sum += ok ? 1 : 0;
    dsum = v == 0 ? drep : dsum + drep;                  // 0 + arr is exact
    if (a.src_mask) {
      const size_t o = (size_t)v * HW + pix;
      a.src_mask[o] = ok ? 1 : 0; a.src_depth_reproj[o] = drep; a.src_x[o] = xsf; a.src_y[o] = ysf;
    }
  }
  // runner.py:344-347: (sum + ref) float32, divided by an int32 array -> float64
  const float tot = a.n_src > 0 ? dsum + dref : dref;
  a.depth_avg[pix] = (double)tot / (double)(geo_sum + 1);
  const bool photo = a.confidence[pix] > a.conf;
  const bool geo = geo_sum >= a.thres_view;
  bool fin = photo && geo;
  if (a.extra_mask) fin = fin && a.extra_mask[pix] != 0;
  a.photo_mask[pix] = photo ? 1 : 0; a.geo_mask[pix] = geo ? 1 : 0; a.final_mask[pix] = fin ? 1 : 0;
}

// ---- ordered compaction: vertices in row-major order of the surviving pixels (runner.py:377-386): svs_scan.h -----------

struct PointArgs {
  const double* depth_avg;
  const uint8_t* mask;
  const int* offset;
  const float* img;                  // (H,W,3) float32 in [0,1] (read_img: uint8 / 255), or nullptr
  const double* mats;                // Kri(9) then Eri(16): inv(K_ref), inv(E_ref) formed in float32 by the host
  int H, W;
  float* xyz;                        // (count,3) float32 (the PLY's 'f4' fields)
  uint8_t* rgb;                      // (count,3)
};

__global__ __launch_bounds__(256) void fuse_points_kernel(PointArgs a) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= a.H * a.W || !a.mask[pix]) return;
  const int y = pix / a.W, x = pix - y * a.W;
  const double d = a.depth_avg[pix];
  double p[3], q[3];
  mat3(a.mats, (double)x * d, (double)y * d, 1.0 * d, p);
  mat4(a.mats + 9, p[0], p[1], p[2], q);
  const size_t o = (size_t)a.offset[pix] * 3;
  a.xyz[o] = (float)q[0]; a.xyz[o + 1] = (float)q[1]; a.xyz[o + 2] = (float)q[2];
  if (a.img) {
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb[o + c] = (uint8_t)(int)(a.img[(size_t)pix * 3 + c] * 255.0f);   // astype(uint8) truncates
  }
}

}  // namespace fusion
}  // namespace svs

using namespace svs;
using namespace svs::fusion;

extern "C" {

int svs_fuse_mats_per_src(void) { return kMatsPerSrc; }

int svs_fuse_view(const float* ref_depth, const float* confidence, const float* const* src_depths, const double* mats,
                  int n_src, int H, int W, float conf, double filter_dist, float filter_diff, int thres_view,
                  const uint8_t* extra_mask, double* depth_avg, uint8_t* photo_mask, uint8_t* geo_mask, uint8_t* final_mask,
                  uint8_t* src_mask, float* src_depth_reproj, float* src_x, float* src_y, void* hip_stream) {
  if (!ref_depth || !confidence || !depth_avg || !photo_mask || !geo_mask || !final_mask || (n_src > 0 && (!src_depths || !mats))) {
    set_error("svs_fuse_view: null argument"); return SVS_EINVAL;
  }
  if (n_src < 0 || n_src > kMaxSrc || H < 1 || W < 1) { set_error("svs_fuse_view: bad sizes (n_src <= %d)", kMaxSrc); return SVS_ESHAPE; }
  if (src_mask && (!src_depth_reproj || !src_x || !src_y)) { set_error("svs_fuse_view: per-source outputs come as a set"); return SVS_EINVAL; }
  FuseArgs a;
  a.ref_depth = ref_depth; a.confidence = confidence; a.mats = mats; a.extra_mask = extra_mask;
  for (int v = 0; v < kMaxSrc; ++v) a.src_depth[v] = nullptr;
  for (int v = 0; v < n_src; ++v) {
    if (!src_depths[v]) { set_error("svs_fuse_view: null source depth %d", v); return SVS_EINVAL; }
    a.src_depth[v] = src_depths[v];
  }
  a.n_src = n_src; a.H = H; a.W = W; a.thres_view = thres_view; a.conf = conf; a.filter_diff = filter_diff; a.filter_dist = filter_dist;
  a.depth_avg = depth_avg; a.photo_mask = photo_mask; a.geo_mask = geo_mask; a.final_mask = final_mask;
  a.src_mask = src_mask; a.src_depth_reproj = src_depth_reproj; a.src_x = src_x; a.src_y = src_y;
  fuse_view_kernel<<<(H * W + 255) / 256, 256, 0, (hipStream_t)hip_stream>>>(a);
  return check_launch("svs_fuse_view");
}

int svs_fuse_points(const double* depth_avg, const uint8_t* final_mask, const float* ref_img, const double* mats, int H, int W,
                    int* offset_ws, float* xyz, uint8_t* rgb, int* count, void* hip_stream) {
  if (!depth_avg || !final_mask || !mats || !offset_ws || !xyz || !count || (ref_img && !rgb)) {
    set_error("svs_fuse_points: null argument"); return SVS_EINVAL;
  }
  if (H < 1 || W < 1) { set_error("svs_fuse_points: bad sizes"); return SVS_ESHAPE; }
  hipStream_t s = (hipStream_t)hip_stream;
  scan::mask_offsets(final_mask, (long long)H * W, offset_ws, count, s);
  int rc = check_launch("svs_fuse_points(scan)");
  if (rc) return rc;
  PointArgs a{depth_avg, final_mask, offset_ws, ref_img, mats, H, W, xyz, rgb};
  fuse_points_kernel<<<(H * W + 255) / 256, 256, 0, s>>>(a);
  return check_launch("svs_fuse_points");
}

}  // extern "C"
