// Host-side helpers shared by the C-ABI entry points of the MLP kernels.
#pragma once
#include "svs_mlp_dev.h"

namespace svs {
namespace mlp {

inline int fill_src(PointSrc& src, const float* points, int n_points, const float* cam, int cam_stride,
                    const float* dirs, const float* z, int S, int n_rays, const char* who) {
  if (n_points < 0 || n_rays < 0 || (n_points == 0 && n_rays == 0)) { set_error("%s: no points", who); return SVS_ESHAPE; }
  if (n_points > 0 && !points) { set_error("%s: n_points > 0 but points is null", who); return SVS_EINVAL; }
  if (n_rays > 0 && !(cam && dirs && z && S > 0 && (cam_stride == 0 || cam_stride == 3))) {
    set_error("%s: the ray part needs cam/dirs/z, S > 0 and cam_stride in {0,3}", who); return SVS_EINVAL;
  }
  if ((long long)n_rays * S + n_points > 0x7fffffffLL) { set_error("%s: too many points", who); return SVS_ESHAPE; }
  src.pts = points; src.cam = cam; src.dirs = dirs; src.z = z; src.cam_stride = cam_stride; src.S = S > 0 ? S : 1;
  src.n_ray = n_rays * (S > 0 ? S : 0);
  src.P = src.n_ray + n_points;
  return SVS_OK;
}
inline int wave_tiles(int n_points) { return (n_points + kWgPts - 1) / kWgPts * kWaves; }

template <typename K>
inline int set_lds(K kernel, int bytes, const char* who) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e)); return (int)e; }
  return SVS_OK;
}

}  // namespace mlp
}  // namespace svs
