// The "split volume": the cost volume in the form conv0 of the regularisation U-Net consumes it (svs_conv_pair.hip),
// written by the fused warp + variance kernel (svs_costvol.hip).  For C = 8 G channels:
//   [D+2][Hp][2 pieces][G][Wp] units of 16 B = the fp16 hi (piece 0) or mid (piece 1) parts of 8 consecutive channels
//   of one voxel; voxel (z,y,x) at padded coordinates (z+1, y+1, x+1); everything outside the interior is zero.
#pragma once
#include <cstddef>

namespace svs {
namespace splitvol {

__host__ __device__ inline int padded_h(int H) { return 4 * ((H + 3) / 4) + 2; }
__host__ __device__ inline int padded_w(int W) { return 32 * ((W + 31) / 32) + 4; }
// index of the unit (z, y, piece, g, x) (unpadded voxel coordinates)
__host__ __device__ inline size_t unit(int z, int y, int piece, int g, int x, int G, int Hp, int Wp) {
  return ((((size_t)(z + 1) * Hp + (y + 1)) * 2 + piece) * G + g) * Wp + (x + 1);
}

}  // namespace splitvol
}  // namespace svs
