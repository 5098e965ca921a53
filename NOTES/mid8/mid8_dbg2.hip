#include <hip/hip_runtime.h>
#include <cstdio>
#include "svs_blocks_h2.h"
using namespace svs::mlp;
__global__ void dbg(const float* x, float* o, unsigned* ob) {
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = x[j];
  f16x8 h; u32x2 m8;
  split8_mid8(v, h, m8);
  ob[0] = m8[0]; ob[1] = m8[1];
  for (int j = 0; j < 8; ++j) { o[j] = mid8_value(h, m8, j); o[8 + j] = (float)h[j]; o[16 + j] = 0.0f; }
}
int main() {
  float hx[8] = {0.0f, 1.0f, -1.0f, 32.0f, 1.00048828125f, 65504.0f, 0.3337f, -17.777f};
  float *dx, *d; unsigned* db; float ho[24]; unsigned hb[2];
  (void)hipMalloc(&dx, 32); (void)hipMalloc(&d, 96); (void)hipMalloc(&db, 8);
  (void)hipMemcpy(dx, hx, 32, hipMemcpyHostToDevice);
  dbg<<<1, 1>>>(dx, d, db);
  (void)hipMemcpy(ho, d, 96, hipMemcpyDeviceToHost); (void)hipMemcpy(hb, db, 8, hipMemcpyDeviceToHost);
  printf("bytes %08x %08x\n", hb[0], hb[1]);
  for (int j = 0; j < 8; ++j) printf("x %.9g hi %.9g mid %.9g unit %.9g -> %.9g\n", hx[j], ho[8 + j], ho[j], ho[16 + j], ho[8 + j] + ho[j]);
}
