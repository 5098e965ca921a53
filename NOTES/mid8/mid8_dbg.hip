#include <hip/hip_runtime.h>
#include <cstdio>
#include "svs_blocks_h2.h"
using namespace svs::mlp;
__global__ void dbg(const float* x, unsigned* o) {
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = x[j];
  f16x8 h, m; split8(v, h, m);
  const u32x4 hw = __builtin_bit_cast(u32x4, h), mw = __builtin_bit_cast(u32x4, m);
  for (int i = 0; i < 4; ++i) {
    const u16x2v e = __builtin_bit_cast(u16x2v, hw[i] & 0x7c007c00u);
    const u16x2v lo = {(unsigned short)0x2800, (unsigned short)0x2800};
    const u16x2v top = {(unsigned short)0xA000, (unsigned short)0xA000};
    const u16x2v ec = __builtin_elementwise_max(e, lo);
    const u16x2v invb = (u16x2v)(top - ec);
    const f16x2v inv = __builtin_bit_cast(f16x2v, invb);
    f16x2v t = __builtin_bit_cast(f16x2v, mw[i]) * inv;
    const f16x2v c256 = {(_Float16)256.0f, (_Float16)256.0f}, magic = {(_Float16)1536.0f, (_Float16)1536.0f};
    const unsigned kb = __builtin_bit_cast(unsigned, (f16x2v)(t * c256 + magic));
    o[8 * i + 0] = hw[i]; o[8 * i + 1] = mw[i]; o[8 * i + 2] = __builtin_bit_cast(unsigned, e); o[8 * i + 3] = __builtin_bit_cast(unsigned, ec);
    o[8 * i + 4] = __builtin_bit_cast(unsigned, invb); o[8 * i + 5] = __builtin_bit_cast(unsigned, t); o[8 * i + 6] = kb; o[8 * i + 7] = 0;
  }
  const u32x2 b = mid8_bytes(h, m);
  o[32] = b[0]; o[33] = b[1];
}
int main() {
  float hx[8] = {1.0003f, -1.0003f, 0.3337f, 17.777f, 1.00048828125f, 0.001234f, -0.0f, 300.25f};
  float* dx; unsigned* d; unsigned ho[34];
  (void)hipMalloc(&dx, 32); (void)hipMalloc(&d, 34 * 4);
  (void)hipMemcpy(dx, hx, 32, hipMemcpyHostToDevice);
  dbg<<<1, 1>>>(dx, d);
  (void)hipMemcpy(ho, d, 34 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) printf("pair %d: h %08x m %08x e %08x ec %08x inv %08x t %08x kb %08x\n", i, ho[8*i], ho[8*i+1], ho[8*i+2], ho[8*i+3], ho[8*i+4], ho[8*i+5], ho[8*i+6]);
  printf("bytes %08x %08x\n", ho[32], ho[33]);
}
