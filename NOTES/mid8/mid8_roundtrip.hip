// Round trip of the three-byte block encoding (csrc/svs_blocks_h2.h: split8_mid8 / mid8_value) on random values:
//   hipcc --offload-arch=gfx950 -I s-volsdf_amd/csrc tools/micro/mid8_roundtrip.hip -o /tmp/mid8 && /tmp/mid8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "svs_blocks_h2.h"
using namespace svs::mlp;

__global__ void roundtrip(const float* x, float* y, unsigned* bytes, int n8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = x[8 * i + j];
  f16x8 h; u32x2 m8;
  split8_mid8(v, h, m8);
  bytes[2 * i] = m8[0]; bytes[2 * i + 1] = m8[1];
  const f16x8 mf = mid8_fragment(h, m8);           // what the weight-gradient GEMM multiplies: must be the same value
  for (int j = 0; j < 8; ++j) {
    const float a = mid8_value(h, m8, j), b = (float)mf[j];
    y[8 * i + j] = (float)h[j] + (a == b || (a != a && b != b) ? a : __builtin_nanf(""));
  }
}

int main() {
  const int n8 = 1 << 16, n = 8 * n8;
  std::vector<float> x(n), y(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const double u = rand() / (double)RAND_MAX, s = rand() / (double)RAND_MAX, t = rand() / (double)RAND_MAX;
    x[i] = (float)((t < 0.5 ? -1 : 1) * (0.5 + u) * std::exp2(-26.0 + 31.0 * s));
  }
  x[0] = 0.0f; x[1] = 1.0f; x[2] = -1.0f; x[3] = 32.0f; x[4] = 1.00048828125f; x[5] = 65504.0f;
  float *dx, *dy; unsigned* db;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&db, n8 * 8);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  roundtrip<<<n8 / 256, 256>>>(dx, dy, db, n8);
  hipMemcpy(y.data(), dy, n * 4, hipMemcpyDeviceToHost);
  double worst_rel = 0, worst_abs = 0; int bad = 0;
  for (int i = 0; i < n; ++i) {
    const double e = std::fabs((double)y[i] - x[i]);
    if (!(e == e)) { if (bad++ < 5) printf("nan at %d: x %g y %g\n", i, x[i], y[i]); continue; }
    if (std::fabs(x[i]) >= 1.0 / 32) worst_rel = std::fmax(worst_rel, e / std::fabs(x[i])); else worst_abs = std::fmax(worst_abs, e);
  }
  printf("max rel err (|x| >= 2^-5) %.3g (2^-18 = 3.8e-6), max abs err below %.3g (2^-24 = 6e-8), non-finite %d\n", worst_rel, worst_abs, bad);
  printf("first: %g %g %g %g %g %g\n", y[0], y[1], y[2], y[3], y[4], y[5]);
  return 0;
}
