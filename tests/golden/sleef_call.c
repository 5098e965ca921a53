/* Trampoline used ONLY while golden fixtures are generated in the build container (tests/golden/ref_shim.py):
 * calls a vector routine exported by torch's own libtorch_cpu.so (e.g. Sleef_expf8_u10, Sleef_expm1f8_u10 -- the
 * open-source sleef code bundled with torch) on a contiguous float32 array, eight lanes at a time.  n must be a
 * multiple of 8 (the caller pads).  Build: gcc -O2 -mavx2 -mfma -shared -fPIC sleef_call.c -o /tmp/.../sleef_call.so */
#include <immintrin.h>
typedef __m256 (*vec8_fn)(__m256);
void svs_apply8(void* fn, const float* in, float* out, long n) {
  vec8_fn f = (vec8_fn)fn;
  for (long i = 0; i < n; i += 8) _mm256_storeu_ps(out + i, f(_mm256_loadu_ps(in + i)));
}
