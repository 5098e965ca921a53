// Error reporting shared by every C-ABI entry point (thread-local last-error string).
#include "svs_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
namespace svs {
static std::atomic<int> g_deterministic{0};
bool deterministic() { return g_deterministic.load() != 0; }
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace svs

// ---- host: the training pixels of a step --------------------------------------------------------------------------------------
// The reference draws a step's pixels as torch.randperm(total_pixels)[:num_pixels] (volsdf/datasets/scene_dataset.py:275-279,
// called from volsdf/vsdf.py:234 after every step): a Fisher-Yates shuffle of ALL pixels of the image (442 368 at
// 576 x 768: 2-3.5 ms on one host thread, more than a 256-ray step takes on the GPU) of which the first num_pixels entries
// are kept.  ATen's shuffle (aten/src/ATen/native/TensorFactories.cpp, randperm_cpu, n < 2^32 / 20) is the forward form
//     r[i] = i;   for i in 0 .. n-2:  z = mt19937() % (n - i);  swap(r[i], r[i + z])
// so entry i is final after iteration i.  svs_randperm_prefix makes the first k iterations on a persistent identity array
// (undone afterwards: O(k)), and advances the generator over the remaining n - 1 - k draws without forming them: the same
// k indices, the same generator state afterwards, i.e. the same batches and the same random stream for everything that
// follows.  The generator is the serialised CPU generator of torch.get_rng_state() (at::CPUGeneratorImplState, 5056 bytes:
// seed u64 | left i32 | seeded i32 | next u64 | state u64[624] | normal-sample cache), at::mt19937 of
// ATen/core/MT19937RNGEngine.h.
#include <vector>
namespace svs {
namespace {
struct Mt {
  int left;
  uint32_t next;
  uint32_t s[624];
  void twist() {
    auto mix = [](uint32_t u, uint32_t v) { return ((((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u)); };
    uint32_t* p = s;
    for (int j = 624 - 397 + 1; --j; p++) *p = p[397] ^ mix(p[0], p[1]);
    for (int j = 397; --j; p++) *p = p[397 - 624] ^ mix(p[0], p[1]);
    *p = p[397 - 624] ^ mix(p[0], s[0]);
    left = 624;
    next = 0;
  }
  uint32_t draw() {
    if (--left == 0) twist();
    uint32_t y = s[next++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  void discard(long long m) {
    while (m > 0) {
      if (left == 1) { twist(); ++next; --m; continue; }       // (the draw that finds left == 1 twists and takes s[0])
      const long long c = m < left - 1 ? m : left - 1;
      left -= (int)c;
      next += (uint32_t)c;
      m -= c;
    }
  }
};
}  // namespace
}  // namespace svs

extern "C" {
int svs_randperm_prefix(unsigned char* rng_state, size_t state_bytes, long long n, long long k, long long* out) {
  using svs::set_error;
  if (!rng_state || (!out && k > 0) || state_bytes != 5056) { set_error("svs_randperm_prefix: the 5056-byte state of torch's CPU generator"); return SVS_EINVAL; }
  if (n < 1 || k < 0 || k > n || n >= (long long)(0xffffffffu / 20)) { set_error("svs_randperm_prefix: 0 <= k <= n < 2^32 / 20"); return SVS_ESHAPE; }
  svs::Mt mt;
  int seeded;
  unsigned long long next64;
  memcpy(&mt.left, rng_state + 8, 4);
  memcpy(&seeded, rng_state + 12, 4);
  memcpy(&next64, rng_state + 16, 8);
  if (!seeded || mt.left <= 0 || mt.left > 624 || next64 > 624) { set_error("svs_randperm_prefix: not a seeded mt19937 state"); return SVS_EINVAL; }
  mt.next = (uint32_t)next64;
  for (int j = 0; j < 624; ++j) {
    unsigned long long v;
    memcpy(&v, rng_state + 24 + 8 * j, 8);
    mt.s[j] = (uint32_t)v;
  }
  static thread_local std::vector<long long> r;
  if ((long long)r.size() < n) {
    const size_t old = r.size();
    r.resize((size_t)n);
    for (size_t i = old; i < (size_t)n; ++i) r[i] = (long long)i;
  }
  static thread_local std::vector<long long> partner;
  const long long iters = k < n - 1 ? k : n - 1;
  partner.resize((size_t)iters);
  for (long long i = 0; i < iters; ++i) {
    const long long z = (long long)mt.draw() % (n - i);
    partner[(size_t)i] = i + z;
    const long long sav = r[(size_t)i];
    r[(size_t)i] = r[(size_t)(i + z)];
    r[(size_t)(i + z)] = sav;
  }
  for (long long i = 0; i < k; ++i) out[i] = r[(size_t)i];
  for (long long i = iters; i-- > 0;) {                       // undo: the array is the identity again
    const long long j = partner[(size_t)i], sav = r[(size_t)i];
    r[(size_t)i] = r[(size_t)j];
    r[(size_t)j] = sav;
  }
  mt.discard((n - 1) - iters);
  memcpy(rng_state + 8, &mt.left, 4);
  next64 = mt.next;
  memcpy(rng_state + 16, &next64, 8);
  for (int j = 0; j < 624; ++j) {
    const unsigned long long v = mt.s[j];
    memcpy(rng_state + 24 + 8 * j, &v, 8);
  }
  return SVS_OK;
}

int svs_version(void) { return 101; }
// deterministic accumulation of the weight gradients (csrc/svs_ticket.h): process-wide switch, off by default
int svs_set_deterministic(int on) { const int was = svs::g_deterministic.load(); svs::g_deterministic.store(on ? 1 : 0); return was; }
int svs_get_deterministic(void) { return svs::g_deterministic.load(); }
const char* svs_last_error_string(void) { return svs::g_err; }
}
