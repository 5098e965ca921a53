// Deterministic accumulation (svs_set_deterministic(1), SVS_DETERMINISTIC=1 on the Python side).
//
// The weight-gradient kernels reduce over the points in two levels: a workgroup sums its share of the point tiles in
// registers, then adds its partial sums into the accumulator with float atomics.  The order in which the workgroups' atomics
// arrive differs from run to run, so a step's gradient differs in its last bits from run to run, and an optimisation of a
// few thousand Adam steps amplifies that into visibly different runs.  In deterministic mode the
// workgroups that add into one accumulator take TURNS: each holds a ticket (its rank among those workgroups, in launch
// order), waits until the accumulator's turn counter shows its ticket, adds, and passes the turn on; the last one resets the
// counter.  The sum is then formed in one fixed order, ((0 + p_0) + p_1) + ..., and a step is a function of its inputs.
// A workgroup only ever waits for workgroups with LOWER block indices of its own launch, which the dispatcher has started
// before it: no deadlock.  Slower (the flushes of a layer are serialised: +10 ... 20 % per step), which is why it is a mode.
//
// Turn counters live in a static device array of the launching translation unit (no allocation behind the C-ABI); a launch
// takes the next `n` of them round-robin, so launches in flight on different streams do not share counters unless more than
// kSlots / n launches of that translation unit are in flight.  Launches that add into the SAME accumulator must be stream-ordered by the caller
// (the Python side runs a deterministic step on one stream).
#pragma once
#include "svs_common.h"
#include <mutex>

namespace svs {
bool deterministic();            // host: svs_set_deterministic (svs_common.cpp)

namespace det {
constexpr unsigned kSlots = 1024;
static __device__ unsigned g_turn[kSlots];

struct Ticket {
  unsigned* turn;      // the accumulator's turn counter, or nullptr (atomics in arrival order)
  unsigned first;      // ticket of workgroup 0 of this job
  unsigned total;      // tickets handed out for this counter in this launch
};

// all threads of the workgroup call both; uniform branch on t.turn
__device__ __forceinline__ void wait_turn(const Ticket& t, unsigned rank) {
  if (!t.turn) return;
  if (threadIdx.x == 0) {
    const unsigned mine = t.first + rank;
    while (__hip_atomic_load(t.turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != mine) __builtin_amdgcn_s_sleep(4);
  }
  __syncthreads();
}
__device__ __forceinline__ void pass_turn(const Ticket& t, unsigned rank) {
  if (!t.turn) return;
  __threadfence();               // this thread's atomics have been performed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned mine = t.first + rank;
    __hip_atomic_store(t.turn, mine + 1 == t.total ? 0u : mine + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// host: `n` counters for one launch (nullptr when the mode is off).  Internal linkage like g_turn itself: every translation unit
// that launches ticketed kernels has its own counter array and its own cursor into it.
static inline unsigned* take_slots(unsigned n) {
  if (!deterministic() || n == 0 || n > kSlots) return nullptr;
  static std::mutex mu;
  static unsigned* base = nullptr;
  static unsigned next = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (!base) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_turn)) != hipSuccess) return nullptr;
    base = static_cast<unsigned*>(p);
  }
  if (next + n > kSlots) next = 0;             // no wrap inside a launch
  unsigned* out = base + next;
  next += n;
  return out;
}
}  // namespace det
}  // namespace svs
