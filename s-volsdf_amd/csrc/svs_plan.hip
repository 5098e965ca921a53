// Launch plans: the device part of a train step enqueued by ONE library call.
//
// The step's launch sequence contains no host decision (svs_hip/trainer.py::_device_step), so it can be recorded once per
// configuration.  The recording is a stream capture (a hipGraph); replaying that graph through hipGraphLaunch is slower than
// the eager launches on this ROCm stack, so the graph is only READ here: svs_plan_build walks its nodes
// and edges and lays them out as a plan -- the nodes in a topological order, every node on one of a few HIP streams (chains
// of the dependency graph), an event for every edge that crosses streams -- and svs_plan_run enqueues that plan with plain
// hipLaunchKernel / hipMemcpyAsync / hipMemsetAsync calls: the same launches, streams and dependencies the Python
// orchestration makes (≈ 100 per step), from a C++ loop that holds no interpreter lock.
//
// The kernel arguments of a node stay where the graph keeps them: the caller keeps the graph alive as long as the plan.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <queue>
#include <unordered_map>
#include <vector>

#include "svs_common.h"

namespace svs {
namespace plan {

constexpr int kMaxStreams = 12;

struct Node {
  hipGraphNodeType type;
  hipKernelNodeParams k;
  hipMemsetParams m;            // (copy nodes are refused at build time: no parameters of theirs are kept)
  int stream = 0;
  int record = -1;              // event recorded behind this node (it has a successor on another stream)
  std::vector<int> wait;        // events this node's stream waits for in front of it
  bool via_module = false;      // k.func is a hipFunction_t (launch through hipModuleLaunchKernel)
};

struct Plan {
  std::vector<Node> nodes;      // issue order (topological)
  std::vector<hipStream_t> streams;   // [0] is the caller's stream at run time
  std::vector<bool> owned;            // created here (the others were handed in by the caller)
  std::vector<hipEvent_t> events;
  std::vector<int> entry_waiters;     // streams != 0 that start with a root node: they wait for `entry`
  hipEvent_t entry = nullptr;
  std::vector<hipEvent_t> tails;      // per stream != 0: recorded behind its last node, stream 0 waits for it
  int n_kernels = 0, n_memsets = 0, n_empty = 0;
  long long runs = 0;
};

#define PLAN_CHECK(expr)                                                                                 \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      set_error("svs_plan: %s -> %s", #expr, hipGetErrorString(e_));                                     \
      return (int)e_;                                                                                    \
    }                                                                                                    \
  } while (0)

static void destroy(Plan* p) {
  if (!p) return;
  for (size_t s = 1; s < p->streams.size(); ++s)
    if (p->streams[s] && s < p->owned.size() && p->owned[s]) (void)hipStreamDestroy(p->streams[s]);
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->tails)
    if (e) (void)hipEventDestroy(e);
  if (p->entry) (void)hipEventDestroy(p->entry);
  delete p;
}

static int build(hipGraph_t graph, const hipStream_t* given, int n_given, Plan** out) {
  size_t n = 0, ne = 0;
  PLAN_CHECK(hipGraphGetNodes(graph, nullptr, &n));
  if (n == 0) return SVS_EINVAL;
  std::vector<hipGraphNode_t> handles(n);
  PLAN_CHECK(hipGraphGetNodes(graph, handles.data(), &n));
  PLAN_CHECK(hipGraphGetEdges(graph, nullptr, nullptr, &ne));
  std::vector<hipGraphNode_t> from(ne), to(ne);
  if (ne) PLAN_CHECK(hipGraphGetEdges(graph, from.data(), to.data(), &ne));
  std::unordered_map<hipGraphNode_t, int> index;
  for (size_t i = 0; i < n; ++i) index[handles[i]] = (int)i;
  std::vector<std::vector<int>> preds(n), succs(n);
  for (size_t e = 0; e < ne; ++e) {
    auto a = index.find(from[e]), b = index.find(to[e]);
    if (a == index.end() || b == index.end()) return SVS_EINVAL;
    preds[b->second].push_back(a->second);
    succs[a->second].push_back(b->second);
  }
  // topological order, ties in the order the capture created the nodes (= the order the host issued them)
  std::vector<int> indeg(n), order;
  std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
  for (size_t i = 0; i < n; ++i) {
    indeg[i] = (int)preds[i].size();
    if (!indeg[i]) ready.push((int)i);
  }
  while (!ready.empty()) {
    const int i = ready.top();
    ready.pop();
    order.push_back(i);
    for (int s : succs[i])
      if (--indeg[s] == 0) ready.push(s);
  }
  if (order.size() != n) return SVS_EINVAL;      // a cycle: not a capture

  // longest way from the start to a node and from a node to the end.  A node wants the stream of its deepest
  // predecessor (the end of the longest chain that leads to it); of the successors that want a node's stream, the one
  // with the longest way to the end of the sequence gets it (ties: the one issued first), the others fork off.  The step's
  // critical chain therefore stays on one stream from its first to its last launch -- a dependency that crosses streams
  // costs 10-25 us on this runtime -- and the side branches (weight packing, the radiance weight gradients, a second ray
  // group, the background networks) each keep a stream of their own from fork to join.
  // "Long" is an ESTIMATE OF TIME since round 5, not a number of nodes: the weight-packing chain of the fg + background model
  // is 16 launches of 5-12 us, more nodes than the whole forward has -- counted in nodes it was the "deepest" predecessor
  // of the compositing backward, the critical chain hopped onto its stream there and back at the end (13 + 12 us of
  // crossings per 256-ray step).  A node costs max(one small launch, waves x weight): the fused sweeps and the weight-
  // gradient GEMM (one wave per SIMD walking its points through all layers; recognised by their >= 48 KiB of dynamic LDS)
  // ~0.25 us per wave, everything else ~5 ns per wave, a launch at least ~5 us.
  std::vector<long long> depth(n, 0), height(n, 0), cost(n, 470);
  std::vector<int> wanted(n, -1), heir(n, -1);
  // (second criterion where two ways have the same number of nodes: the threads launched along them -- the fg sweep
  // against the background network's, which join at compositing after two launches each)
  std::vector<double> threads(n, 0.0), below(n, 0.0), above(n, 0.0);
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    hipKernelNodeParams k;
    if (hipGraphNodeGetType(handles[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel &&
        hipGraphKernelNodeGetParams(handles[i], &k) == hipSuccess) {
      threads[i] = (double)k.gridDim.x * k.gridDim.y * k.gridDim.z * k.blockDim.x * k.blockDim.y * k.blockDim.z;
      const double waves = threads[i] / 64.0;
      cost[i] = std::max(470LL, (long long)(waves * (k.sharedMemBytes >= 48 * 1024 ? 24.0 : 0.5)));
    }
  }
  for (size_t q = 0; q < n; ++q) {
    const int i = order[q];
    for (int pr : preds[i]) {
      const int w = wanted[i];
      const double via = above[pr] + threads[pr];
      if (w < 0 || depth[pr] + cost[pr] > depth[w] + cost[w] ||
          (depth[pr] + cost[pr] == depth[w] + cost[w] && (via > above[i] || (via == above[i] && pr < w)))) {
        wanted[i] = pr;
        above[i] = via;
      }
      depth[i] = std::max(depth[i], depth[pr] + cost[pr]);
    }
  }
  for (size_t q = n; q-- > 0;) {
    const int i = order[q];
    for (int su : succs[i])
      if (height[su] + cost[su] > height[i] || (height[su] + cost[su] == height[i] && below[su] + threads[su] > below[i])) {
        height[i] = height[su] + cost[su];
        below[i] = below[su] + threads[su];
      }
    for (int su : succs[i]) {
      if (wanted[su] != i) continue;
      const int h = heir[i];
      if (h < 0 || height[su] + cost[su] > height[h] + cost[h] ||
          (height[su] + cost[su] == height[h] + cost[h] && (below[su] + threads[su] > below[h] + threads[h] ||
                                       (below[su] + threads[su] == below[h] + threads[h] && su < h))))
        heir[i] = su;
    }
  }
  // ancestors[i]: the nodes node i (transitively) depends on, as a bit set
  const size_t words = (n + 63) / 64;
  std::vector<uint64_t> anc(n * words, 0);
  for (size_t q = 0; q < n; ++q) {
    const int i = order[q];
    for (int pr : preds[i]) {
      for (size_t w = 0; w < words; ++w) anc[i * words + w] |= anc[pr * words + w];
      anc[i * words + pr / 64] |= 1ull << (pr % 64);
    }
  }
  auto depends_on = [&](int i, int a) { return (anc[(size_t)i * words + a / 64] >> (a % 64)) & 1ull; };

  Plan* p = new Plan();
  std::vector<int> stream_of(n, -1), pos(n, -1);
  std::vector<int> tail;                          // per stream: graph index of its last node
  p->nodes.resize(n);
  for (size_t q = 0; q < n; ++q) {
    const int i = order[q];
    pos[i] = (int)q;
    Node& nd = p->nodes[q];
    if (hipGraphNodeGetType(handles[i], &nd.type) != hipSuccess) { destroy(p); return SVS_ESHAPE; }
    hipError_t e = hipSuccess;
    switch (nd.type) {
      case hipGraphNodeTypeKernel:
        e = hipGraphKernelNodeGetParams(handles[i], &nd.k);
        if (e == hipSuccess && !nd.k.func) e = hipErrorInvalidValue;
        ++p->n_kernels;
        break;
      case hipGraphNodeTypeMemcpy:
        // hipGraphMemcpyNodeGetParams returns an unfilled structure for the 1-D copy nodes a captured hipMemcpyAsync
        // becomes (ROCm 7.0): such a node cannot be replayed from here.  The step contains none (copies that are part of
        // it are kernels); one that appears is an error of the caller's sequence, reported rather than guessed at.
        set_error("svs_plan: node %d is a copy node (a captured hipMemcpyAsync); its parameters are not readable -- the "
                  "sequence must do its device copies in kernels", i);
        destroy(p);
        return SVS_EINVAL;
      case hipGraphNodeTypeMemset: e = hipGraphMemsetNodeGetParams(handles[i], &nd.m); ++p->n_memsets; break;
      case hipGraphNodeTypeEmpty: ++p->n_empty; break;
      default:
        set_error("svs_plan: node type %d is not replayable", (int)nd.type);
        destroy(p);
        return SVS_EINVAL;
    }
    if (e != hipSuccess) {
      set_error("svs_plan: reading node %d (type %d): %s", i, (int)nd.type, hipGetErrorString(e));
      destroy(p);
      return SVS_ESHAPE;
    }
    // the stream: that of the predecessor it wants, if it is that node's heir; else a stream of its own; else -- all
    // streams taken -- the stream of its first predecessor
    int s = -1;
    if (wanted[i] >= 0 && heir[wanted[i]] == i && tail[stream_of[wanted[i]]] == wanted[i]) s = stream_of[wanted[i]];
    if (s < 0) {
      // a branch that forks off: a side stream whose last node this one depends on anyway is free for it (no ordering is
      // added) -- the radiance weight gradients run where the weight packing ran -- so the step gets by with few streams:
      // the runtime maps streams onto 4 hardware queues, and two busy chains that share a queue run one after the other
      // (not one whose last node still waits for its heir)
      for (size_t t = 1; t < tail.size() && s < 0; ++t)
        if (tail[t] >= 0 && depends_on(i, tail[t]) && !(heir[tail[t]] >= 0 && stream_of[heir[tail[t]]] < 0)) s = (int)t;
    }
    if (s < 0) {
      if (tail.empty() || (int)tail.size() < kMaxStreams) {
        s = (int)tail.size();
        tail.push_back(-1);
        if (preds[i].empty() && s != 0) p->entry_waiters.push_back(s);
      } else {
        s = preds[i].empty() ? 0 : stream_of[preds[i][0]];
      }
    }
    stream_of[i] = s;
    tail[s] = i;
    nd.stream = s;
  }
  // events for the edges that cross streams
  for (size_t q = 0; q < n; ++q) {
    const int i = order[q];
    Node& nd = p->nodes[q];
    // of several predecessors on one other stream only the youngest needs waiting for
    std::unordered_map<int, int> youngest;
    for (int pr : preds[i]) {
      const int s = stream_of[pr];
      if (s == nd.stream) continue;
      auto it = youngest.find(s);
      if (it == youngest.end() || pos[pr] > pos[it->second]) youngest[s] = pr;
    }
    for (auto& kv : youngest) {
      Node& src = p->nodes[pos[kv.second]];
      if (src.record < 0) {
        src.record = (int)p->events.size();
        p->events.push_back(nullptr);
      }
      nd.wait.push_back(src.record);
    }
    std::sort(nd.wait.begin(), nd.wait.end());
  }
  p->streams.assign(tail.size(), nullptr);
  p->owned.assign(tail.size(), false);
  for (size_t s = 1; s < p->streams.size(); ++s) {
    if ((int)s - 1 < n_given && given[s - 1]) { p->streams[s] = given[s - 1]; continue; }
    if (hipStreamCreateWithFlags(&p->streams[s], hipStreamNonBlocking) != hipSuccess) { destroy(p); return SVS_ESHAPE; }
    p->owned[s] = true;
  }
  for (auto& ev : p->events)
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { destroy(p); return SVS_ESHAPE; }
  // stream 0 waits at the end for the side streams that END the sequence somewhere (their last node has no successor);
  // a side stream whose last node has successors was joined where they run
  p->tails.assign(p->streams.size(), nullptr);
  for (size_t s = 1; s < p->streams.size(); ++s)
    if (!succs[tail[s]].empty()) continue;
    else if (hipEventCreateWithFlags(&p->tails[s], hipEventDisableTiming) != hipSuccess) { destroy(p); return SVS_ESHAPE; }
  if (hipEventCreateWithFlags(&p->entry, hipEventDisableTiming) != hipSuccess) { destroy(p); return SVS_ESHAPE; }
  *out = p;
  return SVS_OK;
}

static int run(Plan* p, hipStream_t s0) {
  p->streams[0] = s0;
  if (!p->entry_waiters.empty()) {
    PLAN_CHECK(hipEventRecord(p->entry, s0));
    for (int s : p->entry_waiters) PLAN_CHECK(hipStreamWaitEvent(p->streams[s], p->entry, 0));
  }
  for (Node& nd : p->nodes) {
    hipStream_t st = p->streams[nd.stream];
    for (int e : nd.wait) PLAN_CHECK(hipStreamWaitEvent(st, p->events[e], 0));
    switch (nd.type) {
      case hipGraphNodeTypeKernel: {
        const hipKernelNodeParams& k = nd.k;
        hipError_t e = hipErrorInvalidDeviceFunction;
        if (!nd.via_module) {
          e = hipLaunchKernel(k.func, k.gridDim, k.blockDim, k.kernelParams, k.sharedMemBytes, st);
          // a node captured from a module launch carries a hipFunction_t, which hipLaunchKernel does not know: only THAT
          // error switches the node over (any other -- a bad launch configuration -- is the caller's to see)
          if ((e == hipErrorInvalidDeviceFunction || e == hipErrorInvalidSymbol || e == hipErrorNotFound ||
               e == hipErrorInvalidResourceHandle) && p->runs == 0) {
            (void)hipGetLastError();
            nd.via_module = true;
          }
        }
        if (nd.via_module)
          e = hipModuleLaunchKernel((hipFunction_t)k.func, k.gridDim.x, k.gridDim.y, k.gridDim.z, k.blockDim.x, k.blockDim.y,
                                    k.blockDim.z, k.sharedMemBytes, st, k.kernelParams, k.kernelParams ? nullptr : k.extra);
        PLAN_CHECK(e);
        break;
      }
      case hipGraphNodeTypeMemset: {
        const hipMemsetParams& m = nd.m;
        if (m.height <= 1) {
          if (m.elementSize == 4) PLAN_CHECK(hipMemsetD32Async((hipDeviceptr_t)m.dst, (int)m.value, m.width, st));
          else if (m.elementSize == 2) PLAN_CHECK(hipMemsetD16Async((hipDeviceptr_t)m.dst, (unsigned short)m.value, m.width, st));
          else PLAN_CHECK(hipMemsetAsync(m.dst, (int)m.value, m.width, st));
        } else {
          PLAN_CHECK(hipMemset2DAsync(m.dst, m.pitch, (int)m.value, m.width * m.elementSize, m.height, st));
        }
        break;
      }
      default: break;      // empty node: a join point, carried by the waits / the record around it
    }
    if (nd.record >= 0) PLAN_CHECK(hipEventRecord(p->events[nd.record], st));
  }
  for (size_t s = 1; s < p->streams.size(); ++s) {
    if (!p->tails[s]) continue;
    PLAN_CHECK(hipEventRecord(p->tails[s], p->streams[s]));
    PLAN_CHECK(hipStreamWaitEvent(s0, p->tails[s], 0));
  }
  ++p->runs;
  return SVS_OK;
}

}  // namespace plan
}  // namespace svs

extern "C" {

int svs_plan_build(void* hip_graph, void* const* side_streams, int n_side_streams, void** plan_out) {
  if (!hip_graph || !plan_out || n_side_streams < 0 || (n_side_streams && !side_streams)) return SVS_EINVAL;
  svs::plan::Plan* p = nullptr;
  const int rc = svs::plan::build((hipGraph_t)hip_graph, (const hipStream_t*)side_streams, n_side_streams, &p);
  *plan_out = rc == SVS_OK ? p : nullptr;
  return rc;
}

int svs_plan_run(void* plan, void* stream) {
  if (!plan) return SVS_EINVAL;
  return svs::plan::run((svs::plan::Plan*)plan, (hipStream_t)stream);
}

int svs_plan_info(void* plan, int* counts) {
  if (!plan || !counts) return SVS_EINVAL;
  const svs::plan::Plan* p = (const svs::plan::Plan*)plan;
  counts[0] = (int)p->nodes.size();
  counts[1] = p->n_kernels;
  counts[2] = 0;                 // copy nodes: always 0 (svs_plan_build refuses them)
  counts[3] = p->n_memsets;
  counts[4] = p->n_empty;
  counts[5] = (int)p->streams.size();
  counts[6] = (int)p->events.size();
  counts[7] = (int)p->entry_waiters.size();
  return SVS_OK;
}

int svs_plan_describe(void* plan, char* text, size_t capacity) {
  if (!plan || !text || capacity < 2) return SVS_EINVAL;
  const svs::plan::Plan* p = (const svs::plan::Plan*)plan;
  size_t used = 0;
  text[0] = 0;
  for (size_t q = 0; q < p->nodes.size(); ++q) {
    const svs::plan::Node& nd = p->nodes[q];
    char line[768];
    int len = 0;
    switch (nd.type) {
      case hipGraphNodeTypeKernel: {
        const char* name = hipKernelNameRefByPtr(nd.k.func, nullptr);
        len = std::snprintf(line, sizeof line, "%zu s%d kernel %.480s grid %u,%u,%u block %u lds %u", q, nd.stream,
                            name ? name : "?", nd.k.gridDim.x, nd.k.gridDim.y, nd.k.gridDim.z, nd.k.blockDim.x,
                            nd.k.sharedMemBytes);
        break;
      }
      case hipGraphNodeTypeMemset:
        len = std::snprintf(line, sizeof line, "%zu s%d memset dst %p value %u elem %u width %zu height %zu", q, nd.stream,
                            nd.m.dst, nd.m.value, nd.m.elementSize, nd.m.width, nd.m.height);
        break;
      default: len = std::snprintf(line, sizeof line, "%zu s%d empty", q, nd.stream); break;
    }
    if (len < 0) return SVS_ESHAPE;
    len = std::min(len, (int)sizeof line - 64);
    for (int e : nd.wait) len += std::snprintf(line + len, sizeof line - len, " w%d", e);
    if (nd.record >= 0) len += std::snprintf(line + len, sizeof line - len, " r%d", nd.record);
    if (used + (size_t)len + 2 > capacity) break;       // truncated listing
    std::memcpy(text + used, line, (size_t)len);
    used += (size_t)len;
    text[used++] = '\n';
    text[used] = 0;
  }
  return SVS_OK;
}

int svs_plan_destroy(void* plan) {
  if (!plan) return SVS_EINVAL;
  svs::plan::destroy((svs::plan::Plan*)plan);
  return SVS_OK;
}

}  // extern "C"
