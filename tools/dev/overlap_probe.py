"""Overlapped DataLoader loop (VolOpt._epoch_overlapped): how long the helper's fetch and the main thread's enqueue take
when they run concurrently, and how long the main thread waits at the join."""
import json, os, sys, tempfile, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
import torch
from loop_probe import build, timed_run


def main():
    os.chdir(tempfile.mkdtemp())
    v = build(overlap_loader=True)
    v.run(opt_stepN=60)
    T = dict(fetch=[], enq=[], join=[], step=[])
    orig_step = v.train_step
    ds = v.train_dataset
    o_change = ds.change_sampling_idx

    def change(n):
        t0 = time.perf_counter(); o_change(n); T["fetch"].append(time.perf_counter() - t0)
    ds.change_sampling_idx = change

    def step(*a, **k):
        t0 = time.perf_counter(); r = orig_step(*a, **k); T["enq"].append(time.perf_counter() - t0); return r
    v.train_step = step
    o_join = threading.Thread.join

    def join(self, *a, **k):
        t0 = time.perf_counter(); o_join(self, *a, **k); T["join"].append(time.perf_counter() - t0)
    threading.Thread.join = join
    ms = timed_run(v, 200)
    threading.Thread.join = o_join
    med = lambda x: 1e3 * sorted(x)[len(x) // 2] if x else None
    print(json.dumps(dict(overlap_ms=ms, randperm_in_helper_ms=med(T["fetch"]), train_step_host_ms=med(T["enq"]), join_wait_ms=med(T["join"]),
                          n=len(T["enq"]), switchinterval=sys.getswitchinterval())))


if __name__ == "__main__":
    main()
