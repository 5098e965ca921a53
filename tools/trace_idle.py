"""GPU idle time per step from a rocprofv3 kernel trace of bench.py: the union of all kernel intervals (all streams) over the
last steps of the run, against the wall time they span.
    rocprofv3 --kernel-trace -d <dir> --output-format csv -- python3 bench.py --steps 30 --warmup 50 --settle 0 \
              --no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-kernel-timing
(--warmup 50: the auto-schedule measurement of trainer.TrainStep alternates split and whole steps during steps 24-47 of a
process; the traced steps must all run the schedule it settled on -- or pass --groups none / a fixed split)
    python tools/trace_idle.py <dir> [n_steps]
"""
import csv
import glob
import os
import sys


def main(d, n_steps=20):
    f = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    # step boundaries: the Adam launch ends a step
    ends = [e for s, e, n in rows if "adam_kernel" in n]
    if len(ends) < n_steps + 1:
        raise SystemExit("not enough steps in the trace")
    t0, t1 = ends[-n_steps - 1], ends[-1]
    iv = [(max(s, t0), min(e, t1)) for s, e, n in rows if e > t0 and s < t1]
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, cur_e))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = t1 - t0
    print(f"{n_steps} steps: {wall / n_steps / 1e6:.3f} ms/step, GPU busy {busy / wall:.3f}, idle {(wall - busy) / n_steps / 1e3:.1f} us/step, "
          f"{len(gaps) / n_steps:.1f} gaps/step")
    # which kernel follows the largest gaps
    by_next = {}
    starts = {s: n for s, e, n in rows}
    for g, at in gaps:
        nxt = min((s for s, e, n in rows if s >= at + g), default=None)
        name = starts.get(nxt, "?").split("(")[0][-40:]
        a = by_next.setdefault(name, [0, 0]); a[0] += g; a[1] += 1
    for name, (tot, cnt) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  before {name:42s} {tot / n_steps / 1e3:7.1f} us/step in {cnt / n_steps:.1f} gaps")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20)
