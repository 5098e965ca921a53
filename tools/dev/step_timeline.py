"""dev aid: kernel timeline of the last train step in a rocprofv3 --kernel-trace of bench.py (per queue)
    python tools/dev/step_timeline.py <dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))
              for r in csv.DictReader(open(f)))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
i0, i1 = ends[-2] + 1, ends[-1]
t0 = rows[i0][0]
qs = sorted({r[3] for r in rows[i0:i1 + 1]})
busy_any = 0
cur = None
for s, e, n, q, st in rows[i0:i1 + 1]:
    short = n.split("(")[0].split("::")[-1][:34]
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  q{qs.index(q)} {' ' * (4 * qs.index(q))}{short}")
iv = sorted((s, e) for s, e, *_ in rows[i0:i1 + 1])
cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy_any += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy_any += ce - cs
print(f"step span {(rows[i1][1] - t0) / 1e3:.1f} us, GPU busy (any queue) {busy_any / 1e3:.1f} us, sum of kernel times {sum(e - s for s, e, *_ in rows[i0:i1 + 1]) / 1e3:.1f} us")
