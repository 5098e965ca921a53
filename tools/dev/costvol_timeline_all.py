"""dev aid: kernel timeline of the last three-stage forward in a rocprofv3 --kernel-trace of tools/bench_costvol.py
    python tools/dev/costvol_timeline_all.py <dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
pdc = [i for i, r in enumerate(rows) if "prob_depth_conf" in r[2]]
i1 = pdc[-1]
i0 = pdc[-4] + 1 if len(pdc) >= 4 else 0
t0 = rows[i0][0]
prev_end = t0
stage_start = t0
for s, e, n in rows[i0:i1 + 1]:
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {n.split('(')[0][-64:]}")
    prev_end = max(prev_end, e)
    if "prob_depth_conf" in n:
        print(f"   ---- stage: {(e - stage_start) / 1e3:.1f} us")
        stage_start = e
