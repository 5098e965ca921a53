"""dev aid: per-launch durations of the last FeatureNet call in a rocprofv3 --kernel-trace of tools/bench_featurenet.py
    python tools/dev/fpn_timeline.py <dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
              for r in csv.DictReader(open(f)) if "svs::conv2d" in r["Kernel_Name"])
last = rows[-13:]
t0 = last[0][0]
for s, e, n, gx, gy, wx in last:
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  {n.split('(')[0].split('::')[-1]:34s} blocks {int(gx) // int(wx)} x {gy}")
print(f"span {(last[-1][1] - t0) / 1e3:.1f} us, sum {sum(e - s for s, e, *_ in last) / 1e3:.1f} us")
