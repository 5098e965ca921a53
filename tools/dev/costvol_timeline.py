"""dev aid: kernel timeline of the last stage-1 pass in a rocprofv3 --kernel-trace of tools/bench_costvol.py
    python tools/dev/costvol_timeline.py <dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
# the last stage-1 pass: from the last depth_hypotheses launch that precedes a warp_variance_kernel<32 to the prob_depth_conf after it
idx32 = [i for i, r in enumerate(rows) if "warp_variance_kernel<32" in r[2]]
i_w = idx32[-2]                      # the model's call (the last one is the bench's standalone call)
i0 = max(i for i in range(i_w) if "depth_hypotheses" in rows[i][2])
i1 = min(i for i in range(i_w, len(rows)) if "prob_depth_conf" in rows[i][2])
t0 = rows[i0][0]
busy = 0
prev_end = t0
for s, e, n in rows[i0:i1 + 1]:
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {n.split('(')[0][-60:]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(rows[i1][1] - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, {i1 - i0 + 1} launches")
