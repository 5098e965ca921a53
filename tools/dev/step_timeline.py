"""Timeline of ONE steady-state train step from a rocprofv3 kernel trace: start offset, duration, hardware queue of every
dispatch (dev aid; run on the GPU box).

    python tools/dev/step_timeline.py "<bench.py args>" [label]
runs `rocprofv3 --kernel-trace -- python3 bench.py <args>` as a child and prints the step that starts with the
`--anchor` kernel (default: the first rownorm_kernel of a step) about two thirds into the trace.
"""
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1].split()
label = sys.argv[2] if len(sys.argv) > 2 else "timeline"
out = os.path.join(ROOT, "gpurun_out", "timeline_" + label)
shutil.rmtree(out, ignore_errors=True)
os.makedirs(out)
env = dict(os.environ, TMPDIR="/tmp")
cmd = ["rocprofv3", "--kernel-trace", "-d", out, "--output-format", "csv", "--", "python3", os.path.join(ROOT, "bench.py"),
       "--no-cpu-baseline", "--no-exact-f32", "--no-gpu-torch", "--no-volopt-loop", "--no-extras", "--no-kernel-timing",
       "--steps", "40", "--warmup", "10", "--settle", "0.3"] + args
r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
f = glob.glob(os.path.join(out, "*", "*kernel_trace.csv"))
if not f:
    print(r.stdout[-2000:], r.stderr[-2000:])
    raise SystemExit(1)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda x: int(x["Start_Timestamp"]))
name = lambda x: x["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0][:28]
adam = [i for i, x in enumerate(rows) if "adam_kernel" in x["Kernel_Name"]]
a, b = adam[len(adam) * 2 // 3], adam[len(adam) * 2 // 3 + 1]
step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
print(f"{label}: step of {len(step)} dispatches, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us from the previous Adam's end "
      f"to this Adam's end; mean step over the trace {(int(rows[adam[-1]]['End_Timestamp']) - int(rows[adam[5]]['End_Timestamp'])) / 1e3 / (len(adam) - 6):.1f} us")
qs = sorted({x["Queue_Id"] for x in step})
busy_end = t0
for x in step:
    s, e = int(x["Start_Timestamp"]), int(x["End_Timestamp"])
    gap = (s - busy_end) / 1e3
    busy_end = max(busy_end, e)
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  q{qs.index(x['Queue_Id'])}  gap {gap:6.1f}  {name(x)}  grid {x.get('Grid_Size_X', x.get('Grid_Size', '?'))}")
shutil.rmtree(out, ignore_errors=True)
