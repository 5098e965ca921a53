"""HIP runtime calls the host makes for one steady-state train step, in order, with counts (dev aid; run on the GPU box):
    python tools/dev/api_between_steps.py "<bench args>" """
import csv, glob, os, shutil, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = "/tmp/api_trace_out"
shutil.rmtree(out, ignore_errors=True)
cmd = ["rocprofv3", "--hip-runtime-trace", "-d", out, "--output-format", "csv", "--", "python3", os.path.join(ROOT, "bench.py"),
       "--no-cpu-baseline", "--no-exact-f32", "--no-gpu-torch", "--no-volopt-loop", "--no-extras", "--no-kernel-timing",
       "--steps", "30", "--warmup", "10", "--settle", "0.3"] + sys.argv[1].split()
subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
f = glob.glob(out + "/*/*hip_api_trace.csv")
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Function"] for r in rows]
# a step = from one hipLaunchKernel burst to the next: split at the copies that do not belong to kernels
idx = [i for i, n in enumerate(names) if n.startswith("hipMemcpy") or n.startswith("hipMemset")]
tail = names[len(names) * 2 // 3:]
print(collections.Counter(tail).most_common(12))
last = [i for i in idx if i > len(names) * 2 // 3][:12]
for i in last[:2]:
    print(i, [n.replace("hip", "") for n in names[i - 14:i + 12] if n not in ("hipGetDevice", "hipSetDevice", "hipGetLastError", "hipDevicePrimaryCtxGetState")])
    print("   args:", {k: v for k, v in rows[i].items() if k not in ("Function",)})
shutil.rmtree(out, ignore_errors=True)
