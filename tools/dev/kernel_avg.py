"""Average duration of the kernels whose name contains one of the given words, from a rocprofv3 --stats run of bench.py
(dev aid; run on the GPU box):  python tools/dev/kernel_avg.py "<bench args>" word [word ...]"""
import csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = "/tmp/kernel_avg_out"
shutil.rmtree(out, ignore_errors=True)
cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", out, "--output-format", "csv", "--", "python3", os.path.join(ROOT, "bench.py"),
       "--no-cpu-baseline", "--no-exact-f32", "--no-gpu-torch", "--no-volopt-loop", "--no-extras", "--no-kernel-timing",
       "--steps", "50", "--warmup", "10"] + sys.argv[1].split()
subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
for row in csv.DictReader(open(glob.glob(out + "/*/*kernel_stats.csv")[0])):
    if any(w in row["Name"] for w in sys.argv[2:]):
        print(f"{sys.argv[1]:28s} {row['Name'][:60]:60s} calls {row['Calls']:>6s}  avg {float(row['AverageNs']) / 1e3:8.1f} us")
shutil.rmtree(out, ignore_errors=True)
