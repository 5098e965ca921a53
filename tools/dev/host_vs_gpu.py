"""Step time against the time the host needs to ENQUEUE a step, for eager launches and for launch plans (dev aid; run on
the GPU box).  Every case is a `bench.py` child process; this process never touches the GPU.

    python tools/dev/host_vs_gpu.py ["--rays 256" "--rays 256 --model bmvs" ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = sys.argv[1:] or ["--rays 256", "--rays 256 --model bmvs", "--rays 128", "--rays 128 --model bmvs", "--rays 1024",
                         "--rays 2048 --model bmvs"]
BASE = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-exact-f32", "--no-gpu-torch",
        "--no-kernel-timing", "--no-volopt-loop", "--no-extras", "--steps", "100"]
for case in CASES:
    for g in os.environ.get("MODES", "off plan").split():
        r = subprocess.run(BASE + case.split() + ["--graph", g], capture_output=True, text=True, timeout=600)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{case:28s} {g:5s} {d['ms_per_step']:.3f} ms/step  host {d['host_enqueue_ms_per_step']:.3f}  "
                  f"{d['config']['ray_groups']} {d['config'].get('launch_plan')}", flush=True)
        except Exception as e:
            print(case, g, "FAILED", repr(e), r.stderr[-3000:], flush=True)
