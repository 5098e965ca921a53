"""`VolOpt.run` end to end at small per-GPU batches (config 4's 256 rays per GPU, 128), eager launches against launch plans
(dev aid; run on the GPU box).  Every case is a child process."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = ("import sys, json; sys.path.insert(0, %r); import bench; r = bench._volopt_loop(int(sys.argv[1]), warm=80, steps=300); "
        "print(json.dumps({k: round(v['ms_per_step'], 3) for k, v in r.items() if isinstance(v, dict)}))" % ROOT)
for rays in [int(x) for x in (sys.argv[1:] or ["256", "128", "1024"])]:
    for mode in os.environ.get("LOOP_MODES", "0 auto").split():
        r = subprocess.run([sys.executable, "-c", code, str(rays)], env=dict(os.environ, SVS_TRAIN_GRAPH=mode), capture_output=True, text=True)
        print(rays, "SVS_TRAIN_GRAPH=" + mode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-1500:], flush=True)
