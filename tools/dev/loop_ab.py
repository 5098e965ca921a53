"""A/B of the overlapped loop's two forms on one box: DataLoader next() in the helper thread vs in the main thread."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
import torch
from loop_probe import build, timed_run
os.chdir(tempfile.mkdtemp())
res = {}
for rep in range(2):
    for mode in ("helper", "main", "sequential"):
        if mode == "sequential":
            v = build(overlap_loader=False)
        else:
            os.environ["SVS_OVERLAP_NEXT"] = mode
            v = build()
        v.run(opt_stepN=60)
        res.setdefault(mode, []).append(round(timed_run(v, 200), 3))
        del v
print(json.dumps(res))
