#!/bin/bash
# kernel stats of the 1024-ray step, eager launches against the launch plan (dev aid; run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/plan_trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-exact-f32 --no-gpu-torch --no-volopt-loop --no-extras --no-kernel-timing --steps 50 --warmup 10 ${ARGS}"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/eager --output-format csv -- $B --graph off > $O/eager.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/plan --output-format csv -- $B --graph plan > $O/plan.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/plan_trace")
tab = {}
for m in ("eager", "plan"):
    f = glob.glob(f"{O}/{m}/*/*kernel_stats.csv")
    for row in csv.DictReader(open(f[0])):
        tab.setdefault(row["Name"][:70], {})[m] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6)
print(f"{'kernel':70s} {'eager calls':>11s} {'avg us':>8s} {'tot ms':>8s} | {'plan calls':>10s} {'avg us':>8s} {'tot ms':>8s}")
for k, v in sorted(tab.items(), key=lambda kv: -max(x[2] for x in kv[1].values()))[:28]:
    e, p = v.get("eager", (0, 0, 0)), v.get("plan", (0, 0, 0))
    print(f"{k:70s} {e[0]:11d} {e[1]:8.1f} {e[2]:8.1f} | {p[0]:10d} {p[1]:8.1f} {p[2]:8.1f}")
for m in ("eager", "plan"):
    print(m, open(f"{O}/{m}.log").read().strip().splitlines()[-1][:160])
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
