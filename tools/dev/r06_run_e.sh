#!/bin/bash
# round 6, GPU call E: the round's rocprofv3 evidence (tools/profile_round.sh r06), then the driver's bench command and the bench
# with its opt-in extras (config 4's projection, VolOpt.run, other precisions, in-line A/B)
bash tools/profile_round.sh r06 > gpurun_out/r06_profile.log 2>&1; tail -3 gpurun_out/r06_profile.log
O=gpurun_out/r06e; mkdir -p $O
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -4 $O/bench_driver_cmd.err | grep real; cp bench_extras.json $O/bench_extras_default.json
python3 bench.py --steps 200 --no-cpu-baseline --other-precisions --inline-ab --volopt-loop --config4 > $O/bench_extras_run.json 2> $O/bench_extras_run.err; cp bench_extras.json $O/bench_extras_all.json
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]); print("driver cmd:", d["ms_per_step"], d["value"], len(open("$O/bench_driver_cmd.json").read()))
e=json.load(open("$O/bench_extras_all.json")); print("200 steps:", e["ms_per_step"], e.get("fast_grad_ms_per_step"), e.get("exact_f32_ms_per_step"))
print("config4:", json.dumps(e.get("config4"))[:1500]); print("volopt:", json.dumps(e.get("volopt_run"))[:400], json.dumps(e.get("volopt_run_256_rays"))[:300])
PY
