#!/bin/bash
# round 6, GPU call X: the final check of HEAD -- the whole gpu suite, smoke, the driver's bench command, the bench with its opt-in extras
O=gpurun_out/r06x; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep real $O/bench_driver_cmd.err; wc -c $O/bench_driver_cmd.json; cp bench_extras.json $O/bench_extras_default.json
( time python3 bench.py --steps 200 --no-cpu-baseline --other-precisions --inline-ab --volopt-loop --config4 ) > $O/bench_extras_run.json 2> $O/bench_extras_run.err; grep real $O/bench_extras_run.err; cp bench_extras.json $O/bench_extras_all.json
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]); print("driver cmd:", d["ms_per_step"], d["value"], d.get("self_test_failed"))
e=json.load(open("$O/bench_extras_all.json")); print("200 steps:", e["ms_per_step"], e["value"], e.get("fast_grad_ms_per_step"), e.get("exact_f32_ms_per_step"))
print("config4:", json.dumps(e.get("config4"))[:700])
PY
