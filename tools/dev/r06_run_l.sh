#!/bin/bash
# round 6, GPU call L: the deterministic path on the 12 seeds of the first paired study (one run at a time), the opt-in Chamfer
# extra of bench.py, three 200-step bench runs
O=gpurun_out/r06l; mkdir -p $O
timeout 1500 python3 tools/chamfer_parity.py --steps 3000 --seeds 0,1,2,3,4,5,6,7,8,9,10,11 --paths hip_det --prior --out $O/chamfer_det12.json > $O/chamfer_det12.log 2>&1
python - <<PY
import json
try:
    d=json.load(open("$O/chamfer_det12.json")); print("hip_det", [round(r.get("overall_mm",-1),3) for r in d["hip_det"]["runs"]], "mean", round(d["hip_det"]["overall_mm"],3))
except Exception as e: print("failed", e)
PY
for i in 1 2 3; do python3 bench.py --steps 200 --no-cpu-baseline --no-gpu-torch --no-extras 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('200 steps:', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done | tee $O/bench_200x3.txt
( time python3 bench.py --steps 50 --no-cpu-baseline --no-gpu-torch --chamfer ) > $O/bench_chamfer.json 2> $O/bench_chamfer.err; grep real $O/bench_chamfer.err; python3 -c "
import json; e=json.load(open('bench_extras.json')); print(json.dumps(e.get('chamfer_parity'))[:600])"
