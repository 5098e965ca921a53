#!/bin/bash
# round 6, GPU call B: the gpu test suite, the weight-gradient GEMM's L2 touch A/B, 12 more paired Chamfer seeds + the
# deterministic path
O=gpurun_out/r06b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
bash tools/dev/ab_env.sh "SVS_WGRAD_TOUCH=0" "SVS_WGRAD_TOUCH=1" 3 --steps 100 > $O/ab_touch.txt 2>&1; cat $O/ab_touch.txt
bash tools/dev/ab_env.sh "SVS_WGRAD_TOUCH=0" "SVS_WGRAD_TOUCH=1" 2 --steps 100 --groups none > $O/ab_touch_onegroup.txt 2>&1; cat $O/ab_touch_onegroup.txt
for t in 0 1; do SVS_WGRAD_TOUCH=$t python bench.py --no-cpu-baseline --no-gpu-torch --no-extras --steps 50 2>/dev/null | tail -1 > $O/bench_touch$t.json; python - <<PY
import json
d=json.load(open("$O/bench_touch$t.json"))
print("touch=$t", d["ms_per_step"], [(r["kernel"], r["what"][-28:], round(r["kernel_ms"],4), round(r["frac"],3)) for r in d["roofline"]["kernels"]])
PY
done
timeout 1500 python3 tools/chamfer_parity.py --steps 3000 --seeds 12,13,14,15,16,17,18,19,20,21,22,23 --paths hip,hip_f32,torch_f32 --prior --parallel --out $O/chamfer_paired_12-23.json > $O/chamfer_paired_12-23.log 2>&1
timeout 900 python3 tools/chamfer_parity.py --steps 3000 --seeds 0,0,1,2,3,4,5,6,7,8,9,10,11 --paths hip_det --prior --parallel --out $O/chamfer_det.json > $O/chamfer_det.log 2>&1
python - <<PY
import json
for f in ("chamfer_paired_12-23","chamfer_det"):
    try:
        d=json.load(open("$O/"+f+".json"))
        for p,v in d.items():
            if isinstance(v,dict) and "runs" in v: print(f,p,[round(r.get("overall_mm",-1),3) for r in v["runs"]])
        print({k:{q:x for q,x in v.items() if q!="differences_mm"} for k,v in d.get("paired",{}).items()})
    except Exception as e: print(f, "failed", e)
PY
