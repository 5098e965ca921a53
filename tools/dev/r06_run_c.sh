#!/bin/bash
# round 6, GPU call C: the eight-channels-per-lane cost-volume producer: parity tests, then timings against the round-5 kernel
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_costvol.py -x -q > $O/pytest_costvol.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_costvol.log; tail -5 $O/pytest_costvol.log
for rep in 1 2; do
  echo "== new (default lib)"; python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
  echo "== old (SVS_WARP_KERNEL=4)"; SVS_WARP_KERNEL=4 python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
  for v in minb2 plain; do echo "== new, $v"; SVS_LIB_PATH=$PWD/s-volsdf_amd/lib_ab/lib$v.so python tools/dev/time_warp.py 2>/dev/null | grep "split=True"; done
done 2>&1 | tee $O/time_warp.txt
python tools/bench_costvol.py 2>/dev/null | tail -1 > $O/costvol_new.json; SVS_WARP_KERNEL=4 python tools/bench_costvol.py 2>/dev/null | tail -1 > $O/costvol_old.json
python - <<PY
import json
for n in ("new","old"):
    d=json.load(open("$O/costvol_%s.json"%n)); print(n, {k: round(v,4) for k,v in d.items() if isinstance(v,float)})
PY
# the eval render (the sdf_full render instance no longer spills) and the deterministic Chamfer path, one run at a time
python tools/bench_render_eval.py 2>/dev/null | tail -1 | cut -c1-400 | tee $O/render_eval.json
timeout 1200 python3 tools/chamfer_parity.py --steps 3000 --seeds 0,0,1,2,3 --paths hip_det --prior --out $O/chamfer_det.json > $O/chamfer_det.log 2>&1
python - <<PY
import json
try:
    d=json.load(open("$O/chamfer_det.json")); print("hip_det", [(r["seed"], r.get("overall_mm"), r.get("train_s")) for r in d["hip_det"]["runs"]])
except Exception as e: print("chamfer_det failed", e)
PY
