#!/bin/bash
# round 6, GPU call D: the round-6 cost-volume producer (XCD-aware row mapping on / off) against round 5's kernel
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_costvol.py -x -q > $O/pytest_costvol.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_costvol.log; tail -3 $O/pytest_costvol.log
for rep in 1 2 3; do
  echo "== r6 kernel, XCD rows"; python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
  echo "== r6 kernel, launch order"; SVS_LIB_PATH=$PWD/s-volsdf_amd/lib_ab/libnoxcd.so python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
  echo "== r5 kernel"; SVS_WARP_KERNEL=5 python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
done 2>&1 | tee $O/time_warp_xcd.txt
for k in "" 5; do SVS_WARP_KERNEL=$k python tools/bench_costvol.py 2>/dev/null | tail -1 > $O/costvol_k${k:-4}.json; done
SVS_LIB_PATH=$PWD/s-volsdf_amd/lib_ab/libnoxcd.so python tools/bench_costvol.py 2>/dev/null | tail -1 > $O/costvol_knoxcd.json
python - <<PY
import json
for n in ("4","noxcd","5"):
    d=json.load(open("$O/costvol_k%s.json"%n)); print(n, {k: round(v,4) for k,v in d.items() if isinstance(v,float)})
PY
