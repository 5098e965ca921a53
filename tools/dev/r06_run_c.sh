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
