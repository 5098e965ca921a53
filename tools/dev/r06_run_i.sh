#!/bin/bash
# round 6, GPU call I: the producer's stores as one 16-byte store per lane (lane pairs swap halves) against two 8-byte stores
O=gpurun_out/r06i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_costvol.py -x -q -k "warp or three_stage or tail or homo" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for rep in 1 2 3; do
  echo "== 16-byte stores"; python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
  echo "== two 8-byte stores"; SVS_LIB_PATH=$PWD/s-volsdf_amd/lib_ab/libstore64.so python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
done 2>&1 | tee $O/time_warp_store.txt
