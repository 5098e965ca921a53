#!/bin/bash
# round 6, GPU call O: diagnostic builds of the round-6 producer (SVS_WARP_ABL: 1 no stores, 2 no re-gathers, 4 constant tables)
O=gpurun_out/r06o; mkdir -p $O
for rep in 1 2; do
  echo "== all"; python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
  for m in 1 2 4 3 5 6 7; do echo "== ABL=$m"; SVS_LIB_PATH=$PWD/s-volsdf_amd/lib_ab/libabl$m.so python tools/dev/time_warp.py 2>/dev/null | grep "split=True"; done
done 2>&1 | tee $O/warp_ablation.txt
