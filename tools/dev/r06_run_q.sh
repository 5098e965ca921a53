#!/bin/bash
# round 6, GPU call Q: cache hints on the producer's stores (nt / sc0 sc1 / sc1 nt)
O=gpurun_out/r06q; mkdir -p $O
for rep in 1 2; do
  echo "== default"; python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
  for m in 1 2 3; do echo "== hint $m"; SVS_LIB_PATH=$PWD/s-volsdf_amd/lib_ab/libnt$m.so python tools/dev/time_warp.py 2>/dev/null | grep "split=True"; done
done 2>&1 | tee $O/warp_store_hints.txt
