#!/bin/bash
# round 6, GPU call P: the producer with line-aligned stores (timing experiment: split-volume rows padded by 8 units)
O=gpurun_out/r06p; mkdir -p $O
for rep in 1 2 3; do
  echo "== pad 1 (the layout the readers use)"; python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
  echo "== pad 8 (line-aligned runs)"; SVS_LIB_PATH=$PWD/s-volsdf_amd/lib_ab/libpad8.so python tools/dev/time_warp.py 2>/dev/null | grep "split=True"
done 2>&1 | tee $O/warp_pad8.txt
