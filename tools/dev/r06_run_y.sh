#!/bin/bash
# round 6, GPU call Y: what bounds the weight-gradient GEMM -- diagnostic builds (-DSVS_WGRAD_DIAG: 1 no flush, 4 fragments read but
# nothing multiplied, 8 copies + barriers only) of the launch replayed alone (tools/dev/time_wgrad.py)
O=gpurun_out/r06y; mkdir -p $O
A=$PWD/s-volsdf_amd/lib_ab
for rep in 1 2; do
  echo "== product"; python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep wgrad
  for d in 4 5 8 9; do echo "== DIAG=$d"; SVS_LIB_PATH=$A/libsvolsdf_hip_wgdiag$d.so python tools/dev/time_wgrad.py 256 1024 2>/dev/null | grep wgrad; done
done | tee $O/wgrad_diag.txt
