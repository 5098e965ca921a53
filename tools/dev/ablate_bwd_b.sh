#!/bin/bash
# Times pass B (and pass A) with parts compiled out (SVS_ABL switches of svs_mlp_bwd_h2.hip; variants built by tools/dev/ab_defs.sh
# svs_mlp_bwd_h2.hip b<mask> -DSVS_ABL=<mask>): 4 no MFMA, 4096 no side-tile loads, 8192 no abar stores, 16384 no second-order
# term, 32768 no softplus' arithmetic.   tools/dev/ablate_bwd_b.sh <mask> ...
for rays in 256 1024; do
  for m in "$@"; do
    out=$(BK_RAYS=$rays SVS_LIB_PATH=$PWD/s-volsdf_amd/lib_ab/libsvolsdf_hip_b$m.so timeout 300 python tools/bench_kernels.py 2>/dev/null | tail -1)
    echo "rays $rays mask $m $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A', d['sdf_bwd_a']['ms'], 'B', d['sdf_bwd_b']['ms'], 'rgb_bwd', d['rgb_bwd']['ms'])")"
  done
done
