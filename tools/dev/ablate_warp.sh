#!/bin/bash
# Diagnostic builds of the cost-volume producer (csrc/svs_costvol.hip, -DSVS_WARP_ABL=<mask>) as libraries of their own under
# s-volsdf_amd/lib_ab/, then (on the GPU box) tools/bench_costvol.py with each:   tools/dev/ablate_warp.sh build | run
cd "$(dirname "$0")/../.." || exit 1
L=s-volsdf_amd/lib; A=s-volsdf_amd/lib_ab
if [ "$1" = build ]; then
  mkdir -p $A
  for m in ${MASKS:-1 2 4 7} ${EXTRA_MASKS}; do
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -x hip -ffp-contract=off -DSVS_WARP_ABL=$m ${EXTRA_DEFS} \
      -I s-volsdf_amd/csrc -c s-volsdf_amd/csrc/svs_costvol.hip -o $A/costvol_abl$m.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $A/libabl$m.so $(ls $L/*.o | grep -v svs_costvol.o) $A/costvol_abl$m.o
  done
  ls -la $A/*.so
else
  for m in "" ${MASKS:-1 2 4 7} ${EXTRA_MASKS}; do
    lib=${m:+$PWD/$A/libabl$m.so}
    echo -n "SVS_WARP_ABL=${m:-0}: "
    SVS_LIB_PATH=$lib python tools/bench_costvol.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: round(d[k],4) for k in ('stage1_ms','stage1_warp_ms','stage2_warp_ms','stage3_warp_ms')})"
  done
fi
