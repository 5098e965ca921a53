#!/bin/bash
# tools/dev/ab_units.sh <name> "<unit> <unit> ..." [-Dswitch ...] -> s-volsdf_amd/lib_ab/libsvolsdf_hip_<name>.so
# (dev aid: a variant library with several translation units rebuilt under extra switches; SVS_LIB_PATH selects it)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; UNITS=$2; shift 2
mkdir -p $R/s-volsdf_amd/lib_ab /tmp/ab_$NAME
OBJS=$(ls $R/s-volsdf_amd/lib/*.o)
for U in $UNITS; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -x hip "$@" -I $R/s-volsdf_amd/csrc -c $R/s-volsdf_amd/csrc/$U -o /tmp/ab_$NAME/${U%.*}.o &
  OBJS=$(echo "$OBJS" | grep -v "/${U%.*}.o")
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/s-volsdf_amd/lib_ab/libsvolsdf_hip_$NAME.so $OBJS /tmp/ab_$NAME/*.o
