#!/bin/bash
# tools/dev/ab_defs.sh <unit> <name> [-Dswitch ...]  -> s-volsdf_amd/lib_ab/libsvolsdf_hip_<name>.so (dev aid: variant builds)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
UNIT=$1; NAME=$2; shift 2
mkdir -p $R/s-volsdf_amd/lib_ab /tmp/ab_$NAME
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -x hip "$@" -I $R/s-volsdf_amd/csrc -c $R/s-volsdf_amd/csrc/$UNIT -o /tmp/ab_$NAME/unit.o
OBJS=$(ls $R/s-volsdf_amd/lib/*.o | grep -v "/${UNIT%.*}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/s-volsdf_amd/lib_ab/libsvolsdf_hip_$NAME.so $OBJS /tmp/ab_$NAME/unit.o
