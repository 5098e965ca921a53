#!/bin/bash
# Builds a variant of the library from one alternative source file, for A/B timing on one GPU box:
#   tools/ab_variant.sh <name> <unit> <alternative source file>      -> s-volsdf_amd/lib_ab/libsvolsdf_hip_<name>.so
# (the other translation units are taken from the objects of the regular build); run with SVS_LIB_PATH=<that file>.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; UNIT=$2; SRC=$3
mkdir -p $R/s-volsdf_amd/lib_ab /tmp/ab_$NAME
cp $SRC /tmp/ab_$NAME/$UNIT
EXTRA=""
case $UNIT in svs_sampler.hip|svs_render.hip|svs_costvol.hip|svs_optim.hip|svs_fusion.hip|svs_cloud.hip) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -x hip $EXTRA -I $R/s-volsdf_amd/csrc -c /tmp/ab_$NAME/$UNIT -o /tmp/ab_$NAME/unit.o
OBJS=$(ls $R/s-volsdf_amd/lib/*.o | grep -v "/${UNIT%.*}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/s-volsdf_amd/lib_ab/libsvolsdf_hip_$NAME.so $OBJS /tmp/ab_$NAME/unit.o
echo $R/s-volsdf_amd/lib_ab/libsvolsdf_hip_$NAME.so
