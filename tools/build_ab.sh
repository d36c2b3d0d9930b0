#!/bin/bash
# Build an A/B variant of the product library: one source recompiled with extra flags, every other object reused.
# usage: bash tools/build_ab.sh <tag> <source.hip> "<extra flags>"   ->  ab_libs/libomx_<tag>.so   (run them with tools/ab_bench.sh)
set -e
TAG=$1; SRC=$2; EXTRA=$3
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/openmeters_amd/csrc
mkdir -p $ROOT/ab_libs/obj
OBJ=$ROOT/ab_libs/obj/${TAG}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $EXTRA -I$C -x hip -c $C/$SRC -o $OBJ \
  -Rpass-analysis=kernel-resource-usage > $OBJ.log 2>&1 || { grep -E "error" -A3 $OBJ.log | head -20; exit 1; }
grep -E "VGPRs:|ScratchSize" $OBJ.log | tr -s ' ' | sed 's/.*remark://' | tr '\n' ' '; echo
OTHERS=$(ls $C/*.o | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/libomx_$TAG.so $OTHERS $OBJ
echo "built ab_libs/libomx_$TAG.so"
