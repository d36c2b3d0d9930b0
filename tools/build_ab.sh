#!/bin/bash
# Build an A/B variant of the product library: the named sources recompiled with extra flags, every other object reused.
# usage: bash tools/build_ab.sh <tag> <source.hip|source.cpp[,source2...]> "<extra flags>"   ->  ab_libs/libomx_<tag>.so
# (run them with tools/ab_bench.sh / tools/ab_spectrum.sh, or OMX_HIP_LIB=$PWD/ab_libs/libomx_<tag>.so <any tool>)
set -e
TAG=$1; SRCS=$2; EXTRA=$3
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/openmeters_amd/csrc
mkdir -p $ROOT/ab_libs/obj
OTHERS=$(ls $C/*.o)
NEW=""
for SRC in ${SRCS//,/ }; do
  BASE=$(basename ${SRC%.*})
  OBJ=$ROOT/ab_libs/obj/${TAG}_$BASE.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $EXTRA -I$C -x hip -c $C/$SRC -o $OBJ \
    -Rpass-analysis=kernel-resource-usage > $OBJ.log 2>&1 || { grep -E "error" -A3 $OBJ.log | head -20; exit 1; }
  OTHERS=$(echo "$OTHERS" | grep -v "/$BASE.o")
  NEW="$NEW $OBJ"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/libomx_$TAG.so $OTHERS $NEW
echo "built ab_libs/libomx_$TAG.so (kernel resources: ab_libs/obj/${TAG}_*.o.log)"
