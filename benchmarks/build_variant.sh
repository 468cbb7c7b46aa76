#!/bin/bash
# Builds lidar_graph_slam_amd/variants/<name>.so: the library with one or more translation units recompiled under extra -D flags
# (A/B of kernel variants with benchmarks/ab_libs.sh).  Usage: bash benchmarks/build_variant.sh <name> <unit[,unit...] (cloud|gicp|...)> <flags...>
set -e
NAME=$1; UNITS=$2; shift; shift
cd "$(dirname "$0")/../lidar_graph_slam_amd/csrc"
make -s
mkdir -p ../variants build/var
OBJS=$(ls build/*.o)
for UNIT in ${UNITS//,/ }; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unknown-pragmas "$@" -c $UNIT.hip -o build/var/$NAME.$UNIT.o &
  OBJS=$(echo "$OBJS" | grep -v "build/$UNIT.o")
done
wait
OBJS="$OBJS $(ls build/var/$NAME.*.o)"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/$NAME.so $OBJS
echo built ../variants/$NAME.so
