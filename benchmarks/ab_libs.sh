#!/bin/bash
# Runs ON THE GPU BOX: bench.py once per library build found in lidar_graph_slam_amd/variants/*.so (A/B of kernel variants inside one
# box: boxes of the pool differ by several per cent).  Usage: bash benchmarks/ab_libs.sh [bench args]
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for r in 1 2; do
  for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
    cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
    echo "== $(basename $f)"
    python bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > /tmp/ab.json 2>/dev/null
    python benchmarks/show_bench.py /tmp/ab.json | head -${AB_LINES:-3}
  done
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
