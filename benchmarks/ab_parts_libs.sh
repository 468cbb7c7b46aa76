#!/bin/bash
# Runs ON THE GPU BOX: bench.py per library build (tree + variants/*.so) x MI355_BATCH_PARTS settings.  Usage: PARTS="1 2" bash benchmarks/ab_parts_libs.sh
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  for parts in ${PARTS:-1 2}; do
    echo "== $(basename $f) parts=$parts"
    MI355_BATCH_PARTS=$parts timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 8 --warmup 2 --bases 8 > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
    python benchmarks/show_bench.py /tmp/ab.json | head -${AB_LINES:-4}
  done
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
