#!/bin/bash
# Runs ON THE GPU BOX: the bench line (and its correspondence rows) for the tree's library and every lidar_graph_slam_amd/variants/*.so.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f)"
  python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null > /tmp/ab.json; python benchmarks/show_bench.py /tmp/ab.json | grep -E "pairs/s|${AB_GREP:-correspond}"
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
