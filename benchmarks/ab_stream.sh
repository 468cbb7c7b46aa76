#!/bin/bash
# Runs ON THE GPU BOX: A/B of library variants on the odometry stream (config 4) and on the bench's correspondence kernels.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f)"
  python benchmarks/stream_profile.py 60 | tail -1
  python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null > /tmp/ab.json; python benchmarks/show_bench.py /tmp/ab.json | grep -E "pairs/s|correspond"
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
