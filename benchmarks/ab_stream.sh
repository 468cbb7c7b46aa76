#!/bin/bash
# Runs ON THE GPU BOX: A/B of library variants (lidar_graph_slam_amd/variants/*.so) on the odometry stream (config 4).
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f)"
  python benchmarks/stream_profile.py 60 | tail -1
  python benchmarks/stream_profile.py 60 | tail -1
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
