#!/bin/bash
# Runs ON THE GPU BOX: the single-pair and stream measurements for the tree's library and every variant under variants/, alternating.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for rep in 1 2; do
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f) (run $rep)"
  timeout 300 python benchmarks/single_pair.py 2>&1 | tail -2
  timeout 300 python benchmarks/stream_profile.py 200 FAST_GICP 2>&1 | tail -1
done; done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
