#!/bin/bash
# Runs ON THE GPU BOX: A/B of library variants (lidar_graph_slam_amd/variants/*.so) on the NDT paths: batched loop candidates
# (config 5 with NDT_OMP) and the odometry stream with NDT_OMP.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f)"
  python benchmarks/config5.py --candidates 128 --submaps 8 --method NDT_OMP 2>/dev/null | tail -1 | cut -c1-400
  python benchmarks/stream_profile.py 60 NDT_OMP 2>/dev/null | tail -1
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
