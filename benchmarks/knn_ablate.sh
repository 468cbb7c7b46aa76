#!/bin/bash
# Runs ON THE GPU BOX: benchmarks/knn_only.py for the tree's library and every variant under lidar_graph_slam_amd/variants/.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f)"
  MI355_KNN_STATS=1 timeout 300 python benchmarks/knn_only.py 1024 2>&1 | grep -E "handed over|k_knn_cov_tile" | tail -2; timeout 300 python benchmarks/knn_only.py 1024 2>&1 | tail -1
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
