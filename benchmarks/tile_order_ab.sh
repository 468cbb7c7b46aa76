#!/bin/bash
# Runs ON THE GPU BOX: the preparation chain alone (benchmarks/knn_only.py) with the tiles of k_knn_cov_tile along the Morton order
# of the voxel hash (MI355_TILE_ORDER=0) and along the 2-D Hilbert curve (1), for the tree's library and every variant under variants/.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  [ -f "$f" ] || continue
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  for o in 0 1; do
    echo "== $(basename $f) MI355_TILE_ORDER=$o"
    MI355_TILE_ORDER=$o MI355_KNN_STATS=1 timeout 300 python benchmarks/knn_only.py 1024 2>&1 | grep -E "by segments|phases|lists per tile|handed" | tail -4
    MI355_TILE_ORDER=$o timeout 300 python benchmarks/knn_only.py 1024 2>&1 | tail -1
  done
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
