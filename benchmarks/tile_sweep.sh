#!/bin/bash
# Runs ON THE GPU BOX: knn_only.py for the tree's library (optionally under tunings given as arguments "A=1 B=2" ...), then for every variant under variants/.
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" MI355_KNN_STATS=1 timeout 300 python benchmarks/knn_only.py 1024 2>&1 | grep -E "lists per tile|handed" | tail -2; env "$@" timeout 300 python benchmarks/knn_only.py 1024 2>&1 | tail -1; }
run A=0
for t in "$@"; do run $t; done
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in lidar_graph_slam_amd/variants/*.so; do
  [ -f "$f" ] || continue
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  run V=$(basename $f)
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
