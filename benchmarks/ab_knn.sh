#!/bin/bash
# Runs ON THE GPU BOX: A/B of library variants (lidar_graph_slam_amd/variants/*.so) on the covariance kernels — hand-over rate,
# isolated kernel times (MI355_KNN_STATS synchronises after every covariance launch) and the step.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f)"
  MI355_KNN_STATS=1 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/tmp/err.txt > /tmp/ab.json
  grep "k-NN tiles" /tmp/err.txt | head -1
  python benchmarks/show_bench.py /tmp/ab.json | grep -E "pairs/s|k_knn"
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
