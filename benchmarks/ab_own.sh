#!/bin/bash
# Runs ON THE GPU BOX: parity tests of the tree's library, then for the tree's library and every variant under variants/ the bench with ONE
# part and the own (isolated) per-launch duration of every kernel family, twice.
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_filters.py tests/test_golden.py tests/test_gpu_gicp.py tests/test_a_tile_order.py -m gpu -x -q --timeout 900 2>&1 | tail -4
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for rep in 1 2; do
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f) (run $rep)"
  MI355_BATCH_PARTS=1 timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 5 --warmup 2 --bases 8 > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python benchmarks/show_bench.py /tmp/ab.json | cut -c1-118 | grep -E "pairs/s|knn|correspond|centroid|hash"
done; done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
