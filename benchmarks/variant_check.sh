#!/bin/bash
# Runs ON THE GPU BOX: parity tests of the cloud kernels for the tree's library, then for every variant under variants/, then the bench lines.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f)"
  timeout 900 python -m pytest tests/test_gpu_filters.py tests/test_golden.py tests/test_gpu_device_clouds.py -m gpu -x -q --timeout 300 2>&1 | tail -4
  MI355_BATCH_PARTS=${PARTS:-1} timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 6 --warmup 2 --bases 8 > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python benchmarks/show_bench.py /tmp/ab.json | head -12
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
