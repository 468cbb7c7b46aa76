#!/bin/bash
# Runs ON THE GPU BOX: LM-loop parity tests, then bench lines with the "safe" bits of the correspondence margin check on / off, then the library variants.
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_a_batch_parts.py tests/test_a_tile_order.py tests/test_gpu_gicp.py tests/test_gpu_properties.py tests/test_golden.py tests/test_gpu_config5_multi.py -m gpu -x -q --timeout 900 2>&1 | tail -6
for m in 1 0; do
  echo "== MI355_CORR_MASK=$m"
  MI355_CORR_MASK=$m timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 8 --warmup 2 --bases 8 > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python benchmarks/show_bench.py /tmp/ab.json | head -12
done
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f)"
  timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 8 --warmup 2 --bases 8 > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python benchmarks/show_bench.py /tmp/ab.json | head -12
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
