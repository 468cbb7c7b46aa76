#!/bin/bash
# Runs ON THE GPU BOX: the covariance / k-NN parity tests and the bench line with the brick kernel (default) and the round-2 tile kernel.
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_filters.py tests/test_gpu_properties.py tests/test_golden.py tests/test_gpu_gicp.py -m gpu -q 2>&1 | tail -25
run() {
  echo "== $*"
  env "$@" python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/tmp/ab.err > /tmp/ab.json; grep "mi355" /tmp/ab.err | tail -4; python benchmarks/show_bench.py /tmp/ab.json | grep -E "pairs/s|knn"
}
run MI355_KNN_TILES=1
run MI355_KNN_TILES=2
run MI355_KNN_TILES=2 MI355_KNN_STATS=1
for h in 0.45 0.8; do run MI355_BRICK_H=$h MI355_KNN_STATS=1; done
for g in 1.25 1.5; do run MI355_BRICK_GATHER=$g MI355_KNN_STATS=1; done
run MI355_TILE_RSCALE=1.2 MI355_KNN_STATS=1
run MI355_TILE_RSCALE=0.85 MI355_KNN_STATS=1
