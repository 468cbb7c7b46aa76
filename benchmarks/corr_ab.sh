#!/bin/bash
# Runs ON THE GPU BOX: correspondence parity tests, then the bench line with the round-1 brick on / off and for more tile iterations.
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_properties.py tests/test_golden.py tests/test_gpu_multi.py tests/test_gpu_loop_closure.py -m gpu -q 2>&1 | tail -15
run() {
  echo "== $*"
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/tmp/ab.err > /tmp/ab.json; grep "mi355" /tmp/ab.err | tail -3; python benchmarks/show_bench.py /tmp/ab.json | grep -E "pairs/s|correspond"
}
run MI355_CORR_BRICK_H=0
run MI355_CORR_BRICK_H=0.5
run MI355_CORR_BRICK_H=0.5 MI355_CORR_STATS=1
run MI355_CORR_BRICK_H=0.35
run MI355_CORR_BRICK_H=0.7
run MI355_CORR_TILES=3
run MI355_CORR_TILES=4
run MI355_CORR_TILES=6
run MI355_CORR_TILES=4 MI355_CORR_R1=0.35
run MI355_CORR_TILES=4 MI355_CORR_R1=0.7
