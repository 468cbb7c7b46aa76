#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() {
  echo "== $*"
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/tmp/ab.err > /tmp/ab.json; grep "mi355" /tmp/ab.err | tail -2; python - <<'PY'
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
bk=d['roofline']['by_kernel']
print('pairs/s', d['value'], 'ms/step', d['ms_per_step'], 'knn iso', bk['k_knn_cov_tile'].get('isolated_avg_launch_ms'), 'hard iso', bk['k_knn_hard'].get('isolated_avg_launch_ms'))
PY
}
run A=1
run MI355_TILE_GATHER=1.25
run MI355_TILE_GATHER=1.5
run MI355_TILE_GATHER=2.0
run MI355_TILE_GATHER=1.5 MI355_KNN_STATS=1
run MI355_TILE_GATHER=1.5 MI355_TILE_RSCALE=0.8
run MI355_TILE_GATHER=1.5 MI355_TILE_RSCALE=0.8 MI355_KNN_STATS=1
run MI355_TILE_GATHER=2.0 MI355_TILE_RSCALE=0.7
timeout 600 python -m pytest tests/test_gpu_filters.py -m gpu -q 2>&1 | tail -3
MI355_TILE_GATHER=1.5 timeout 600 python -m pytest tests/test_gpu_filters.py -m gpu -q 2>&1 | tail -3
