#!/bin/bash
# Runs ON THE GPU BOX: env_ab.sh on the bench's own 32 scenes (the default input), REPS times each, the families given in FILTER.
cd "$GRAFT_REPO_ROOT"
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in "$@"; do
    echo "== $cfg (run $rep)"
    env $cfg timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 8 --warmup 2 > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
    python benchmarks/show_bench.py /tmp/ab.json | cut -c1-118 | grep -E "pairs/s|${FILTER:-k_knn}"
  done
done
