#!/bin/bash
# Runs ON THE GPU BOX: the bench under alternative settings of environment knobs, alternating, REPS times each.
# Usage: REPS=2 bash benchmarks/env_ab.sh "A=1 B=2" "A=0 B=2" ...
cd "$GRAFT_REPO_ROOT"
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in "$@"; do
    echo "== $cfg (run $rep)"
    env $cfg timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 8 --warmup 2 --bases 8 > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
    python benchmarks/show_bench.py /tmp/ab.json | cut -c1-118 | grep -E "pairs/s|correspond |linearize|k_gicp_step"
  done
done
