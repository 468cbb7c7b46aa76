#!/bin/bash
# Runs ON THE GPU BOX: the bench line under alternative settings of MI355_* tuning knobs (common.hpp: Tuning), the default before and after
# (boxes of the pool differ by several per cent: only compare inside one call).  usage: bash benchmarks/sweep_env.sh "A=1" "B=2 C=3" ...
cd "$GRAFT_REPO_ROOT"
run() { env $1 python bench.py --no-cpu-baseline --no-h2d --steps ${SWEEP_STEPS:-8} --warmup 3 $SWEEP_ARGS 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split('(')[0]: b for n, b in d['roofline']['by_kernel'].items()}
own = lambda n: (k[n].get('isolated_avg_launch_ms', float('nan')) * k[n]['launches_per_step'] if n in k else float('nan'))
print('%-34s %8.0f pairs/s %7.2f ms/step | knn %.1f hard %.1f tile %.1f corr %.1f lin %.1f sort %.1f hash %.1f' % (sys.argv[1] or 'default', d['value'], d['ms_per_step'], own('k_knn_cov_tile'), own('k_knn_hard'), own('k_gicp_correspond_tile'), own('k_gicp_correspond'), own('k_gicp_linearize'), own('sort_pairs'), own('hash_build')))
" "$1"; }
run "X=0"
for a in "$@"; do run "$a"; done
run "X=0"
