#!/bin/bash
# Runs ON THE GPU BOX: bench.py under MI355_CORR_TILES x MI355_CORR_TILES_TRACKING (tile search of the correspondences in the first iterations).
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['accuracy'])
bk=d['roofline']['by_kernel']
for k in ('k_gicp_correspond','k_gicp_correspond_tile'): print('  ',k, bk[k]['total_ms_per_step'], bk[k]['launches_per_step'])"; }
for t in 2 3 4; do for k in 0 1; do run MI355_CORR_TILES=$t MI355_CORR_TILES_TRACKING=$k; done; done
run MI355_CORR_TILES=3 MI355_CORR_TILES_TRACKING=0 MI355_MARGIN_STEP=0.05
run MI355_CORR_TILES=3 MI355_CORR_TILES_TRACKING=0 MI355_MARGIN_STEP=0.02
