#!/bin/bash
# Runs ON THE GPU BOX: launch geometry of the streaming LM kernels (MI355_BLOCKS_LIN blocks per pair) at 512 .. 4 096 pairs.
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "${@:2}" timeout 300 python bench.py --no-cpu-baseline --pairs $1 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
bk=d['roofline']['by_kernel']
print(d['value'], d['ms_per_step'], 'linearize', bk['k_gicp_linearize']['total_ms_per_step'], 'error', bk['k_gicp_error']['total_ms_per_step'])"; }
for p in 512 1024 2048 4096; do for g in 3 4 6; do run $p MI355_BLOCKS_LIN=$g; done; done
