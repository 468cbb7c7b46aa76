#!/bin/bash
# Runs ON THE GPU BOX: k_gicp_correspond / _tile blocks per pair (MI355_BLOCKS_PER_PAIR) at 1 024 / 2 048 / 4 096 pairs.
cd "$GRAFT_REPO_ROOT"
for p in 1024 2048 4096; do
for g in 2 4 6 8 16; do
echo "== --pairs $p MI355_BLOCKS_PER_PAIR=$g"
MI355_BLOCKS_PER_PAIR=$g timeout 300 python bench.py --no-cpu-baseline --pairs $p --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
bk=d['roofline']['by_kernel']
print(d['value'], d['ms_per_step'], 'correspond', bk['k_gicp_correspond']['total_ms_per_step'], 'tile', bk['k_gicp_correspond_tile']['total_ms_per_step'])"
done; done
