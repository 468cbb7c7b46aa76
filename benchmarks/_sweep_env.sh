#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo -n "[$cfg] "
  env $cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
