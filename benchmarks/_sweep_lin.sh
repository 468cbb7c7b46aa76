#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "2 4" "2 8" "1 4" "1 8" "4 4"; do
  set -- $cfg
  echo -n "PPT=$1 G=$2: "
  MI355_LIN_PPT=$1 MI355_BLOCKS_LIN=$2 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['accuracy'])"
done
