#!/bin/bash
# Runs ON THE GPU BOX: the pipeline of a batch with its preparation chains and LM loops on (partly) disjoint compute units.
cd "$GRAFT_REPO_ROOT"
run() {
  echo "== $*"
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-h2d --no-kernel-timing --steps 8 --warmup 2 --bases 8 2>/tmp/ab.err | tail -1 || tail -3 /tmp/ab.err
}
run MI355_BATCH_PARTS=1
run MI355_BATCH_PARTS=2
run MI355_BATCH_PARTS=2 MI355_PREP_CUS=0.75
run MI355_BATCH_PARTS=2 MI355_PREP_CUS=0.75 MI355_LM_CUS=0.25
run MI355_BATCH_PARTS=2 MI355_PREP_CUS=0.875
run MI355_BATCH_PARTS=2 MI355_PREP_CUS=0.625 MI355_LM_CUS=0.375
run MI355_BATCH_PARTS=4 MI355_PREP_CUS=0.75
run MI355_BATCH_PARTS=4 MI355_PREP_CUS=0.75 MI355_LM_CUS=0.25
run MI355_BATCH_PARTS=4 MI355_PREP_CUS=0.5 MI355_LM_CUS=0.5
run MI355_BATCH_PARTS=8 MI355_PREP_CUS=0.75 MI355_LM_CUS=0.25
