#!/bin/bash
# Runs ON THE GPU BOX: env-knob A/B of the early correspondence iterations (bench line + the correspondence rows).
cd "$GRAFT_REPO_ROOT"
run() {
  echo "== $*"
  env "$@" python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null > /tmp/ab.json; python benchmarks/show_bench.py /tmp/ab.json | grep -E "pairs/s|correspond|error|linearize"
}
run A=1
run MI355_MARGIN_STEP=1e9
run MI355_MARGIN_STEP=0.3
run MI355_CORR_TILES=1
run MI355_CORR_TILES=3
run MI355_CORR_TILES=3 MI355_MARGIN_STEP=1e9
run MI355_CORR_TILES=4 MI355_MARGIN_STEP=1e9
