#!/bin/bash
# Runs ON THE GPU BOX: bench.py under several values of one MI355_* tuning knob.  Usage: bash benchmarks/sweep_env.sh VAR v1 v2 ... [-- bench args]
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" = "--" ] && shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/sweep
for v in "${VALS[@]}"; do
  echo "== $VAR=$v"
  env $VAR=$v MI355_KNN_STATS=${KNN_STATS:-0} python bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > gpurun_out/sweep/$VAR.$v.json 2> gpurun_out/sweep/$VAR.$v.err
  grep "k-NN tiles" gpurun_out/sweep/$VAR.$v.err | head -2
  python benchmarks/show_bench.py gpurun_out/sweep/$VAR.$v.json | head -4
done
