#!/bin/bash
# Runs ON THE GPU BOX: config 5 with the back end's default engine (ICP) under variants of its two search knobs (common.hpp:
# MI355_ICP_R1_CELLS, MI355_ICP_MIN_MOVE).  Every variant gives the same poses (the knobs decide who searches a point, never the answer).
# Usage: bash benchmarks/icp_ab.sh [out-name] [config5 args]
NAME=${1:-icp_ab}; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
: > gpurun_out/$NAME.txt
run() {
  echo "== $*" | tee -a gpurun_out/$NAME.txt
  env "$@" timeout 900 python3 benchmarks/config5.py --method ICP --candidates 2048 --submaps 32 --repeats 3 --no-cpu-baseline $EXTRA 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('pairs_per_s_registration','seconds_register_max_over_ranks','within_30cm_of_truth','median_pose_err_m','mean_iterations')})" | tee -a gpurun_out/$NAME.txt
}
EXTRA="$*"
run A=default
run A=default_again
