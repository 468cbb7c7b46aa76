#!/bin/bash
# usage: _sweep.sh VAR v1 v2 ... ; prints avg kernel time of k_knn_cov_tile for each value
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  rm -rf gpurun_out/sw_$v; export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sw_$v -- python bench.py --steps 2 --warmup 1 --pairs 64 --no-cpu-baseline > /dev/null 2>&1
  f=$(ls gpurun_out/sw_$v/*/*_kernel_stats.csv | tail -1)
  echo "$VAR=$v $(grep -E 'k_knn_cov_tile|k_knn_hard' $f | awk -F, '{printf "%s avg_us=%.1f  ", substr($1,1,30), $4/1000}')"
done
