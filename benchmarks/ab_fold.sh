#!/bin/bash
# Runs ON THE GPU BOX: the headline bench with the margin check folded into k_gicp_linearize (MI355_CORR_FOLD=1, the default), without it
# (=0), and with the round-4 library (margin-safe bits) where lidar_graph_slam_amd/variants/r04.so exists; REPS runs each, one box.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_fold
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 8 --warmup 3 ${AB_ARGS:-} > gpurun_out/ab_fold/$name.json 2> gpurun_out/ab_fold/$name.err || tail -3 gpurun_out/ab_fold/$name.err
  echo "== $name"; python benchmarks/show_bench.py gpurun_out/ab_fold/$name.json | cut -c1-190 | head -${AB_LINES:-11}
}
for r in $(seq 1 ${REPS:-1}); do
  run fold1.$r MI355_CORR_FOLD=1
  run fold0.$r MI355_CORR_FOLD=0
  [ -f lidar_graph_slam_amd/variants/r04.so ] && run r04lib.$r MI355_LIB=$PWD/lidar_graph_slam_amd/variants/r04.so
done
