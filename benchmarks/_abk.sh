#!/bin/bash
# kernel-level A/B of two library builds inside one box: average duration of the kernels matching $1 (rocprofv3 stats)
PAT=${1:-k_gicp_error}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/new.so; cp lidar_graph_slam_amd/libmi355_lidar_old.so /tmp/old.so
for r in 1 2; do
  for v in old new; do
    cp /tmp/$v.so lidar_graph_slam_amd/libmi355_lidar.so
    rm -rf gpurun_out/abk_$v
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_$v -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    f=$(ls gpurun_out/abk_$v/*/*_kernel_stats.csv | tail -1)
    echo "[$v] $(python - "$f" "$PAT" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Name']]
print('  '.join(f"{r['Name'][:28]} avg_us={float(r['AverageNs'])/1e3:.1f} n={r['Calls']}" for r in rows))
PY
)"
  done
done
