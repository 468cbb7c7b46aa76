#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of benchmarks/single_pair.py for the tree's library and every variant under variants/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f)"
  rm -rf /tmp/sp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python3 benchmarks/single_pair.py > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
f=sorted(glob.glob('/tmp/sp/*/*_kernel_stats.csv'))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print('  %-52s calls %6s avg %8.1f us total %8.2f ms'%(r['Name'].replace('mi355::','')[:52], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
print('  total kernel ms', tot/1e6)
PY
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
