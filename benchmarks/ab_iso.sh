#!/bin/bash
# Runs ON THE GPU BOX: like ab_libs.sh, but prints the ISOLATED per-launch duration of every kernel family (bench.py's extra steps
# with the side streams off) next to the step time.  Usage: bash benchmarks/ab_iso.sh [bench args]
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for r in 1 2; do
  for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
    cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
    python bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > /tmp/ab.json 2>/dev/null
    python - "$(basename $f)" <<'PY'
import json,sys
j=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
bk=j['roofline']['by_kernel']
print(sys.argv[1].ljust(12), '%.0f pairs/s %.2f ms/step |' % (j['value'], j['ms_per_step']), ' '.join('%s %.3f' % (k.split('(')[0][-14:], v.get('isolated_avg_launch_ms') or 0) for k,v in bk.items()))
PY
  done
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
