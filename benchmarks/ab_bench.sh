#!/bin/bash
# Runs ON THE GPU BOX: python bench.py for the tree's library and every variant under variants/, twice, alternating.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for rep in 1 2; do
for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
  cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
  echo "== $(basename $f) (run $rep)"
  timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print(d['value'], d['ms_per_step'], 'centroids', bk['k_voxel_centroids']['isolated_avg_launch_ms'], 'hash', bk['hash_build(k_count_cells,k_table_clear,k_gather_insert)']['isolated_avg_launch_ms'])"
done; done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
