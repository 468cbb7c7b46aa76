#!/bin/bash
# A/B two builds of the library inside one box: libmi355_lidar.so (new) vs libmi355_lidar_old.so
cd $GRAFT_REPO_ROOT/lidar_graph_slam_amd
cp libmi355_lidar.so /tmp/new.so; cp libmi355_lidar_old.so /tmp/old.so
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in old new; do
    cp /tmp/$v.so lidar_graph_slam_amd/libmi355_lidar.so
    echo -n "[$v] "
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
