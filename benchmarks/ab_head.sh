#!/bin/bash
# Runs ON THE GPU BOX: the bench of a copy of an older tree (ab_head/: bench.py + package with its own built library, staged by hand
# from a git worktree) beside the current library and its variants (lidar_graph_slam_amd/variants/*.so), inside one box.
cd "$GRAFT_REPO_ROOT"
cp lidar_graph_slam_amd/libmi355_lidar.so /tmp/base.so
for r in $(seq 1 ${REPS:-2}); do
  if [ -d ab_head ]; then
    echo "== ab_head (run $r)"
    (cd ab_head && python bench.py --no-cpu-baseline --no-h2d --steps 6 --warmup 2 "$@" > /tmp/abh.json 2>/tmp/abh.err; python benchmarks/show_bench.py /tmp/abh.json | cut -c1-200 | head -${AB_LINES:-12})
  fi
  for f in /tmp/base.so lidar_graph_slam_amd/variants/*.so; do
    [ -f "$f" ] || continue
    cp "$f" lidar_graph_slam_amd/libmi355_lidar.so
    echo "== $(basename $f) (run $r)"
    python bench.py --no-cpu-baseline --no-h2d --steps 6 --warmup 2 "$@" > /tmp/ab.json 2>/tmp/ab.err || tail -5 /tmp/ab.err
    python benchmarks/show_bench.py /tmp/ab.json | cut -c1-200 | head -${AB_LINES:-12}
  done
done
cp /tmp/base.so lidar_graph_slam_amd/libmi355_lidar.so
