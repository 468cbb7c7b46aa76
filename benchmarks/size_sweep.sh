#!/bin/bash
# Runs ON THE GPU BOX: the bench line at several batch sizes - every kernel family's OWN duration per cloud / pair (does a working set
# that fits the 256 MB Infinity Cache make the gather-bound kernels faster?).
cd "$GRAFT_REPO_ROOT"
for P in ${SIZES:-64 128 256 512 2048}; do
  echo "== --pairs $P"
  timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 6 --warmup 2 --bases 8 --pairs $P > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python benchmarks/show_bench.py /tmp/ab.json | cut -c1-150
done
