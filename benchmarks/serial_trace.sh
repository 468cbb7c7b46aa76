#!/bin/bash
# Runs ON THE GPU BOX: every kernel of the bench step BY ITSELF - one part, the preparation chains one after the other on the batch's
# stream - under rocprofv3 --kernel-trace --stats: the per-kernel "own" durations a change to the prefilter / index chain is judged by.
# Usage: bash benchmarks/serial_trace.sh [out-name] [bench args]   (MI355_LIB=<variant .so> selects a library build)
NAME=${1:-serial}; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out "$(dirname gpurun_out/$NAME)" "$(dirname /tmp/st_$NAME)"
export MI355_BATCH_SIDE_STREAMS=0 MI355_BATCH_PARTS=1
rm -rf /tmp/st_$NAME
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$NAME -- python3 bench.py --pairs 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-kernel-timing "$@" > gpurun_out/${NAME}_bench.json 2> gpurun_out/${NAME}_trace.err
f=$(ls /tmp/st_$NAME/*/*_kernel_stats.csv | tail -1)
cp "$f" gpurun_out/${NAME}_kernel_stats.csv
python3 benchmarks/kstats.py gpurun_out/${NAME}_kernel_stats.csv 40 | grep -v "at::native" | tee gpurun_out/${NAME}_kstats.txt
python3 benchmarks/show_bench.py gpurun_out/${NAME}_bench.json | head -3
