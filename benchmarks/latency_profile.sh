#!/bin/bash
# Runs ON THE GPU BOX: where the time of ONE pair / ONE scan goes (configs 2 and 4) — wall clock without a profiler, then the
# rocprofv3 kernel trace of the same programs (GPU-busy time = sum of kernel durations).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/${1:-lat}; mkdir -p $OUT
python benchmarks/single_pair.py > $OUT/single_pair.txt 2>&1
python benchmarks/stream_profile.py 120 FAST_GICP > $OUT/stream.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sp -- python3 benchmarks/single_pair.py > $OUT/sp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 benchmarks/stream_profile.py 120 FAST_GICP > $OUT/st.log 2>&1
cat $OUT/single_pair.txt $OUT/stream.txt
for d in sp st; do f=$(ls $OUT/$d/*/*_kernel_stats.csv | head -1); echo "== $d"; python benchmarks/kstats.py $f 30; done
