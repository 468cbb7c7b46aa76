#!/bin/bash
# Runs ON THE GPU BOX: GPU test suite + a short bench line with the per-kernel table (+ optional kernel trace).
# Usage: gpurun -- 'bash benchmarks/quick.sh <tag> [notest] [trace] [-- pytest args]'
TAG=${1:-quick}; shift
NOTEST=0; TRACE=0
while [ $# -gt 0 ]; do case "$1" in notest) NOTEST=1;; trace) TRACE=1;; --) shift; break;; *) break;; esac; shift; done
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
if [ $NOTEST = 0 ]; then
  timeout 600 python -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread "$@" > "$O/pytest.txt" 2>&1; echo "pytest exit $?" >> "$O/pytest.txt"
  tail -12 "$O/pytest.txt"
fi
MI355_KNN_STATS=1 timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --pairs 128 2>&1 | grep "k-NN tiles" | head -4
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > "$O/bench.json" 2> "$O/bench.err"; echo "bench exit $?"
tail -3 "$O/bench.err"
python benchmarks/show_bench.py "$O/bench.json"
if [ $TRACE = 1 ]; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 > "$O/bench_trace.json" 2> "$O/trace.err"
  f=$(ls $O/trace/*/*_kernel_stats.csv | tail -1); cp "$f" "$O/kernel_stats.csv"
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:10.1f} share={100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
fi
