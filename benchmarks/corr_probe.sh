#!/bin/bash
# Runs ON THE GPU BOX: correspondence-tile tuning aid — parity tests of the GICP path, the tile counters, a kernel trace of the
# per-iteration launches.  Usage: gpurun -- 'bash benchmarks/corr_probe.sh <tag>'
TAG=${1:-corr}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_gicp.py tests/test_golden.py -m gpu -x -q --timeout 150 --timeout-method thread > "$O/pytest.txt" 2>&1; echo "pytest exit $?" >> "$O/pytest.txt"
tail -5 "$O/pytest.txt"
MI355_CORR_STATS=1 timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --pairs 128 2>&1 | grep "correspondence tiles" | head -3
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-kernel-timing > "$O/bench_trace.json" 2> "$O/trace.err"
f=$(ls $O/trace/*/*_kernel_trace.csv | tail -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_gicp_correspond' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tile=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows if 'tile' in r['Kernel_Name']]
lane=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows if 'tile' not in r['Kernel_Name']]
print("tile launches (ms), last step:", " ".join(f"{x:.2f}" for x in tile[-10:]))
print("lane launches (ms), last step:", " ".join(f"{x:.2f}" for x in lane[-20:]))
PY
python benchmarks/show_bench.py "$O/bench_trace.json" | head -3
