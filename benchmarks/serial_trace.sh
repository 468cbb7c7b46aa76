#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the bench with ONE part and the preparation chains serialised - every kernel's own duration.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-serial}; rm -rf "$O"; mkdir -p "$O"
MI355_BATCH_PARTS=1 MI355_BATCH_SIDE_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python bench.py --no-cpu-baseline --no-h2d --no-kernel-timing --steps 4 --warmup 1 --bases 8 > "$O/bench.json" 2> "$O/trace.err"
f=$(ls $O/trace/*/*_kernel_stats.csv | tail -1); cp "$f" "$O/kernel_stats.csv"
python - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'mi355' in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
steps=5.0
for r in rows[:40]:
    print(f"{r['Name'][:58]:58s} calls={int(r['Calls'])/steps:7.1f}/step avg_us={float(r['AverageNs'])/1e3:9.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:8.3f} share={100*float(r['TotalDurationNs'])/tot:5.1f}%")
print("sum ms/step", tot/1e6/steps)
PY
rm -rf "$O/trace"
