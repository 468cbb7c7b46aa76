#!/bin/bash
# Runs ON THE GPU BOX: GPU test suite + a short bench line with the per-kernel table.  Usage: gpurun -- 'bash benchmarks/quick.sh <tag> [pytest args]'
TAG=${1:-quick}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
timeout 900 python -m pytest tests -m gpu -x -q "$@" > "$O/pytest.txt" 2>&1; echo "pytest exit $?" >> "$O/pytest.txt"
tail -15 "$O/pytest.txt"
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > "$O/bench.json" 2> "$O/bench.err"; echo "bench exit $?"
tail -3 "$O/bench.err"
python - "$O/bench.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench json", e); sys.exit(0)
print("pairs/s", d["value"], "ms/step", d["ms_per_step"], "e2e frac", d["roofline"]["end_to_end"]["frac"], "acc", d["accuracy"])
for k,v in d["roofline"]["by_kernel"].items():
    print(f'  {k[:40]:40s} {v["total_ms_per_step"]:8.3f} ms/step  n={v["launches_per_step"]:5.1f} avg={v["avg_launch_ms"]:8.4f} frac={v.get("frac_of_hbm_peak")}')
PY
