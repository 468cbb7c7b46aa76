#!/bin/bash
# Runs ON THE GPU BOX after profiles/<round>_issue.json has been committed: the lines that READ it (python bench.py, ndt_batch.py), one full
# `pytest -m gpu` for the parity deltas, and smoke().
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R; mkdir -p "$O/ndt"
rm -f gpurun_out/parity_deltas.jsonl
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > "$O/pytest_gpu.txt" 2>&1
timeout 900 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
timeout 600 python benchmarks/ndt_batch.py --config3 > "$O/ndt/ndt_batch.json" 2> "$O/ndt/ndt_batch.err"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.txt" 2>&1
tail -3 "$O/pytest_gpu.txt"; python benchmarks/show_bench.py "$O/bench_default.json" | head -3; cut -c1-300 "$O/ndt/ndt_batch.json"; cat "$O/smoke.txt" | tail -2
