#!/bin/bash
# Runs ON THE GPU BOX: the batch pipeline of api.hip (MI355_BATCH_PARTS = parts a batch is cut into) - parity tests, then the bench line per setting.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/parts
timeout 900 python -m pytest tests/test_a_batch_parts.py tests/test_gpu_gicp.py tests/test_gpu_multi.py tests/test_gpu_loop_closure.py tests/test_gpu_scan_matcher.py -m gpu -x -q --timeout 300 2>&1 | tail -15
for parts in ${PARTS:-1 2 4 6}; do
  echo "== MI355_BATCH_PARTS=$parts"
  MI355_BATCH_PARTS=$parts timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 10 --warmup 3 > gpurun_out/parts/b$parts.json 2> gpurun_out/parts/b$parts.err || tail -5 gpurun_out/parts/b$parts.err
  python benchmarks/show_bench.py gpurun_out/parts/b$parts.json | head -12
done
