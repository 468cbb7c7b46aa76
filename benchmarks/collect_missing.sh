#!/bin/bash
# Runs ON THE GPU BOX: the legs of collect_evidence.sh whose outputs a first run lost (the kernel trace and the counter passes of the bench,
# config 5 with GICP), on top of that run's small files staged under r06_stage/ (gpurun_out/ does not travel to the box), then the distillation.
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R
mkdir -p "$O"; cp -r r06_stage/* "$O"/
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python bench.py --no-cpu-baseline --no-h2d > "$O/bench_under_trace.json" 2> "$O/trace.err"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> "$O/pmc_fetch.err"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> "$O/pmc_write.err"
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
timeout 1200 rocprofv3 --kernel-trace --pmc $G1 --output-format csv -d "$O/pmc_sq1" -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> "$O/pmc_sq1.err"
timeout 1200 rocprofv3 --kernel-trace --pmc $G2 --output-format csv -d "$O/pmc_sq2" -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> "$O/pmc_sq2.err"
timeout 900 python benchmarks/config5.py --candidates 4096 --submaps 4 --method GICP 2> "$O/config5_GICP.err" | tail -1 > "$O/config5_GICP.json"
# the new sparse-target ICP test and the ABI-5 result layout (built after the first run)
timeout 900 python -m pytest tests/test_gpu_icp_batch.py tests/test_gpu_ndt.py tests/test_gpu_gicp_bfgs_batch.py tests/test_gpu_loop_closure.py -m gpu -q > "$O/pytest_gpu_after.txt" 2>&1
if EVIDENCE_DST="$O/profiles" python benchmarks/summarize_evidence.py $R > "$O/summarize.log" 2>&1; then
  rm -rf "$O/trace" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq1" "$O/pmc_sq2"
else
  echo "summarize_evidence.py FAILED"
fi
tail -5 "$O/summarize.log"; tail -3 "$O/pytest_gpu_after.txt"; ls "$O/profiles"; cat "$O/config5_GICP.json" | cut -c1-400
