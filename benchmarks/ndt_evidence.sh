#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash benchmarks/ndt_evidence.sh r05'): the NDT path measured - bench line, rocprofv3 kernel stats,
# FETCH / WRITE and SQ counters (separate --pmc passes, --kernel-trace only) - under gpurun_out/<round>/ndt/.
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R/ndt
rm -rf "$O"; mkdir -p "$O"
timeout 600 python benchmarks/ndt_batch.py --config3 > "$O/ndt_batch.json" 2> "$O/ndt_batch.err"
if [ -f lidar_graph_slam_amd/variants/r04.so ]; then
  MI355_LIB=$PWD/lidar_graph_slam_amd/variants/r04.so timeout 600 python benchmarks/ndt_batch.py --config3 > "$O/ndt_batch_r04lib.json" 2> "$O/ndt_batch_r04lib.err"
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python benchmarks/ndt_batch.py --steps 5 --warmup 2 --no-oracle --config3 > "$O/ndt_under_trace.json" 2> "$O/trace.err"
if [ "${PMC:-1}" = "1" ]; then
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python benchmarks/ndt_batch.py --steps 2 --warmup 1 --no-oracle > /dev/null 2> "$O/pmc_fetch.err"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python benchmarks/ndt_batch.py --steps 2 --warmup 1 --no-oracle > /dev/null 2> "$O/pmc_write.err"
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
timeout 300 rocprofv3 --kernel-trace --pmc $G1 --output-format csv -d "$O/pmc_sq1" -- python benchmarks/ndt_batch.py --steps 1 --warmup 1 --no-oracle > /dev/null 2> "$O/pmc_sq1.err"
timeout 300 rocprofv3 --kernel-trace --pmc $G2 --output-format csv -d "$O/pmc_sq2" -- python benchmarks/ndt_batch.py --steps 1 --warmup 1 --no-oracle > /dev/null 2> "$O/pmc_sq2.err"
fi
python benchmarks/summarize_ndt.py "$O" > "$O/summarize.log" 2>&1
rm -rf "$O/trace" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq1" "$O/pmc_sq2"
ls -la "$O"; tail -3 "$O"/*.err | tail -40; cat "$O/summarize.log" | tail -40
