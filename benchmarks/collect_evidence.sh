#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash benchmarks/collect_evidence.sh r05'): everything profiles/README.md cites, written
# under gpurun_out/<round>/; benchmarks/summarize_evidence.py then distils it into profiles/.
# rocprofv3 is given the program directly (no env/bash hop) and PMC passes carry --kernel-trace only.
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R
mkdir -p "$O"; rm -rf "$O/trace" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq1" "$O/pmc_sq2" "$O/profiles"
# ---- the headline line (default: 4 096 pairs per step) and its profile
timeout 900 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python bench.py --no-cpu-baseline --no-h2d > "$O/bench_under_trace.json" 2> "$O/trace.err"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> "$O/pmc_fetch.err"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> "$O/pmc_write.err"
# VALU / LDS / wait counters of the hot kernels at the bench's full batch (two passes: 8 SQ slots each), --kernel-trace only
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
timeout 1200 rocprofv3 --kernel-trace --pmc $G1 --output-format csv -d "$O/pmc_sq1" -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> "$O/pmc_sq1.err"
timeout 1200 rocprofv3 --kernel-trace --pmc $G2 --output-format csv -d "$O/pmc_sq2" -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> "$O/pmc_sq2.err"
# ---- the same workload at other batch sizes / settings (never the headline)
for P in 128 1024 2048; do
  timeout 600 python bench.py --pairs $P --no-cpu-baseline --no-h2d > "$O/bench_${P}pairs.json" 2> "$O/bench_${P}pairs.err"
done
MI355_BATCH_SIDE_STREAMS=0 timeout 600 python bench.py --no-cpu-baseline --no-h2d > "$O/bench_serial_streams.json" 2> "$O/bench_serial_streams.err"
timeout 900 python bench.py --host-inputs --pairs 2048 --no-cpu-baseline > "$O/bench_hostinputs.json" 2> "$O/bench_hostinputs.err"
timeout 900 python bench.py --sor --pairs 2048 --no-cpu-baseline --no-h2d --steps 5 --warmup 2 > "$O/bench_sor.json" 2> "$O/bench_sor.err"
# ---- the BASELINE configs that are not the bench line
timeout 300 python benchmarks/single_pair.py > "$O/single_pair.txt" 2>&1
timeout 300 python benchmarks/stream_profile.py 200 FAST_GICP > "$O/stream_profile.txt" 2>&1
timeout 300 python benchmarks/stream_profile.py 200 NDT_OMP >> "$O/stream_profile.txt" 2>&1
timeout 900 python benchmarks/other_configs.py --scans 200 > "$O/other_configs.txt" 2> "$O/other_configs.err"
# config 5 at its stated size (4 096 candidates x 64 sub-maps) on ONE GPU: through the multi entry (host views in, uploads included) ...
timeout 900 python benchmarks/config5.py --candidates 4096 --submaps 64 --multi-lanes 2 2> "$O/config5_1gpu.err" | tail -1 > "$O/config5_1gpu.json"
# ... and through one BatchRegistration per rank with each engine GraphBasedSLAM::get_registration offers a batch for
for M in FAST_GICP ICP; do
  timeout 900 python benchmarks/config5.py --candidates 4096 --submaps 64 --method $M 2> "$O/config5_$M.err" | tail -1 > "$O/config5_$M.json"
done
# (NDT_OMP: 4 096 candidates against 8 sub-maps within 160 m of the map origin.  NDT's pose is (translation, Euler angles about the ORIGIN)
# with fp32 per-point maths: a kilometre out - the far sub-maps of the 64 - the restated reference itself no longer converges, the
# oracle exactly like the GPU path: DESIGN 5.)
timeout 900 python benchmarks/config5.py --candidates 4096 --submaps 8 --method NDT_OMP --min-good 0.8 2> "$O/config5_NDT_OMP.err" | tail -1 > "$O/config5_NDT_OMP.json"
# (pclomp's BFGS GICP: the same parametrisation; 1 - 2 cm at <= 40 m from the origin, 0.3 m at 80 m in the oracle itself: 4 sub-maps)
timeout 900 python benchmarks/config5.py --candidates 4096 --submaps 4 --method GICP 2> "$O/config5_GICP.err" | tail -1 > "$O/config5_GICP.json"
# kernel-level rows of the two engines batched in round 5 (missing 1): rocprofv3 kernel stats of the same commands, and the
# search kernel's launches in launch order
bash benchmarks/config5_trace.sh $R/c5trace_ICP ICP --candidates 4096 --submaps 64 --repeats 1 --no-cpu-baseline > "$O/c5trace_ICP.log" 2>&1
bash benchmarks/config5_trace.sh $R/c5trace_GICP GICP --candidates 4096 --submaps 4 --repeats 1 --no-cpu-baseline > "$O/c5trace_GICP.log" 2>&1
# every kernel of the bench step BY ITSELF (one part, chains serialised): the per-kernel own durations of the prefilter / index chain
bash benchmarks/serial_trace.sh $R/serial > "$O/serial.log" 2>&1
# k_knn_cov_tile's own counters (the STATS instantiation) for profiles/<round>_knn_tile_counters.txt
MI355_KNN_STATS=1 timeout 600 python bench.py --pairs 512 --steps 1 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> "$O/knn_stats.err"
# ---- NDT, the front end's default engine: its own bench line and profile
bash benchmarks/ndt_evidence.sh $R > "$O/ndt_evidence.log" 2>&1
# ---- the GPU test suite once, for the parity deltas it logs (tests/parity.py -> gpurun_out/parity_deltas.jsonl)
rm -f gpurun_out/parity_deltas.jsonl
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > "$O/pytest_gpu.txt" 2>&1
# distil on the box (the raw rocprofv3 output is hundreds of MB; gpurun copies back at most 64 MiB) and drop the raw traces
if EVIDENCE_DST="$O/profiles" python benchmarks/summarize_evidence.py $R > "$O/summarize.log" 2>&1; then
  rm -rf "$O/trace" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq1" "$O/pmc_sq2"
else
  echo "summarize_evidence.py FAILED: the raw traces stay under $O (too large to travel back: fix and re-run on a box)"
fi
du -sh "$O"; ls "$O" "$O/profiles"; tail -n 3 "$O"/*.err | tail -60; tail -5 "$O/pytest_gpu.txt"; cat "$O/summarize.log" | tail -5
