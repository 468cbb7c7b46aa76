#!/usr/bin/env python3
"""Regenerate profiles/README.md from the committed profiles/<round>_* files (after summarize_evidence.py).

usage: python benchmarks/write_profiles_readme.py r06
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(ROOT, "profiles")

NOTES = {
    "k_knn_cov_tile": "exact 20-NN + covariance of every point: 64-point tiles taken along a 2-D Hilbert curve over the cloud's two long axes, 16-lane sub-tile candidate lists in LDS, (k+2)-deep v_med3 key network (DESIGN §3); bound by its own instruction stream (valu_issue_frac), not by bytes",
    "k_gicp_correspond": "F3, incremental and exact: per-lane walks of the voxel hash, margins and margin-safe bits (DESIGN §3); latency-bound",
    "k_gicp_correspond_tile": "F3 while a pair still moves >= 3 cm per step (launched in the first four iterations; the fourth finds no such pair here): 64-point tiles in the k-NN tile order, packed distance/ordinal keys",
    "k_knn_hard": "the ~0.6 % of the points the tiles hand over (far field, ties), one wave per query",
    "sort_pairs(k_sort_*)": "per cloud three sorts - voxel index, Morton code of the hash, 2-D Hilbert index of the tile order: in-LDS tile sort (24 576 elements per block, all radix passes on chip) + merge-path merges for the raw 64 k-point clouds (round 6: the partner-less third tile is placed directly, the first merge round moves two thirds of the cloud)",
    "k_voxel_centroids": "fp32 centroid sums per voxel in ascending point order (the oracle's order); round 6: a lane takes four consecutive sorted elements (16-byte loads), folds the runs inside them in registers, only the run still open goes on through lane shifts",
    "k_gicp_linearize": "F4 + the sums F5 needs, streaming: 80 algorithmic B per source point; 72 model sums (8 x 8 outer products on fp64 MFMA + 20 registers) from which k_gicp_step forms H, b and every LM trial cost",
    "hash_build(k_hash_build|k_gather_insert)": "multi-level voxel hash sized to the occupied cells; round 6: one workgroup per cloud builds the table in LDS, one probe block (8 192 slots) at a time, and writes every slot once with coalesced stores (k_hash_build)",
    "k_gicp_step": "partials -> model -> H, b -> all LM trial poses and their costs; accept / reject walk (F5-F9)",
}


def pmc_row(rows, kernel):
    """row of <round>_pmc_valu.csv whose kernel name contains `kernel` -> {column: float}"""
    for r in rows:
        if kernel in r[0]:
            return r
    return None


def load(name):
    with open(os.path.join(P, f"{R}_{name}.json")) as f:
        return json.load(f)


def maybe(name):
    p = os.path.join(P, f"{R}_{name}.json")
    return json.load(open(p)) if os.path.exists(p) else None


def main():
    b, tr, tj = load("bench_default"), load("bench_under_trace"), load("traffic")
    hi, i2, i3, p128, ser, sor = (maybe(n) for n in ("bench_hostinputs", "bench_inflight2", "bench_inflight3", "bench_128pairs", "bench_serial_streams", "bench_sor"))
    p1k, p2k, p4k = maybe("bench_1024pairs"), maybe("bench_2048pairs"), maybe("bench_4096pairs")
    ndt = maybe("ndt_batch")
    rows = list(csv.DictReader(open(os.path.join(P, f"{R}_kernel_stats_bench_default.csv"))))
    tot_ns = sum(float(x["TotalDurationNs"]) for x in rows)
    rf, cpu, e2e = b["roofline"], b["cpu_baseline"], b["roofline"]["end_to_end"]
    pairs = b["config"]["pairs_per_gpu_per_step"]
    dom = rf["kernel"]
    top = next(x for x in rows if dom in x["Name"])
    live = rf["timed_region_event_brackets"]
    one = cpu["one_process"]["pairs_per_s_by_omp_threads"]
    per_launch = int(round(2 * pairs / rf["by_kernel"][dom]["launches_per_step"])) if dom in rf["by_kernel"] else pairs  # clouds per launch of the dominant kernel (a batch runs in parts)
    issue = maybe("issue") or {"kernels": {}}
    ik = issue["kernels"].get(dom.split("(")[0], {})
    out = f"""# profiles/ — round {int(R[1:])} (MI355X, gfx950, ROCm 7.2)

Everything here comes from ONE command family on a fresh 1-GPU box — `python bench.py` (defaults: {pairs} pairs per step = per
batch, one batch at a time, 20 timed steps, 5 warm-ups; 64 k-point pairs, distance filter 1.0 m + voxel 0.2 m, k = 20, exactly
20 LM iterations) — run by `benchmarks/collect_evidence.sh {R}` and distilled by `benchmarks/summarize_evidence.py` /
`write_profiles_readme.py`.  The files of earlier rounds (`r01_*` … `r05_*`) are kept for the history; their experiment logs
(`r03_knn_*`, `r03_corr_*`, `r04_parts_ab.txt`, `r05_fold_ab.txt`, `r05_ndt_variants.txt` …) are what `docs/history.md` cites for the kernel variants that did not pay.

| file | what |
|---|---|
"""
    files = [
        ((f"{R}_bench_default.json",), f"the JSON line of `python bench.py`: **{b['value']:.0f} pairs/s**, {b['ms_per_step']:.1f} ms/step, {b['config'].get('hbm_bytes_in_use', 0) / 1e9:.0f} GB of HBM in use; `roofline` on `{dom}` (frac {rf['frac']:.4f}"
         + (f", `valu_issue_frac` {rf['valu_issue_frac']:.2f}" if rf.get('valu_issue_frac') else "") + f"), `roofline.end_to_end` {e2e['frac']:.3f}; `cpu_baseline` {cpu['value']:.1f} pairs/s on all {cpu['cores']} hardware threads of the host ({cpu['cpu_model']})"),
        ((f"{R}_kernel_stats_bench_default.csv",), f"`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-h2d` (its bench line: `{R}_bench_under_trace.json`, {tr['value']:.0f} pairs/s)"),
        ((f"{R}_kernel_stats_serial_2048pairs.csv", f"{R}_bench_serial_one_part_under_trace.json"), "every kernel of the step BY ITSELF: `benchmarks/serial_trace.sh` = the bench at 2 048 pairs, ONE part, the preparation chains one after the other on the batch's stream, under `rocprofv3 --kernel-trace --stats` - the per-kernel own durations (one launch = 2 048 clouds) the prefilter / index chain is judged by"),
        ((f"{R}_pmc_fetch_write.csv",), "per-kernel HBM traffic from two separate passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE …` / `--pmc WRITE_SIZE …` of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d`"),
        ((f"{R}_traffic.json",), f"the `{tj['kernel']}` row of the above (FETCH_SIZE × 2 + WRITE_SIZE, MI355X_MICROARCH.md §HBM), read by `bench.py` into `roofline.traffic`"),
        ((f"{R}_pmc_valu.csv", f"{R}_issue.json"), f"SQ counters per launch of every hot kernel at the bench's full batch ({pairs} pairs): `SQ_INSTS_VALU`, `SQ_ACTIVE_INST_VALU`, `SQ_INSTS_LDS`, `SQ_WAIT_ANY`, `SQ_WAVE_CYCLES`, `SQ_BUSY_CYCLES`, `SQ_LDS_BANK_CONFLICT` … from two `--pmc` passes (`--kernel-trace` only) of `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2d`; the last two columns - resident waves per SIMD and `valu_issue_frac`, the share of the chip's VALU issue slots a kernel fills - also as JSON, read by `bench.py` into `roofline.valu_issue_frac` / `by_kernel[*].valu_issue_frac`"),
        ((f"{R}_ndt_batch.json", f"{R}_ndt_batch_under_trace.json", f"{R}_kernel_stats_ndt.csv", f"{R}_pmc_ndt.csv", f"{R}_ndt_issue.json"), "**NDT_OMP, the front end's default engine** (`benchmarks/ndt_evidence.sh`): the line of `python benchmarks/ndt_batch.py --config3` (1 024 scan-vs-sub-map pairs per step, 2.0 m cells; + BASELINE config 3), the rocprofv3 kernel stats of the same command, per-kernel FETCH / WRITE / SQ counters from separate `--pmc` passes and the issue fractions derived from them"),
        ((f"{R}_config5_1gpu.json",), "BASELINE config 5 at its stated size — 4 096 candidates × 64 sub-maps — through `mi355_multi_batch_register` on ONE GPU with 2 lanes, host views in (pinned staging), results out: `benchmarks/config5.py --candidates 4096 --submaps 64 --multi-lanes 2`"),
        ((f"{R}_config5_FAST_GICP.json", f"{R}_config5_ICP.json", f"{R}_config5_NDT_OMP.json", f"{R}_config5_GICP.json"), "the same candidate list through one `BatchRegistration` per rank (`benchmarks/config5.py --method …`: uploads + registration + fitness + read-back, the shorter of two runs on one batch object), every engine `GraphBasedSLAM::get_registration` builds — ICP is that node's yaml default; each line carries a `cpu_baseline` (the oracle on four of the same candidates, 4 OpenMP threads).  NDT_OMP on 8 and BFGS GICP on 4 sub-maps near the map origin (DESIGN §3; `tests/test_oracle_ndt.py::test_far_from_the_map_origin_*` pins the NDT half of that claim against an all-fp64 NumPy NDT)"),
        ((f"{R}_kernel_stats_icp.csv", f"{R}_icp_launches.txt", f"{R}_config5_ICP_under_trace.json"), "**kernel-level rows of batched ICP** (the back end's default engine): `rocprofv3 --kernel-trace --stats` of `benchmarks/config5.py --method ICP --candidates 4096 --submaps 64` (`benchmarks/config5_trace.sh`), and the durations of `k_icp_correspond_tile` / `k_icp_step` in launch order"),
        ((f"{R}_kernel_stats_bfgs.csv", f"{R}_bfgs_launches.txt", f"{R}_config5_GICP_under_trace.json"), "the same for pclomp's BFGS GICP (`--method GICP --submaps 4`): `k_bfgs_move` + `k_moved_correspond_tile` (the tiles, for the pairs that start an outer iteration), `k_bfgs_correspond`, `k_bfgs_cost`, `k_bfgs_advance`"),
        ((f"{R}_knn_tile_counters.txt",), "`k_knn_cov_tile`'s own counters (`MI355_KNN_STATS=1`: candidates gathered / ranked per lane, rounds, phase clocks, hand-over) beside the PMC instruction counts"),
        ((f"{R}_parity_deltas.txt",), "worst GPU-vs-oracle pose difference per test family of one `pytest -m gpu` run (BASELINE.md §4)"),
        ((f"{R}_bench_sor.json",), "`python bench.py --sor --pairs 2048`: the reference's launch-file prefilter (leaf 0.1 m + StatisticalOutlierRemoval 30 / 1.2) — a different workload, never the headline; carries the SOR row"),
        ((f"{R}_bench_hostinputs.json",), "`python bench.py --host-inputs --pairs 2048`: HOST buffers (pageable memory) handed to the C-ABI every step — the PCIe-inclusive rate, never the headline"),
        ((f"{R}_bench_serial_streams.json",), "the default with `MI355_BATCH_SIDE_STREAMS=0`: the preparation chains one after the other"),
        ((f"{R}_bench_128pairs.json", f"{R}_bench_1024pairs.json", f"{R}_bench_2048pairs.json"), "`python bench.py --pairs N`"),
        ((f"{R}_other_configs.txt", f"{R}_single_pair.txt", f"{R}_stream_profile.txt"), "the BASELINE configs that are not the bench line (`benchmarks/other_configs.py --scans 200`, with a `cpu_baseline` for configs 3 and 4), one 64 k pair, a 200-scan odometry stream (FAST_GICP, then NDT_OMP)"),
    ]
    for names, what in files:
        have = [n for n in names if os.path.exists(os.path.join(P, n))]
        if have:
            out += "| " + ", ".join(f"`{n}`" for n in have) + f" | {what} |\n"
    pv = list(csv.reader(open(os.path.join(P, f"{R}_pmc_valu.csv")))) if os.path.exists(os.path.join(P, f"{R}_pmc_valu.csv")) else []
    issue_txt = (f"`{R}_issue.json`: {ik.get('resident_waves_per_simd', 0):.1f} resident waves per SIMD, each issuing VALU {100 * ik.get('valu_issue_share_per_wave', 0):.0f} % of its cycles and waiting "
                 f"{100 * ik.get('wait_share_per_wave', 0):.0f} % - **{ik.get('valu_issue_frac', 0):.2f} of the chip's VALU issue slots**") if ik else f"`{R}_pmc_valu.csv`"
    out += f"""
## The roofline block, recomputed from these files

* dominant kernel family = the one with the largest own duration × launches per step: **`{dom}`** — also the top row of
  `{R}_kernel_stats_bench_default.csv` (`{top['Name'].split('(')[0].replace('void ', '')}`, {100 * float(top['TotalDurationNs']) / tot_ns:.1f} % of the kernel time).
* algorithmic bytes per launch = 40 B per cloud point (16 B point read + 24 B covariance written, SURVEY §8d) × Σ points of the
  {per_launch} clouds of a launch (the batch's {pairs} sources or targets run in {pairs // per_launch} parts, DESIGN §2) = **{rf['algorithmic_bytes_per_launch'] / 1e6:.0f} MB**;
* launch duration: **{rf['avg_launch_ms']:.2f} ms** — HIP events on the launching stream inside `bench.py`, around {rf['launches_measured']} launches that have the GPU
  to themselves (two extra steps after the timed region, preparation chains serialised); **{float(top['AverageNs']) / 1e6:.2f} ms** average over the {top['Calls']} launches
  of the rocprofv3 kernel trace of the same command (min {float(top['MinNs']) / 1e6:.2f}, max {float(top['MaxNs']) / 1e6:.2f}) — the two agree.  The event brackets of the TIMED
  region (`roofline.timed_region_event_brackets`) read {live['avg_ms']:.2f} ms: the preparation chains run on two concurrent side streams, but since
  round 3 a chain that comes second starts its k-NN when the other's is done (`hipStreamWaitEvent`), so a launch neither queues behind
  nor interleaves with the other one (at 2 048 clouds the two used to interleave and rocprofv3 saw 33.5 ms per launch: 3 % more pairs/s,
  and a duration that was nobody's own); `frac` is still taken from the launches that have the GPU to themselves;
* achieved = {rf['algorithmic_bytes_per_launch'] / 1e6:.0f} MB / {rf['avg_launch_ms']:.2f} ms = **{rf['achieved']:.1f} GB/s = {100 * rf['frac']:.2f} % of the 8 TB/s HBM peak**: the kernel is bound by its own instruction stream and the latencies inside it
  ({issue_txt}), the HBM roofline is simply the wrong ceiling for it, and the
  bench says so instead of picking a flattering kernel;
* PMC traffic per launch: FETCH_SIZE {tj['fetch_size_kb_raw'] / 1e3:.0f} MB raw × 2 + WRITE_SIZE {tj['write_size_kb'] / 1e3:.0f} MB = **{tj['hbm_bytes_per_launch'] / 1e6:.0f} MB = {tj['hbm_bytes_per_launch'] / rf['algorithmic_bytes_per_launch']:.2f} × algorithmic**
  (the ×2 is the guide's correction for streaming reads; for this kernel's gathers it is an upper bound);
* `roofline.end_to_end`: algorithmic bytes of the whole step ({e2e['algorithmic_bytes_per_step'] / 1e9:.2f} GB: prefilter 32 B/raw point, covariances 40 B/point,
  GICP 80 B/source point/iteration) ÷ {b['ms_per_step']:.1f} ms ÷ 8 TB/s = **{e2e['frac']:.3f}**.

## Where the step goes ({pairs} pairs/step)

HIP-event totals per kernel family (`roofline.by_kernel`).  "own" = the kernel with nothing beside it (the rocprofv3-agreeing duration);
"timed region" = brackets on the two concurrent preparation streams (they overlap: their sum exceeds the step).

| kernel family | launches/step | own: ms per launch | × launches = ms/step | frac of HBM peak, own (algorithmic bytes) | valu_issue_frac (`{R}_issue.json`) | timed-region bracket ms/step | note |
|---|---|---|---|---|---|---|---|
"""
    order = sorted(rf["by_kernel"].items(), key=lambda kv: -kv[1].get("isolated_avg_launch_ms", kv[1]["avg_launch_ms"]) * kv[1]["launches_per_step"])
    total_own = 0.0
    for name, v in order:
        frac = v.get("isolated_frac_of_hbm_peak")
        iso = v.get("isolated_avg_launch_ms", v["avg_launch_ms"])
        total_own += iso * v["launches_per_step"]
        kname = {"sort_pairs": "k_sort_tile", "hash_build": "k_hash_build"}.get(name.split("(")[0], name.split("(")[0])
        vi = issue["kernels"].get(kname, {}).get("valu_issue_frac")
        out += (f"| `{name}` | {v['launches_per_step']:.0f} | {iso:.3f} | {iso * v['launches_per_step']:.2f} | "
                + (f"{frac:.4f} ({v['bytes_per_unit']})" if frac is not None else "–") + " | " + (f"{vi:.2f}" + (" (`k_sort_tile`)" if kname == "k_sort_tile" else "") if vi is not None else "–")
                + f" | {v['total_ms_per_step']:.2f} | {NOTES.get(name, '')} |\n")
    out += (f"\nSum of the own durations: {total_own:.1f} ms against {b['ms_per_step']:.1f} ms per step.  The families above leave out the small kernels (keys, scans, bounds, copies: "
            f"see the kernel-stats CSV); what makes the step SHORTER than the sum is the pipeline inside the batch - its parts' preparation chains and LM loops run on "
            f"internal streams and fill each other's tails (DESIGN §2; with the chains serialised the step is the last line of this paragraph).\n")
    if ser:
        out += f"One batch of {pairs} pairs with the preparation chains serialised (`{R}_bench_serial_streams.json`): {ser['ms_per_step']:.1f} ms/step.\n"
    oc_path = os.path.join(P, f"{R}_other_configs.txt")
    ocj = json.loads(open(oc_path).read()) if os.path.exists(oc_path) else {}
    sp = open(os.path.join(P, f"{R}_single_pair.txt")).read().strip().replace("\n", "; ") if os.path.exists(os.path.join(P, f"{R}_single_pair.txt")) else ""
    stp_all = "; ".join(ln for ln in open(os.path.join(P, f"{R}_stream_profile.txt")).read().strip().splitlines() if "scans" in ln) if os.path.exists(os.path.join(P, f"{R}_stream_profile.txt")) else ""
    out += f"""
## Other measurements of the round

| what | value |
|---|---|
| `python bench.py` ({pairs} pairs per step, one batch at a time) | {b['value']:.0f} pairs/s, {b['ms_per_step']:.1f} ms/step (round 5: 28 103 at the same batch size; round 4: 26 668 at 2 048 pairs per step; round 3: 18 644; round 2: 14 383 and round 1: 11 572, both at 1 024; boxes of the pool differ by a few per cent) |
"""
    if i2:
        out += f"| `--inflight 2 --pairs 1024` (two batches in flight, the same memory) | {i2['value']:.0f} pairs/s — the LM iterations of one batch overlap the indexing of the next |\n"
    if i3:
        out += f"| `--inflight 3 --pairs 1024` | {i3['value']:.0f} pairs/s |\n"
    if p128:
        out += f"| `--pairs 128` | {p128['value']:.0f} pairs/s |\n"
    if p1k:
        out += f"| `--pairs 1024` (the default of rounds 1-3; DESIGN.md quotes per-launch kernel figures at this size) | {p1k['value']:.0f} pairs/s, {p1k['ms_per_step']:.1f} ms/step |\n"
    if p2k:
        out += f"| `--pairs 2048` (the default of rounds 3-4) | {p2k['value']:.0f} pairs/s, {p2k['ms_per_step']:.1f} ms/step |\n"
    if p4k:
        out += f"| `--pairs 4096` | {p4k['value']:.0f} pairs/s |\n"
    if hi:
        out += f"| `--host-inputs --pairs {hi['config']['pairs_per_gpu_per_step']}` ({hi['config']['pairs_per_gpu_per_step']} pairs × 2 clouds × 1 MiB pageable host memory per step through `mi355_batch_set_sources/targets`) | {hi['value']:.0f} pairs/s — the PCIe-inclusive rate |\n"
        hi2 = maybe("bench_hostinputs_inflight2")
        if hi2:
            out += f"| `--host-inputs --inflight 2 --pairs 1024` (the uploads of one batch under the kernels of the other) | {hi2['value']:.0f} pairs/s, {hi2['ms_per_step']:.1f} ms per 1 024-pair step |\n"
    if sor:
        srow = next((v for k, v in sor["roofline"]["by_kernel"].items() if k.startswith("sor(")), None)
        out += (f"| `--sor` (leaf 0.1 m + SOR 30 / 1.2: {sor['config']['points_after_voxel_src_mean']:.0f} points per cloud survive) | {sor['value']:.0f} pairs/s, {sor['ms_per_step']:.1f} ms/step; "
                + (f"StatisticalOutlierRemoval of {sor['config']['pairs_per_gpu_per_step']} clouds: {srow.get('isolated_avg_launch_ms', srow['avg_launch_ms']) * srow['launches_per_step'] / 2:.1f} ms "
                   f"(own durations of its {srow['launches_per_step'] / 2:.0f} brackets: sort, hash, k = 31 search, threshold + compaction)" if srow else "") + " |\n")
    out += f"""| CPU oracle on the same host ({cpu['cpu_model']}, {cpu['host_hw_threads']} hardware threads) | whole host, {cpu['cores'] // 4} workers × 4 OpenMP threads on independent pairs: **{cpu['value']:.1f} pairs/s** (`cpu_baseline.value`); one process, OpenMP inside a pair: {', '.join(f'{t} threads: {v}' for t, v in one.items())} pairs/s (4 = the reference's configuration, `lidar_scan_matcher.param.yaml:10`) |
| GPU vs CPU pose of the same world-transformed pair | {cpu['gpu_vs_cpu_pose_dt_m'] * 1e9:.0f} nm / {cpu['gpu_vs_cpu_pose_dr_rad']:.1e} rad (bench.py aborts above 1 mm / 1e-4 rad) |
| one 64 k pair (`{R}_single_pair.txt`) | {sp} |
| 200-scan odometry stream, default (reference-arithmetic) mode (`{R}_stream_profile.txt`) | {stp_all} |
"""
    if ocj:
        c4 = ocj.get("config4_stream_FAST_GICP", {})
        c4o = ocj.get("config4_stream_FAST_GICP_carried_covariances_optin", {})
        c4n = ocj.get("config4_stream_NDT_OMP", {})
        c3 = ocj.get("config3_ndt_1M_submap", {})
        c3b = next((v for k, v in c3.items() if k.startswith("set_target_ms")), 0.0)
        out += f"""| config 3, NDT: 65 k-point scan vs 1.05 M-point sub-map, 1.0 m cells (`other_configs.py`: the FIRST set_target, allocations included; warm: the NDT section below) | target build {c3b:.1f} ms, align {c3.get('align_ms_median', 0):.2f} ms ({c3.get('iterations')} iterations) |
| config 4, 200-scan stream, window 20, FAST_GICP, default: covariances recomputed over the sub-map (the reference's arithmetic) | {c4.get('ms_per_scan', 0):.2f} ms/scan, end pose {c4.get('end_pose_err_vs_truth', [0])[0]:.2f} m off after 160 m |
| … opt-in: key-frame covariances carried (C' = R C Rᵀ) | {c4o.get('ms_per_scan', 0):.2f} ms/scan, {c4o.get('end_pose_err_vs_truth', [0])[0]:.2f} m |
| … NDT_OMP (2.0 m cells, the node's default) | {c4n.get('ms_per_scan', 0):.2f} ms/scan, {c4n.get('end_pose_err_vs_truth', [0])[0]:.2f} m (the 0.5 % scale bias of NDT on this scene, docs/history.md §5) |
| config 4 at its stated 1 000 scans | `tests/test_gpu_fullsize.py::test_config4_at_its_stated_size_1000_scans` (both ground models): see `{R}_parity_deltas.txt` summary rows |
"""
    c5m, c5m4 = maybe("config5_1gpu"), maybe("config5_1gpu_4lanes")
    if c5m and "pairs_per_s_including_uploads" in c5m:
        out += (f"| config 5 at its stated size (4 096 candidates × 64 sub-maps) through `mi355_multi_batch_register`, ONE GPU, host views in | "
                f"{c5m['pairs_per_s_including_uploads']:.0f} pairs/s with {c5m['lanes']} lanes"
                + (f", {c5m4['pairs_per_s_including_uploads']:.0f} with {c5m4['lanes']}" if c5m4 else "")
                + f" — uploads ({c5m['bytes_uploaded_per_call'] / 1e9:.2f} GB per call), prefilter, indexing, registration (early stop, {c5m['mean_iterations']:.1f} iterations on average) and fitness included; "
                f"{c5m['within_5cm_of_truth']} of {c5m['candidates']} within 5 cm of the truth, median {c5m['median_pose_err_m'] * 100:.1f} cm |\n")
    for m in ("FAST_GICP", "ICP", "NDT_OMP", "GICP"):
        c5e = maybe(f"config5_{m}")
        if c5e and "pairs_per_s_registration" in c5e:
            out += (f"| config 5 ({c5e['candidates']} candidates × {c5e['submaps']} sub-maps) through one `BatchRegistration`, engine {m} (`{R}_config5_{m}.json`) | {c5e['pairs_per_s_registration']:.0f} pairs/s of registration "
                    f"({c5e['seconds_register_max_over_ranks'] * 1e3:.0f} ms for {c5e['candidates']} candidates, {c5e['mean_iterations']:.1f} iterations on average); "
                    f"{c5e['accepted(fitness<0.3)']} accepted, median {c5e['median_pose_err_m'] * 100:.1f} cm from the truth"
                    + (f"; CPU oracle on the same host: {c5e['cpu_baseline']['value']:.2f} pairs/s with {c5e['cpu_baseline']['cores']} OpenMP threads (four of the same candidates, one after the other)" if c5e.get('cpu_baseline') else "")
                    + " |\n")
    for tag, title, pats in (("icp", "batched ICP (the back end's default engine)", ("k_icp_",)), ("bfgs", "batched BFGS GICP", ("k_bfgs_",))):
        pth = os.path.join(P, f"{R}_kernel_stats_{tag}.csv")
        if not os.path.exists(pth):
            continue
        krows = [x for x in csv.DictReader(open(pth)) if any(pp in x["Name"] for pp in pats)]
        if krows:
            out += (f"| kernel rows of {title} on config 5 (`{R}_kernel_stats_{tag}.csv`, one registration of 4 096 candidates under rocprofv3) | "
                    + "; ".join(f"`{x['Name'].split('(')[0].replace('void ', '').replace('mi355::', '')}` {x['Calls']} launches, {float(x['TotalDurationNs']) / 1e6:.1f} ms in all "
                                f"({float(x['AverageNs']) / 1e3:.0f} us on average, longest {float(x['MaxNs']) / 1e6:.2f} ms)" for x in krows[:4]) + " |\n")
    hb = b.get("h2d_inclusive")
    if hb:
        out += f"| `h2d_inclusive` block of the default bench line (3 steps from pageable host buffers after the timed region, results bit-equal) | {hb['value']:.0f} pairs/s, {hb['ms_per_step']:.1f} ms/step |\n"
    if ndt:
        k = {x["kernel"].split("(")[0]: x for x in ndt["by_kernel"]}
        d = k["k_ndt_derivatives"]
        if not ndt["roofline"].get("valu_issue_frac"):
            ni = maybe("ndt_issue") or {"kernels": {}}
            ndt["roofline"]["valu_issue_frac"] = ni["kernels"].get("k_ndt_derivatives", {}).get("valu_issue_frac")
        c3n = ndt.get("config3") or {}
        out += f"""
## NDT_OMP — the front end's configured default engine (`lidar_scan_matcher.param.yaml:4`)

`python benchmarks/ndt_batch.py --config3` (`{R}_ndt_batch.json`; not the headline metric): one step = {ndt['config']['workload'].split(';')[-1].strip()}, sub-maps of
{ndt['config']['key_frames_per_submap']} voxelised key frames ({ndt['config']['target_points_mean']:.0f} points) -> leaves; raw 64 k scans -> prefilter ({ndt['config']['source_points_mean']:.0f} points) -> Newton iterations to
convergence ({ndt['iterations_mean']} on average) -> poses.  **{ndt['value']:.0f} pairs/s, {ndt['ms_per_step']:.1f} ms per step** (round 5: 54 443 pairs/s; the round-4 library on the same script: 2 070 pairs/s, 495 ms);
every pair converged, median {ndt['pose_error_vs_truth']['median_m'] * 100:.1f} cm from the truth; the pairs the oracle registers again differ from it by <= {ndt['gpu_vs_oracle']['max_dt_m'] * 1e3:.2f} mm
with the same iteration counts.  The roofline row of `k_ndt_derivatives`:

* COMPULSORY bytes per step = {d['bytes_per_unit'].split(';')[0]} = **{d['algorithmic_bytes_per_step'] / 1e9:.1f} GB**
  ({d['bytes_per_unit'].split(';')[-1].strip()});
* own duration = {d['ms_per_step_own']:.2f} ms per step over {d['launches_per_step']:.0f} launches (HIP events, chains serialised; `{R}_kernel_stats_ndt.csv`, whose trace also holds the
  config-3 launches, agrees on the per-launch average);
* achieved = {d['achieved_GBs']:.0f} GB/s = **{d['frac_of_hbm_peak']:.3f} of the 8 TB/s peak** - far from the HBM ceiling because bytes do not bound this kernel: **`valu_issue_frac` = {(ndt['roofline'].get('valu_issue_frac') or 0):.2f}**
  of the chip's VALU issue slots (`{R}_ndt_issue.json`, from the SQ counters of `{R}_pmc_ndt.csv`) is the fraction that says how close to ITS ceiling the kernel runs;
* SURVEY §8d prices the pass at 268 B per source point and evaluation (16 + 7 cells x 36): {d.get('survey_pricing', {}).get('algorithmic_bytes_per_step', 0) / 1e9:.1f} GB per step, which the kernel would "move" at
  {d.get('survey_pricing', {}).get('achieved_GBs_if_all_of_it_came_from_hbm') or 0:.0f} GB/s - above the HBM peak, i.e. not a roofline fraction: the seven leaves of a point come from L2 (`{R}_pmc_ndt.csv`: FETCH_SIZE x 2 of a
  launch = the source points alone).  Kept in the line as `by_kernel[0].survey_pricing`, never as `frac`.

| NDT kernel family | launches/step | own ms/step | algorithmic bytes | frac of HBM peak |
|---|---|---|---|---|
"""
        for x in ndt["by_kernel"]:
            out += (f"| `{x['kernel']}` | {x['launches_per_step']:.0f} | {x['ms_per_step_own']:.2f} | " + (f"{x['algorithmic_bytes_per_step'] / 1e9:.2f} GB ({x['bytes_per_unit'].split(':')[0].split(';')[0]})" if x.get('algorithmic_bytes_per_step') else "–")
                    + " | " + (f"{x['frac_of_hbm_peak']:.3f}" if x.get('frac_of_hbm_peak') is not None else "–") + " |\n")
        if c3n:
            out += (f"\nBASELINE config 3 (one {c3n['source_points']}-point scan vs a {c3n['target_points']}-point sub-map, 1.0 m cells; warm, median of 7): `setInputTarget` from a host cloud "
                    f"{c3n['set_target_from_host_ms(median,min)'][0]:.2f} ms, from a device cloud {c3n['set_target_device_cloud_ms(median,min)'][0]:.2f} ms (round 4: 3.5 / 2.4 ms on the same script; 9.4 ms with its first-call allocations), "
                    f"`align` {c3n['align_ms(median,min)'][0]:.2f} ms ({c3n['iterations']} iterations, {c3n['evaluations']} evaluations, no read-back between them).\n")

    open(os.path.join(P, "README.md"), "w").write(out)
    print(out[:600])


if __name__ == "__main__":
    main()
