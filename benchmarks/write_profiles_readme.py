#!/usr/bin/env python3
"""Regenerate profiles/README.md from the committed profiles/<round>_* files (after summarize_evidence.py).

usage: python benchmarks/write_profiles_readme.py r01
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = os.path.join(ROOT, "profiles")
STEPS = 25  # bench defaults: 20 timed + 5 warm-up steps per traced run

PROGRESSION = """Progression over the round (1 GPU, pairs/s): 416 first correct path → 2 278 multi-level hash → 3 354 XCD-aware mapping →
4 229 branch-free median-network k-NN → 4 945 speculative all-trials LM → 6 203 (scene fix, LDS-staged centroids) → 6 786
8-wide candidate loads → 7 416 wave-cooperative tile k-NN fused with the covariance → 7 930 two-stream preparation →
8 690 long-lived pipelined linearize blocks → 8 920 rolled-trial error kernel → 9 000 24-byte records → 10.3 k at 512 pairs/step → 10.6 k 4-wide search scans → 11.5 k at 1 024 pairs/step.
"""


def load(name):
    return json.load(open(os.path.join(P, f"{R}_{name}.json")))


def main():
    b, tr, hi, i2, p128, tj = (load(n) for n in ("bench_default", "bench_under_trace", "bench_hostinputs", "bench_inflight2", "bench_128pairs", "traffic"))
    rows = list(csv.DictReader(open(os.path.join(P, f"{R}_kernel_stats_bench_default.csv"))))

    def k(name):
        r = [x for x in rows if name in x["Name"]]
        tot = sum(float(x["TotalDurationNs"]) for x in r)
        calls = sum(int(x["Calls"]) for x in r)
        return tot / 1e6 / STEPS, calls / STEPS, tot / max(1, calls) / 1e3

    lin = k("k_gicp_linearize")
    cpu, rf = b["cpu_baseline"], b["roofline"]
    pairs = b["config"]["pairs_per_gpu_per_step"]
    alg = rf["algorithmic_bytes_per_launch"]
    out = f"""# profiles/ — round 1 (MI355X, gfx950, ROCm 7.2)

All files come from ONE command family on a fresh 1-GPU box, `python bench.py` (defaults: {pairs} pairs/step, 20 steps,
5 warm-ups; 64 k-point pairs, distance filter 1.0 m + voxel 0.2 m, k = 20, exactly 20 LM iterations), run by
`benchmarks/collect_evidence.sh` and distilled by `benchmarks/summarize_evidence.py` / `write_profiles_readme.py`.

| file | what |
|---|---|
| `{R}_bench_default.json` | the JSON line of `python bench.py`: **{b['value']:.0f} pairs/s**, roofline frac {rf['frac']:.3f}; cpu_baseline {cpu['value']:.1f} pairs/s ({cpu['cores']} OpenMP threads, the fastest of 4/8/16/32/64/all on that host) |
| `{R}_kernel_stats_bench_default.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline` (bench line of that run: `{R}_bench_under_trace.json`, {tr['value']:.0f} pairs/s) |
| `{R}_pmc_fetch_write.csv` | per-kernel HBM traffic from two separate passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE …` / `--pmc WRITE_SIZE …` of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline` |
| `{R}_traffic.json` | the `k_gicp_linearize` row of the above, read by `bench.py` into `roofline.traffic` |
| `{R}_bench_hostinputs.json` | `python bench.py --host-inputs`: HOST buffers handed to the C-ABI every step — the PCIe-inclusive rate, never the headline |
| `{R}_bench_inflight2.json` | `python bench.py --inflight 2`: two batch objects on two streams used alternately |
| `{R}_bench_128pairs.json` | `python bench.py --pairs 128` |
| `{R}_other_configs.txt` | `python benchmarks/other_configs.py`: the BASELINE configs that are not the bench line |
| `{R}_prelim_kernel_stats_bench_64pairs.csv` | first-day snapshot (before any of the optimisations below), kept for the history |

## Dominant-kernel roofline (k_gicp_linearize, {pairs} pairs = {alg / 80 / 1e6:.2f} M source points per launch)

* algorithmic bytes per launch = 80 B × Σ source points = **{alg / 1e6:.0f} MB** (SURVEY §8d: src pt 16 + src cov 24 + tgt pt 16 + tgt cov 24);
* mean launch duration: **{rf['avg_launch_ms'] * 1e3:.1f} µs** from HIP events inside `bench.py` ({rf['launches']} launches); **{lin[2]:.1f} µs** in the rocprofv3
  kernel trace of the same command (`{R}_kernel_stats_bench_default.csv`, {int(round(lin[1] * STEPS))} launches) — they agree;
* achieved = {alg / 1e6:.0f} MB / {rf['avg_launch_ms'] * 1e3:.1f} µs = **{rf['achieved'] / 1e3:.2f} TB/s = {rf['frac'] * 100:.1f} % of the 8 TB/s HBM peak**
  (was 24.7 % at the start of the round: one block per 1 024 points, 32-byte padded covariance records);
* PMC traffic per launch: FETCH_SIZE {tj['fetch_size_kb_raw']:.0f} KB raw → ×2 (gfx950 counts the 128-B requests of 16-B/lane streaming reads as
  64 B, MI355X_MICROARCH.md §HBM) = {2 * tj['fetch_size_kb_raw'] * 1024 / 1e6:.0f} MB read, WRITE_SIZE {tj['write_size_kb']:.0f} KB = {tj['write_size_kb'] * 1024 / 1e6:.0f} MB written,
  **{tj['hbm_bytes_per_launch'] / 1e6:.0f} MB total = {tj['hbm_bytes_per_launch'] / alg:.2f} × algorithmic**.  Expected from the stored layout: 16 + 24 + 16 + 24 B read + 4 B
  correspondence + 24 B Mahalanobis written = 108 B/pt = 1.35 × — i.e. **no wasted re-reads**; on raw traffic the kernel
  moves {tj['hbm_bytes_per_launch'] / rf['avg_launch_ms'] / 1e9:.1f} TB/s.

## Where the step time goes ({pairs} pairs/step, {tr['ms_per_step']:.1f} ms/step under the tracer)

Targets and sources are prepared on two concurrent streams, so the kernel times below overlap (their sum exceeds the step).

| kernel | launches/step | ms/step (sum of durations) | note |
|---|---|---|---|
"""
    notes = [
        ("k_knn_cov_tile", "exact 20-NN + covariance of every point, wave-cooperative tiles (DESIGN §4.4); VALU-bound on the median network"),
        ("k_gicp_correspond", "iterations 0–3 do (nearly) full searches, afterwards ≈ 60 of 1.06 M points per 64 pairs are re-searched"),
        ("k_voxel_centroids", "LDS-staged serial fp32 sums per voxel (the order is part of the parity contract)"),
        ("k_sort_", "4 sorts × 4 passes of (hist, scan, scatter)"),
        ("k_gicp_error", "all 10 LM trial costs in one pass"),
        ("k_knn_hard", "the 1 % of points the tiles hand over, one wave per query"),
        ("k_gicp_linearize", "streaming, see above"),
        ("k_gicp_solve", "partials → H, b → all LM trial poses"),
        ("k_gicp_decide", "LM accept / reject walk"),
    ]
    for name, note in notes:
        t = k(name)
        out += f"| `{name}` | {t[1]:.0f} | {t[0]:.2f} | {note} |\n"
    by_threads = ", ".join(f"{t}: {v}" for t, v in cpu["pairs_per_s_by_threads"].items())
    out += f"""
## What bounds the other kernels (`{R}_pmc_valu.csv`, 128 pairs/step, per-launch PMC averages)

| kernel | bound | evidence |
|---|---|---|
| `k_knn_cov_tile` | VALU (median network: ~30 instructions per candidate and wave) | VALUBusy 64 %, 626 M VALU instructions per launch, 2 waves/SIMD (177 VGPRs, 16 KB LDS per wave); 26 % of the wave cycles waiting |
| `k_gicp_correspond` | latency (dependent loads of the per-lane voxel-hash walk) | VALUBusy 19 %, 71 % of the wave cycles waiting; early iterations dominated by the 16 % of points with no neighbour within 1 m |
| `k_gicp_error` | fp64 VALU + HBM, about balanced | VALUBusy 42 %, 60 B/pt read |
| `k_gicp_linearize` | HBM (see above) | VALUBusy 34 %, 58 % of the wave cycles waiting on memory |
| `k_sort_scatter` | HBM, scattered 4-byte writes | VALUBusy 20 %, 67 % waiting |

## Other measurements of the round

| what | value |
|---|---|
| `python bench.py` ({pairs} pairs/step) | {b['value']:.0f} pairs/s (boxes of the pool differ by several per cent) |
| `--pairs 128` | {p128['value']:.0f} pairs/s, roofline frac {p128['roofline']['frac']:.3f} |
| `--inflight 2` | {i2['value']:.0f} pairs/s; `k_gicp_linearize` then shares the GPU with the other batch's indexing: {i2['roofline']['avg_launch_ms'] * 1e3:.0f} µs per launch, frac {i2['roofline']['frac']:.3f} |
| `--host-inputs` ({pairs} pairs × 2 clouds × 1 MiB pageable host memory per step through `mi355_batch_set_sources/targets`) | {hi['value']:.0f} pairs/s |
| CPU oracle on the same host | {cpu['value']:.1f} pairs/s at {cpu['cores']} threads ({by_threads}) |
| GPU vs CPU pose of the same pair | {cpu['gpu_vs_cpu_pose_dt_m'] * 1e3:.2f} mm / {cpu['gpu_vs_cpu_pose_dr_rad']:.1e} rad |

{PROGRESSION}"""
    open(os.path.join(P, "README.md"), "w").write(out)
    print(out[:600])


if __name__ == "__main__":
    main()
