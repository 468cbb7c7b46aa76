#!/usr/bin/env python3
"""NDT_OMP — the reference front end's CONFIGURED DEFAULT engine (lidar_scan_matcher.param.yaml:4, built at
lidar_scan_matcher.cpp:55-72) — measured like the headline bench measures FAST_GICP.  Not the headline metric: its JSON line
says `"not_the_headline": true`.

    python benchmarks/ndt_batch.py                       1 024 scan-vs-sub-map pairs per step, 2.0 m cells (yaml :15)
    python benchmarks/ndt_batch.py --config3             + BASELINE configs[2]: one 64 k scan vs a ~1 M-point sub-map, 1.0 m cells

One step = one pass of the NDT hot path over a batch of independent pairs, inputs resident in HBM: sub-maps (5 key frames of
voxelised scans, map frame) -> VoxelGridCovariance leaves of every target (N1); raw 64 k-point scans -> distance filter 1.0 m +
VoxelGrid 0.2 m (what points_prefiltering publishes) -> Newton iterations with computeDerivatives over DIRECT7 neighbourhoods
(N2 - N4, ndt_omp's line search) until every pair has converged -> poses copied back.  Pair p is base pair p % bases (its own
street, trajectory and noise) under its own world transform (re-bins every voxel grid) and its own sensor-frame rotation.

The line carries `by_kernel` (HIP events recorded by the library around the launches of each family) priced at SURVEY 8(d)'s
algorithmic bytes: the derivative pass at 268 B per source point and evaluation (16 + 7 cells x 36), the target build at 16 B per
target point.  The run FAILS when the poses it timed are wrong (vs ground truth, and vs the oracle on the same inputs)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
N_ORACLE = 4


def base_jobs(n_base, keyframes):
    """(cast jobs, per-base poses): base i = `keyframes` key-frame scans 1 m apart along its own stretch of a street + the next scan
    0.8 m further on (the scan matcher's geometry: key frame every >= 1.0 m, a scan every 0.8 m)."""
    from lidar_graph_slam_amd import synth

    jobs, meta = [], []
    for i in range(n_base):
        r = np.random.default_rng(7000 + 31 * i)
        x0, y0, yaw0 = r.uniform(0.0, 30.0), r.uniform(-1.0, 1.0), r.uniform(-0.05, 0.05)
        length = 60.0 + 10.0 * (i % 4)  # four different streets (box placement follows the length)
        K = [synth.se3(x0, y0, 0.0, 0.0, 0.0, yaw0) @ synth.se3(1.0 * j, 0.0, 0.0) for j in range(keyframes)]
        S = K[-1] @ synth.se3(r.uniform(0.6, 0.9), r.uniform(-0.08, 0.08), 0.0, 0.0, 0.0, np.deg2rad(r.uniform(-1.5, 1.5)))
        for j, P in enumerate(K):
            jobs.append((length, P, 4000 + 100 * i + j, 1024, 30.0))
        jobs.append((length, S, 4000 + 100 * i + 99, 1024, 30.0))
        meta.append(dict(K=K, S=S, guess=K[-1]))  # the guess is the previous pose (lidar_scan_matcher.cpp:165)
    return jobs, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bases", type=int, default=16)
    ap.add_argument("--keyframes", type=int, default=5)
    ap.add_argument("--resolution", type=float, default=2.0, help="NDT cell edge (lidar_scan_matcher.param.yaml:15)")
    ap.add_argument("--leaf", type=float, default=0.2)
    ap.add_argument("--config3", action="store_true", help="also time BASELINE configs[2]: one 64 k scan vs a ~1 M-point sub-map at 1.0 m cells")
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()

    import multiprocessing as mp

    from lidar_graph_slam_amd import synth

    jobs, meta = base_jobs(args.bases, args.keyframes)
    c3_jobs = []
    if args.config3:  # 16 casts along a 16 m straight trajectory + the scan
        c3_poses = [synth.se3(1.0 * i, 0.02 * i, 0, 0, 0, 0.002 * i) for i in range(16)]
        c3_true = synth.se3(7.5, 0.15, 0, 0, 0, 0.015)
        c3_jobs = [(100.0, P, 200 + i, 1024, None) for i, P in enumerate(c3_poses)] + [(100.0, c3_true, 999, 1024, None)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(max(1, min(64, (os.cpu_count() or 8) - 1))) as pool:  # before anything touches the GPU
        raw = pool.map(synth.cast_job, jobs + c3_jobs, chunksize=2)
    t_cast = time.perf_counter() - t0

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("ndt_batch.py needs an MI355X: there is no CPU fallback for the product path")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)

    from lidar_graph_slam_amd import _lib as L
    from lidar_graph_slam_amd.context import Context
    from lidar_graph_slam_amd.filters import PointsPreFiltering
    from lidar_graph_slam_amd.registration import BatchRegistration, NormalDistributionsTransform

    ctx = Context(0)
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    ctx.set_stream(main_stream.cuda_stream)
    dev_name, cus = ctx.device_info()
    pre = PointsPreFiltering(leaf_size=args.leaf, mean_k=0, min_distance_cloud=1.0, ctx=ctx)

    def place(c, T):
        o = c.copy()
        o[:, :3] = (c[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        return o

    # ---- base pairs: sub-map (map frame) + raw scan (sensor frame)
    per = args.keyframes + 1
    base_sub, base_scan = [], []
    for i, m in enumerate(meta):
        casts = raw[i * per:(i + 1) * per]
        base_sub.append(np.concatenate([place(pre.callback_sensor_points(c), K) for c, K in zip(casts[:-1], m["K"])], axis=0))
        base_scan.append(casts[-1])
    P, nb = args.pairs, args.bases
    cap = max(65536, max(len(s) for s in base_sub))
    cap = (cap + 255) // 256 * 256
    g = np.random.default_rng(1234)
    tgt_d = torch.zeros((P, cap, 4), dtype=torch.float32, device=dev)
    src_d = torch.zeros((P, cap, 4), dtype=torch.float32, device=dev)
    sub_t = [torch.from_numpy(s).to(dev) for s in base_sub]
    scan_t = [torch.from_numpy(s).to(dev) for s in base_scan]
    counts_t = np.zeros(P, np.int32)
    counts_s = np.zeros(P, np.int32)
    truth, guesses, Ws, Vs = [], [], [], []
    for p in range(P):
        k = p % nb
        if p < nb:
            W, V = np.eye(4), np.eye(4)
        else:
            W = synth.se3(g.uniform(-50, 50), g.uniform(-50, 50), 0.0, 0.0, 0.0, g.uniform(0, 2 * np.pi))
            V = synth.se3(0, 0, 0, 0, 0, g.uniform(0, 2 * np.pi))
        Wt, Vt = torch.from_numpy(W.astype(np.float32)).to(dev), torch.from_numpy(V.astype(np.float32)).to(dev)
        s = sub_t[k]
        tgt_d[p, : len(s), :3] = s[:, :3] @ Wt[:3, :3].T + Wt[:3, 3]
        tgt_d[p, : len(s), 3] = s[:, 3]
        counts_t[p] = len(s)
        c = scan_t[k]
        valid = (c[:, :3].abs().sum(dim=1) > 0).unsqueeze(1)  # no-return rays stay at the origin (the distance filter drops them)
        xyz = c[:, :3] @ Vt[:3, :3].T
        src_d[p, : len(c), :3] = torch.where(valid, xyz, torch.zeros_like(xyz))
        src_d[p, : len(c), 3] = c[:, 3]
        counts_s[p] = len(c)
        Vi = np.linalg.inv(V)
        truth.append(W @ meta[k]["S"] @ Vi)
        guesses.append((W @ meta[k]["guess"] @ Vi).astype(np.float32))
        Ws.append(W)
        Vs.append(V)
    guesses = np.stack(guesses)
    torch.cuda.synchronize()

    kw = dict(resolution=args.resolution, step_size=0.1, transformation_epsilon=0.01, max_iterations=64)  # lidar_scan_matcher.param.yaml:9-15
    b = BatchRegistration(P, P, cap, ctx=ctx, method="NDT_OMP", **kw)
    b.set_timing(True)
    pf = L.prefilter_params(min_distance=1.0, leaf_size=args.leaf, mean_k=0)

    def step():
        b.set_targets_dev(tgt_d.data_ptr(), counts_t, None)
        b.set_sources_dev(src_d.data_ptr(), counts_s, pf)
        b.align(guesses=guesses)
        return b.results(raw=True)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    b.kernel_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ktimes = b.kernel_times()
    ms_step = elapsed / args.steps * 1e3

    # ---- correctness of what was timed
    errs = np.array([synth.pose_error(res["T"][p], truth[p]) for p in range(P)])
    conv = np.asarray(res["converged"]) != 0
    n_src = np.asarray(res["n_source"]).astype(np.int64)
    n_tgt = np.asarray(res["n_target"]).astype(np.int64)
    evals = np.asarray(res["evaluations"]).astype(np.int64)  # passes over the source the pair asked for
    ok_truth = (errs[:, 0] < 0.30) & (errs[:, 1] < 2e-2)
    oracle = None
    if not args.no_oracle:
        from oracle import oracle_py as O

        worst = []
        for p in list(range(min(N_ORACLE - 1, P))) + [P - 1]:
            tg = tgt_d[p, : counts_t[p]].cpu().numpy()
            sc = O.voxel_grid(O.distance_filter(src_d[p, : counts_s[p]].cpu().numpy(), 1.0), args.leaf)[0]
            ref = O.ndt_align(sc, tg, O.ndt_params(**kw), guess=guesses[p])
            dt, dr = synth.pose_error(res["T"][p], ref["T"])
            worst.append((p, dt, dr, int(res["iterations"][p]), int(ref["iterations"])))
            if dt > 0.05 or dr > 1e-2:
                raise SystemExit(f"ndt_batch: pair {p} differs from the oracle by {dt} m / {dr} rad")
        oracle = {"pairs": [w[0] for w in worst], "max_dt_m": max(w[1] for w in worst), "max_dr_rad": max(w[2] for w in worst),
                  "inside_the_convergence_band(0.01 m, 2e-3 rad)": int(sum(w[1] <= 0.01 and w[2] <= 2e-3 for w in worst)),
                  "iterations_gpu_vs_oracle": [(w[3], w[4]) for w in worst]}
    if conv.mean() < 0.99 or ok_truth.mean() < 0.95:
        raise SystemExit(f"ndt_batch: converged {conv.mean():.3f}, within 0.3 m of the truth {ok_truth.mean():.3f}: results wrong")

    # ---- the same kernels BY THEMSELVES: two more steps with the two preparation chains one after the other on the batch's stream
    # (in the timed region they share the GPU: a bracket there also holds the time a kernel waits for the other chain's)
    b.set_side_streams(False)
    for _ in range(2):
        step()
    iso = b.kernel_times()
    b.set_side_streams(None)

    # ---- pricing
    def fam(prefix):
        for name, (ms, nl) in iso.items():
            if name.startswith(prefix):
                return name, ms / 2, nl / 2, ktimes.get(name, (0.0, 0))[0] / args.steps
        return prefix, 0.0, 0.0, 0.0

    def row(name, nl, ms, ms_timed, nbytes, unit):
        r = {"kernel": name, "launches_per_step": nl, "ms_per_step_own": round(ms, 3), "ms_per_step_timed_region_bracket": round(ms_timed, 3)}
        if nbytes:
            r.update({"algorithmic_bytes_per_step": int(nbytes), "bytes_per_unit": unit, "achieved_GBs": round(nbytes / (ms * 1e-3) / 1e9, 1) if ms else None,
                      "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms else None})
        return r

    by = []
    name, ms, nl, mt = fam("k_ndt_derivatives")
    # COMPULSORY bytes of the derivative pass: the 16-byte source point per evaluation + every leaf of a launch's targets ONCE per launch
    # (neighbouring points share their leaves and find them in L2).  SURVEY 8d's 268 B per point and evaluation (16 + 7 cells x 36) prices every leaf as if it came from HBM for
    # every point: by that figure the kernel "achieves" more than the HBM peak - kept below as `survey_pricing`, never as the fraction.
    # leaves per target: the voxels holding >= 6 points (min_points_per_voxel), counted on the host for a few targets of the batch
    leaves = []
    for p in sorted({0, P // 3, (2 * P) // 3, P - 1}):
        tg = tgt_d[p, : counts_t[p], :3].cpu().numpy()
        _, cnts = np.unique(np.floor(tg / np.float32(args.resolution)).astype(np.int64), axis=0, return_counts=True)
        leaves.append(int((cnts >= 6).sum()))
    leaves_per_target = float(np.mean(leaves))
    bytes_deriv = 16.0 * float((evals * n_src).sum()) + nl * 48.0 * leaves_per_target * P
    by.append(row(name, nl, ms, mt, bytes_deriv, "16 B per source point and evaluation + the 48-byte leaves of the launch's targets once per launch (%.0f leaves per target: "
                                                   "the voxels with >= 6 points, counted on %d of the targets); evaluations per pair: mean %.2f" % (leaves_per_target, len(leaves), float(evals.mean()))))
    survey_bytes = 268.0 * float((evals * n_src).sum())
    by[0]["survey_pricing"] = {"bytes_per_unit": "268 B per source point and evaluation (16 + 7 cells x 36: SURVEY 8d)", "algorithmic_bytes_per_step": int(survey_bytes),
                               "achieved_GBs_if_all_of_it_came_from_hbm": round(survey_bytes / (ms * 1e-3) / 1e9, 1) if ms else None,
                               "note": "above the HBM peak when it is: the 7 leaves of a point come from L2 (counter traffic of a launch = the source points alone); not a roofline fraction"}
    issue = None
    prof = os.path.join(ROOT, "profiles")
    for iname in sorted((f for f in os.listdir(prof) if f.endswith("_ndt_issue.json")), reverse=True):
        hit = json.load(open(os.path.join(prof, iname)))["kernels"].get("k_ndt_derivatives")
        if hit:
            issue = dict(hit, source=f"profiles/{iname} (SQ counters of this command: SQ_ACTIVE_INST_VALU / (8 SQ_BUSY_CYCLES), benchmarks/summarize_evidence.py ISSUE_NOTE)")
        break
    bytes_tgt = 16.0 * float(n_tgt.sum())
    name, ms, nl, mt = fam("ndt_target_sort")
    by.append(row(name, nl, ms, mt, 4.0 * bytes_tgt, "64 B per target point: 4 passes x (8 B read + 8 B written) of (voxel index, point index)"))
    name, ms, nl, mt = fam("k_ndt_leaves")
    by.append(row(name, nl, ms, mt, bytes_tgt * 24.0 / 16.0, "24 B per target point (16 B point gathered + 8 B sorted pair; leaves written: 48 B per voxel, not counted)"))
    name, ms, nl, mt = fam("ndt_target_prep")
    by.append(row(name, nl, ms, mt, bytes_tgt * 40.0 / 16.0, "40 B per target point (bounds 16 + keys 16 + 8 written)"))
    for name in sorted(iso):
        if not name.startswith(("k_ndt_derivatives", "ndt_target", "k_ndt_leaves")):
            by.append(row(name, iso[name][1] / 2, iso[name][0] / 2, ktimes.get(name, (0.0, 0))[0] / args.steps, 0, ""))
    total_bytes = survey_bytes + bytes_tgt + 32.0 * float(counts_s.sum())
    line = {
        "metric": "NDT_OMP scan-vs-sub-map pairs/sec (64k-pt scans, 5-key-frame sub-maps, 2.0 m cells) on 1 MI355X", "not_the_headline": True,
        "value": round(P / (ms_step * 1e-3), 1), "unit": "pairs/s", "ms_per_step": round(ms_step, 3), "steps": args.steps, "warmup": args.warmup,
        "n_gpus": 1, "dtype": "f32 per-point maths, f64 accumulation (ndt_omp's widths)", "data": "synthetic",
        "config": {"workload": "lidar_scan_matcher's default engine: NDT_OMP, DIRECT7, resolution %.1f m, step 0.1, eps 0.01, <= 64 iterations; "
                               "source prefilter 1.0 m + voxel %.1f m; %d pairs per step" % (args.resolution, args.leaf, P),
                   "bases": nb, "key_frames_per_submap": args.keyframes, "capacity": cap, "source_points_mean": float(n_src.mean()),
                   "target_points_mean": float(n_tgt.mean()), "device": dev_name, "compute_units": cus},
        "iterations_mean": round(float(np.asarray(res["iterations"]).mean()), 2), "converged": float(conv.mean()),
        "pose_error_vs_truth": {"median_m": round(float(np.median(errs[:, 0])), 4), "p95_m": round(float(np.percentile(errs[:, 0], 95)), 4),
                                "within_0.3m": float(ok_truth.mean())},
        "gpu_vs_oracle": oracle,
        "roofline": {"bound": "hbm", "kernel": by[0]["kernel"], "achieved": by[0].get("achieved_GBs"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": by[0].get("frac_of_hbm_peak"),
                     "valu_issue_frac": issue["valu_issue_frac"] if issue else None, "valu_issue": issue,
                     "note": "the derivative pass is bound by VALU issue, not by bytes: frac prices its COMPULSORY bytes (by_kernel[0].bytes_per_unit)",
                     "end_to_end_frac_by_survey_pricing": round(total_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "by_kernel": by, "seconds_ray_casting_on_host": round(t_cast, 1),
    }
    b.close()

    if args.config3:
        c3 = raw[len(jobs):]
        sub = np.concatenate([place(c[np.abs(c[:, :3]).sum(1) > 0], Pk) for c, Pk in zip(c3[:-1], c3_poses)], axis=0)
        scan = c3[-1][np.abs(c3[-1][:, :3]).sum(1) > 0]
        guess = synth.se3(7.2, 0.1, 0, 0, 0, 0.01).astype(np.float32)
        ndt = NormalDistributionsTransform(resolution=1.0, ctx=ctx)

        def timed(f, reps=7, warm=2):
            for _ in range(warm):
                f()
            ts = []
            for _ in range(reps):
                ctx.synchronize()
                t0 = time.perf_counter()
                f()
                ctx.synchronize()
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts)) * 1e3, float(np.min(ts)) * 1e3

        dsub = None
        try:
            from lidar_graph_slam_amd.registration import DeviceCloud

            dsub = DeviceCloud.upload(sub, ctx)
        except Exception:
            dsub = None
        t_host = timed(lambda: ndt.setInputTarget(sub))
        t_dev = timed(lambda: ndt.setInputTarget(dsub)) if dsub is not None else None
        ndt.setInputSource(scan)
        t_align = timed(lambda: ndt.align(guess))
        line["config3"] = {"target_points": int(len(sub)), "source_points": int(len(scan)), "cells_m": 1.0,
                           "set_target_from_host_ms(median,min)": t_host, "set_target_device_cloud_ms(median,min)": t_dev,
                           "align_ms(median,min)": t_align, "iterations": int(ndt.result["iterations"]),
                           "evaluations": int(ndt.result["evaluations"]),
                           "pose_err_vs_truth": synth.pose_error(ndt.getFinalTransformation(), c3_true)}
    print(json.dumps(line, default=lambda o: list(o) if isinstance(o, tuple) else float(o)))


if __name__ == "__main__":
    main()
