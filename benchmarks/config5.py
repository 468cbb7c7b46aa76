#!/usr/bin/env python3
"""BASELINE config 5 as a runnable multi-GPU program: a global list of loop-closure candidates against sub-maps, sharded over
the ranks of a torch.distributed job (one process per GPU), registered with one BatchRegistration sequence per rank, results
all-gathered over RCCL, every pair verified against its ground truth on rank 0.

    python benchmarks/config5.py --candidates 512 --submaps 8                         (one GPU: the per-GPU share)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        benchmarks/config5.py --gpus 8 --candidates 4096 --submaps 64                 (the full config)

Synthetic data (no network, the reference ships none): sub-map s = 5 key-frame scans around x = 20 s on the street scene,
transformed to the map frame; a candidate = a scan cast at the true pose P, placed at a wrong graph pose G (what the pose graph
believes before the loop is closed); aligning it to the sub-map must return P G^-1.  Every rank casts only what it owns."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def multi_lanes(args):
    """The whole candidate list through mi355_multi_batch_register from ONE process: the C++ back end's form."""
    import multiprocessing as mp

    from lidar_graph_slam_amd import synth

    n, S, casts = args.candidates, args.submaps, args.casts_per_submap
    length = 20.0 * S + 20.0
    sub_pose = lambda s: synth.se3(20.0 * s, 0.0, 0.0)  # noqa: E731
    true_off = [synth.random_perturbation(np.random.default_rng(1000 + s * 64 + c), 1.0, 2.0) for s in range(S) for c in range(casts)]
    rng = np.random.default_rng(5)
    wrong = [synth.random_perturbation(rng, 1.5, 4.0) for _ in range(n)]
    tgt_of_src = np.arange(n) % S
    cast_of = (np.arange(n) // S) % casts
    jobs = [(length, sub_pose(s) @ true_off[s * casts + c], 9000 + s * 64 + c, args.n_azimuth, 30.0) for s in range(S) for c in range(casts)]
    jobs += [(length, sub_pose(s) @ synth.se3(1.0 * j, 0.0, 0.0), 500 + 10 * s + j, args.n_azimuth, 30.0) for s in range(S) for j in range(-2, 3)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(max(1, min(64, (os.cpu_count() or 8) - 2))) as pool:
        raw = pool.map(synth.cast_job, jobs, chunksize=4)
    t_cast = time.perf_counter() - t0

    import torch  # noqa: F401  (torch's ROCm runtime before the library's: INTEGRATION.md)

    if torch.cuda.is_available():
        torch.cuda.init()
    from lidar_graph_slam_amd import loop_batch
    from lidar_graph_slam_amd.filters import PointsPreFiltering
    from lidar_graph_slam_amd.multi import MultiDevice

    def place(c, T):
        o = c.copy()
        o[:, :3] = (c[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        return o

    pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
    scans = [pre.callback_sensor_points(r) for r in raw]
    submaps = [np.concatenate([place(scans[S * casts + 5 * s + (j + 2)], sub_pose(s) @ synth.se3(1.0 * j, 0.0, 0.0)) for j in range(-2, 3)], axis=0)
               for s in range(S)]
    true_pose = lambda p: sub_pose(int(tgt_of_src[p])) @ true_off[int(tgt_of_src[p]) * casts + int(cast_of[p])]  # noqa: E731
    graph_pose = lambda p: true_pose(p) @ wrong[p]  # noqa: E731
    sources = [place(scans[int(tgt_of_src[p]) * casts + int(cast_of[p])], graph_pose(p)) for p in range(n)]
    md = MultiDevice([0] * args.multi_lanes)
    pf = loop_batch.voxel_prefilter(0.5)
    times = []
    for _ in range(4):  # the first call allocates every lane's batch object
        t0 = time.perf_counter()
        res, fit, lane = md.batch_register(sources, submaps, tgt_of_src, target_prefilter=pf, with_fitness=True, raw=True, max_iterations=args.max_iterations)
        times.append(time.perf_counter() - t0)
    md.close()
    errs = np.array([synth.pose_error(res["T"][p], true_pose(p) @ np.linalg.inv(graph_pose(p))) for p in range(n)])
    accepted = (np.asarray(res["converged"]) != 0) & (fit < 0.3)  # graph_based_slam.cpp:321-328
    good = (errs[:, 0] < 0.05) & (errs[:, 1] < 5e-3)
    t_best = min(times[1:])
    line = {
        "config": "BASELINE configs[4] on ONE MI355X: loop-closure batch through mi355_multi_batch_register (host views in, results out; one process, lanes on device 0)",
        "lanes": args.multi_lanes, "candidates": n, "submaps": S, "method": "FAST_GICP",
        "pairs_per_s_including_uploads": round(n / t_best, 1), "seconds_per_call": [round(t, 4) for t in times],
        "bytes_uploaded_per_call": int(sum(c.nbytes for c in sources) + sum(c.nbytes for c in submaps)),
        "points_per_candidate_mean": float(np.mean([len(c) for c in sources])), "points_per_submap_mean": float(np.mean([len(c) for c in submaps])),
        "accepted(fitness<0.3)": int(accepted.sum()), "within_5cm_of_truth": int(good.sum()), "median_pose_err_m": round(float(np.median(errs[:, 0])), 4),
        "mean_iterations": round(float(np.asarray(res["iterations"]).mean()), 2), "seconds_ray_casting_on_host": round(t_cast, 1),
    }
    print(json.dumps(line))
    if good.sum() < args.min_good * n:
        raise SystemExit("config5: results wrong")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--candidates", type=int, default=512, help="global number of candidates (4096 in BASELINE config 5)")
    ap.add_argument("--submaps", type=int, default=8, help="global number of sub-maps (64 in BASELINE config 5)")
    ap.add_argument("--casts-per-submap", type=int, default=8, help="distinct candidate scans per sub-map; the rest are other wrong placements of those")
    ap.add_argument("--n-azimuth", type=int, default=1024, help="1024 = 65 536 rays per scan")
    ap.add_argument("--capacity", type=int, default=65536)
    ap.add_argument("--max-iterations", type=int, default=64)
    ap.add_argument("--method", default="FAST_GICP", choices=["FAST_GICP", "NDT_OMP", "ICP", "GICP"], help="registration engine of the batch (graph_based_slam.cpp:80-152; ICP is that node's yaml default)")
    ap.add_argument("--repeats", type=int, default=2, help="uploads + registration + read-back this many times on the same batch object, the shortest reported "
                    "(the first call of a process pays the library's workspace allocations)")
    ap.add_argument("--min-good", type=float, default=0.9, help="fail unless this fraction of the pairs lands within 5 cm / 5 mrad of the truth")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip timing the oracle on four of the candidates")
    ap.add_argument("--multi-lanes", type=int, default=0, help="ONE process, this many lanes on device 0 through mi355_multi_batch_register (host views in, "
                    "results out: uploads included in the rate) instead of one rank per GPU")
    args = ap.parse_args()
    if args.multi_lanes > 0:
        return multi_lanes(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N benchmarks/config5.py --gpus N ...")
    dev_index = int(os.environ.get("BENCH_DEVICE", local_rank))  # BENCH_DEVICE / BENCH_BACKEND: several ranks on ONE GPU (tests)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend == "nccl" or world == 1:
        if not torch.cuda.is_available():
            raise SystemExit("config5.py needs an MI355X: there is no CPU fallback for the product path")
        torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from lidar_graph_slam_amd import loop_batch, synth
    from lidar_graph_slam_amd.context import Context
    from lidar_graph_slam_amd.filters import PointsPreFiltering

    ctx = Context(dev_index)
    pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0, ctx=ctx)
    scene = synth.street(20.0 * args.submaps + 20.0)

    # ---- the GLOBAL list: metadata only (every rank builds the same one) ------------------------------------------
    n, S = args.candidates, args.submaps
    rng = np.random.default_rng(5)
    tgt_of_src = np.arange(n) % S
    cast_of = (np.arange(n) // S) % args.casts_per_submap          # which of the sub-map's candidate scans
    true_off = [synth.random_perturbation(np.random.default_rng(1000 + s * 64 + c), 1.0, 2.0) for s in range(S) for c in range(args.casts_per_submap)]
    wrong = [synth.random_perturbation(rng, 1.5, 4.0) for _ in range(n)]  # the graph's error of candidate p (<= 2 m / 5 deg in all)

    def submap_pose(s):
        return synth.se3(20.0 * s, 0.0, 0.0)

    def true_pose(p):
        s = int(tgt_of_src[p])
        return submap_pose(s) @ true_off[s * args.casts_per_submap + int(cast_of[p])]

    def graph_pose(p):  # where the (drifted) pose graph believes the candidate is
        return true_pose(p) @ wrong[p]

    cache = {}

    def scan_at(s, c):  # prefiltered scan (what points_prefiltering publishes), sensor frame
        if (s, c) not in cache:
            P = submap_pose(s) @ true_off[s * args.casts_per_submap + c]
            cache[(s, c)] = pre.callback_sensor_points(synth.cast_scan(scene, P, n_azimuth=args.n_azimuth, seed=9000 + s * 64 + c, ground_max_range=30.0))
        return cache[(s, c)]

    def place(cloud, T):
        o = cloud.copy()
        o[:, :3] = (cloud[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        return o

    def load_source(p):  # the candidate key frame at its graph pose (graph_based_slam.cpp:297-309 transforms with the graph's poses)
        return place(scan_at(int(tgt_of_src[p]), int(cast_of[p])), graph_pose(p))

    def load_target(s):  # +-2 key frames around the sub-map pose, in the map frame (voxel 0.5 m happens on the device)
        parts = []
        for j in range(-2, 3):
            P = submap_pose(s) @ synth.se3(1.0 * j, 0.0, 0.0)
            c = pre.callback_sensor_points(synth.cast_scan(scene, P, n_azimuth=args.n_azimuth, seed=500 + 10 * s + j, ground_max_range=30.0))
            parts.append(place(c, P))
        return np.concatenate(parts, axis=0)

    cap = max(args.capacity, 5 * 22000 if args.n_azimuth >= 1024 else 0)
    t0 = time.perf_counter()
    out = loop_batch.register_sharded(tgt_of_src, load_source, load_target, cap, rank=rank, world=world, ctx=ctx,
                                      target_prefilter=loop_batch.voxel_prefilter(0.5), method=args.method, repeats=args.repeats,
                                      **(dict(max_iterations=args.max_iterations) if args.method == "FAST_GICP"
                                         else dict(max_iterations=100, resolution=1.0, step_size=0.1, transformation_epsilon=0.01) if args.method == "NDT_OMP"
                                         else dict(max_iterations=100, max_correspondence_distance=2.0, max_optimizer_iterations=20, transformation_epsilon=0.01,
                                                   euclidean_fitness_epsilon=1e-8) if args.method == "GICP"  # graph_based_slam.param.yaml:24-29
                                         else dict(max_iterations=100, max_correspondence_distance=30.0, transformation_epsilon=1e-8, euclidean_fitness_epsilon=1e-6)))
    t_total = time.perf_counter() - t0
    if rank == 0:
        from lidar_graph_slam_amd import synth as S_
        errs = np.array([S_.pose_error(out["T"][p], true_pose(p) @ np.linalg.inv(graph_pose(p))) for p in range(n)])
        accepted = out["converged"] & (out["fitness"] < 0.3)  # graph_based_slam.cpp:321-328
        # NDT with 1 m cells lands within decimetres (the oracle does the same, tests/test_gpu_fullsize.py): a wider gate for it
        good = (errs[:, 0] < 0.05) & (errs[:, 1] < 5e-3) if args.method in ("FAST_GICP", "GICP") else (errs[:, 0] < 0.30) & (errs[:, 1] < 2e-2)
        cpu = None
        if not args.no_cpu_baseline:
            # the CPU path beside it: the oracle - the restated reference engine - on a handful of the SAME candidates, one
            # after the other with the reference's 4 OpenMP threads (graph_based_slam registers one candidate per tick): per candidate
            # setInputTarget (kd-tree / covariances / leaves of the voxel-filtered sub-map) + setInputSource + align + getFitnessScore
            from oracle import oracle_py as O

            sample = sorted({(k * n) // 4 for k in range(4)})
            threads = min(4, O.max_threads())
            tc = 0.0
            worst = 0.0
            for p_ in sample:
                src_c = load_source(p_)
                tgt_c = O.voxel_grid(load_target(int(tgt_of_src[p_])), 0.5)[0]
                tc0 = time.perf_counter()
                if args.method == "FAST_GICP":
                    r_ = O.gicp_align(src_c, tgt_c, O.gicp_params(max_iterations=args.max_iterations, num_threads=threads))
                elif args.method == "NDT_OMP":
                    r_ = O.ndt_align(src_c, tgt_c, O.ndt_params(max_iterations=100, resolution=1.0, step_size=0.1, transformation_epsilon=0.01, num_threads=threads))
                elif args.method == "ICP":
                    r_ = O.icp_align(src_c, tgt_c, O.icp_params(max_iterations=100, max_correspondence_distance=30.0, transformation_epsilon=1e-8, euclidean_fitness_epsilon=1e-6, num_threads=threads))
                else:
                    r_ = O.gicp_bfgs_align(src_c, tgt_c, O.gicp_bfgs_params(max_iterations=100, max_correspondence_distance=2.0, max_optimizer_iterations=20,
                                                                             transformation_epsilon=0.01, num_threads=threads))
                O.fitness(src_c, tgt_c, r_["T"])
                tc += time.perf_counter() - tc0
                worst = max(worst, float(S_.pose_error(out["T"][p_], r_["T"])[0]))
            cpu = {"value": round(len(sample) / tc, 3), "unit": "pairs/s", "cores": threads, "kind": "port",
                   "sample": f"candidates {sample} of this list through oracle/liboracle.so ({args.method}: target set-up + align + fitness per candidate, {threads} OpenMP threads - the reference's "
                             f"configuration) in {tc:.1f} s", "gpu_vs_cpu_pose_dt_m_max": round(worst, 9)}
        line = {
            "config": "BASELINE configs[4]: loop-closure batch, candidates sharded by sub-map over the ranks, results all-gathered",
            "n_gpus": world, "backend": backend if world > 1 else "single process", "collective_ranks": dist.get_world_size() if world > 1 else 1,
            "method": args.method, "candidates": n, "submaps": S, "all_reported": bool(out["reported"].all()),
            "pairs_per_s_registration": round(n / max(out["seconds_register"], 1e-9), 1),
            "seconds_register_max_over_ranks": round(out["seconds_register"], 3), "repeats(shortest reported)": args.repeats, "seconds_gather": round(out["seconds_gather"], 4),
            "seconds_total_incl_synthesis": round(t_total, 1),
            "accepted(fitness<0.3)": int(accepted.sum()), "within_5cm_of_truth" if args.method in ("FAST_GICP", "GICP") else "within_30cm_of_truth": int(good.sum()),
            "median_pose_err_m": round(float(np.median(errs[:, 0])), 4), "max_pose_err_m_of_accepted": round(float(errs[accepted, 0].max()), 4) if accepted.any() else None,
            "mean_iterations": round(float(out["iterations"].mean()), 1),
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
        if not out["reported"].all() or good.sum() < args.min_good * n:
            raise SystemExit("config5: results missing or wrong")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
