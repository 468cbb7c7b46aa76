"""Many-scans-vs-sub-map batch mode across GPUs (BASELINE config 5): a GLOBAL list of loop-closure candidates,
each to be registered against one of a set of sub-maps (graph_based_slam.cpp:297-322, one candidate per 1 Hz tick there), is
sharded over the ranks of a torch.distributed job so that all candidates of a sub-map land on one rank; every rank registers its
shard with ONE BatchRegistration sequence on its GPU; the fixed-size results (pose, fitness, flags) are all-gathered — the
only collective of the path (RCCL over xGMI on GPUs, gloo in the CPU tests).

The caller supplies the clouds through two loaders so that a rank materialises only what it owns.
"""
from __future__ import annotations

import time

import numpy as np

from . import _lib as L
from . import sharding

RESULT_COLUMNS = ("T[16]", "fitness", "converged", "iterations", "status", "n_source", "n_target")


def register_sharded(tgt_of_src, load_source, load_target, capacity: int, *, rank: int = 0, world: int = 1, ctx=None, group=None,
                     source_prefilter=None, target_prefilter=None, method: str = "FAST_GICP", repeats: int = 1, **params):
    """Register pair p = (source p, sub-map tgt_of_src[p]) for every p, sharded by sub-map over `world` ranks.

    load_source(p) / load_target(t) -> float32 [n, 4] packed clouds (n <= capacity), called only for what this rank owns.
    method: "FAST_GICP", "NDT_OMP", "ICP" or "GICP" (graph_based_slam.cpp:80-152), params: that engine's (mi355_gicp_params / mi355_ndt_params /
    mi355_icp_params / mi355_gicp_bfgs_params).
    Returns a dict, identical on every rank: T [n,4,4] float32, fitness, converged, iterations, status, n_source, n_target
    (arrays over ALL pairs), plus this rank's timings: shard sizes, seconds of registration (max over ranks) and of the gather.
    repeats > 1 (measurement only): the uploads + registration + read-back run that many times on the same batch object and the
    SHORTEST is reported (the first call of a process also pays the library's grow-only workspace allocations); the results are
    those of the last run - the same every time."""
    from .registration import BatchRegistration

    tgt_of_src = np.asarray(tgt_of_src, dtype=np.int64)
    n_pairs = len(tgt_of_src)
    mine = sharding.shard_by_target(tgt_of_src, world)[rank]
    rows = np.zeros((len(mine), 22))
    t_reg = 0.0
    if len(mine):
        targets = np.unique(tgt_of_src[mine])
        local_of = {int(t): k for k, t in enumerate(targets)}
        tg = np.zeros((len(targets), capacity, 4), np.float32)
        nt = np.zeros(len(targets), np.int32)
        for k, t in enumerate(targets):
            c = np.asarray(load_target(int(t)), dtype=np.float32)
            if len(c) > capacity:
                raise ValueError(f"sub-map {t}: {len(c)} points exceed the batch capacity {capacity}")
            tg[k, : len(c)] = c
            nt[k] = len(c)
        sr = np.zeros((len(mine), capacity, 4), np.float32)
        ns = np.zeros(len(mine), np.int32)
        for k, p in enumerate(mine):
            c = np.asarray(load_source(int(p)), dtype=np.float32)
            if len(c) > capacity:
                raise ValueError(f"candidate {p}: {len(c)} points exceed the batch capacity {capacity}")
            sr[k, : len(c)] = c
            ns[k] = len(c)
        b = BatchRegistration(len(mine), len(targets), capacity, ctx=ctx, method=method, **params)
        t_reg = float("inf")
        for _ in range(max(1, int(repeats))):
            t0 = time.perf_counter()
            b.set_targets(tg, nt, target_prefilter)
            b.set_sources(sr, ns, source_prefilter)
            b.align([local_of[int(t)] for t in tgt_of_src[mine]], with_fitness=True)
            res, fit = b.results(with_fitness=True)
            t_reg = min(t_reg, time.perf_counter() - t0)
        b.close()
        for k, r in enumerate(res):
            rows[k, :16] = np.asarray(r["T"], dtype=np.float64).reshape(16)
            rows[k, 16:] = (fit[k], r["converged"], r["iterations"], r["status"], r["n_source"], r["n_target"])
    t0 = time.perf_counter()
    g = sharding.gather_rows(mine, rows, n_pairs, group)
    t_gather = time.perf_counter() - t0
    return {
        "T": np.nan_to_num(g[:, :16], nan=0.0).reshape(n_pairs, 4, 4).astype(np.float32),
        "fitness": g[:, 16], "converged": g[:, 17] > 0.5, "iterations": np.nan_to_num(g[:, 18]).astype(np.int64),
        "status": np.nan_to_num(g[:, 19], nan=-1).astype(np.int64), "n_source": np.nan_to_num(g[:, 20]).astype(np.int64),
        "n_target": np.nan_to_num(g[:, 21]).astype(np.int64),
        "reported": ~np.isnan(g[:, 16]),
        "shard_pairs": len(mine), "seconds_register": sharding.max_over_ranks(t_reg, group), "seconds_gather": t_gather,
    }


def voxel_prefilter(leaf: float):
    """graph_based_slam.cpp:61,311-313: the sub-map is voxel-filtered (0.5 m) before it becomes the registration target."""
    return L.prefilter_params(min_distance=0.0, leaf_size=float(leaf), mean_k=0)
