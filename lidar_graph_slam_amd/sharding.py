"""Multi-GPU sharding of independent scan pairs (BASELINE config 5).

Scan pairs are independent, so the path shards with NO data-path collective: every rank (one process per GPU)
registers its own pairs against its own copies of the sub-maps it needs.  The only exchange is the final gather of the
fixed-size results (4x4 pose, fitness, flags: 80 B per pair) — `all_gather` over RCCL (xGMI) on GPUs, gloo in the CPU
tests.  Pairs that share a sub-map are kept on one rank so each sub-map is uploaded and indexed once.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous balanced [lo, hi) of `n_items` for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_target(tgt_of_src, world: int):
    """Assign pairs to ranks so that all pairs of one target (sub-map) land on the same rank, greedily balancing the pair
    count.  Returns a list (per rank) of pair-index arrays, each sorted by target then pair index (deterministic)."""
    tgt_of_src = np.asarray(tgt_of_src, dtype=np.int64)
    targets, counts = np.unique(tgt_of_src, return_counts=True)
    order = np.lexsort((targets, -counts))  # biggest groups first, ties by target id
    load = np.zeros(world, dtype=np.int64)
    owner = {}
    for k in order:
        r = int(np.argmin(load))  # lowest rank wins ties
        owner[int(targets[k])] = r
        load[r] += counts[k]
    out = []
    for r in range(world):
        idx = np.flatnonzero(np.fromiter((owner[int(t)] == r for t in tgt_of_src), dtype=bool, count=len(tgt_of_src)))
        idx = idx[np.lexsort((idx, tgt_of_src[idx]))]
        out.append(idx)
    return out


def gather_rows(local_idx, rows, n_total: int, group=None, force_collective: bool = False) -> np.ndarray:
    """All-gather fixed-size per-pair records: `rows` is [n_local, C] float64 for the global pair indices `local_idx`; returns
    [n_total, C] on every rank (NaN rows for pairs nobody reported).  Ragged shards are padded to the largest one — the payload
    is tiny (<= 20 doubles per pair), so ONE all_gather over RCCL / xGMI is the whole exchange step of the path.
    force_collective: run the all_gather even in a one-rank group (a probe that the backend executes it)."""
    import torch
    import torch.distributed as dist

    local_idx = np.asarray(local_idx, dtype=np.int64)
    rows = np.asarray(rows, dtype=np.float64)
    if rows.ndim != 2:
        rows = rows.reshape(len(local_idx), -1) if len(local_idx) else np.zeros((0, 0))
    C = rows.shape[1]
    out = np.full((n_total, C), np.nan)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collective):
        out[local_idx] = rows
        return out
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    n_local = torch.tensor([len(local_idx), C], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    n_max = int(max(int(s[0].item()) for s in sizes))
    C_all = int(max(int(s[1].item()) for s in sizes))  # a rank without pairs does not know C
    payload = torch.zeros((n_max, C_all + 1), dtype=torch.float64, device=dev)
    if len(local_idx):
        payload[: len(local_idx), 0] = torch.from_numpy(local_idx.astype(np.float64)).to(dev)
        payload[: len(local_idx), 1 : C + 1] = torch.from_numpy(rows).to(dev)
    parts = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(parts, payload, group=group)
    out = np.full((n_total, C_all), np.nan)
    for r in range(world):
        k = int(sizes[r][0].item())
        if k == 0:
            continue
        a = parts[r][:k].cpu().numpy()
        out[a[:, 0].astype(np.int64)] = a[:, 1:]
    return out


def gather_results(local_idx, local_T, local_fitness, n_total: int, group=None):
    """All-gather per-pair results: returns (T [n_total,4,4] float32, fitness [n_total] float64) on every rank.
    local_idx: global pair indices this rank processed."""
    local_idx = np.asarray(local_idx, dtype=np.int64)
    rows = np.concatenate([np.asarray(local_T, dtype=np.float64).reshape(len(local_idx), 16),
                           np.asarray(local_fitness, dtype=np.float64).reshape(len(local_idx), 1)], axis=1)
    g = gather_rows(local_idx, rows, n_total, group)
    T = np.nan_to_num(g[:, :16], nan=0.0).reshape(n_total, 4, 4).astype(np.float32)
    return T, g[:, 16]


def max_over_ranks(value: float, group=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
