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


def gather_results(local_idx, local_T, local_fitness, n_total: int, group=None):
    """All-gather per-pair results: returns (T [n_total,4,4] float32, fitness [n_total] float64) on every rank.
    local_idx: global pair indices this rank processed."""
    import torch
    import torch.distributed as dist

    local_idx = np.asarray(local_idx, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        T = np.zeros((n_total, 4, 4), dtype=np.float32)
        f = np.full(n_total, np.nan)
        T[local_idx] = local_T
        f[local_idx] = local_fitness
        return T, f
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    # ragged shards: pad to the largest shard (the payload is tiny: 18 doubles per pair)
    n_local = torch.tensor([len(local_idx)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    n_max = int(max(int(s.item()) for s in sizes))
    payload = torch.zeros((n_max, 18), dtype=torch.float64, device=dev)
    if len(local_idx):
        payload[: len(local_idx), 0] = torch.from_numpy(local_idx.astype(np.float64)).to(dev)
        payload[: len(local_idx), 1:17] = torch.from_numpy(np.asarray(local_T, dtype=np.float64).reshape(-1, 16)).to(dev)
        payload[: len(local_idx), 17] = torch.from_numpy(np.asarray(local_fitness, dtype=np.float64)).to(dev)
    parts = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(parts, payload, group=group)
    T = np.zeros((n_total, 4, 4), dtype=np.float32)
    f = np.full(n_total, np.nan)
    for r in range(world):
        k = int(sizes[r].item())
        if k == 0:
            continue
        a = parts[r][:k].cpu().numpy()
        idx = a[:, 0].astype(np.int64)
        T[idx] = a[:, 1:17].reshape(-1, 4, 4).astype(np.float32)
        f[idx] = a[:, 17]
    return T, f


def max_over_ranks(value: float, group=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
