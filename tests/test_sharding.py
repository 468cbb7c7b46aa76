"""The N > 1 path on CPU: two gloo ranks shard a loop-closure batch (pairs grouped by sub-map), "register" their shards
(a deterministic stand-in result per pair — no GPU here) and all-gather the fixed-size results."""
import os

import numpy as np
import pytest

from lidar_graph_slam_amd import sharding


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 4096):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_shard_by_target_keeps_submaps_together():
    rng = np.random.default_rng(0)
    tgt = rng.integers(0, 64, 4096)
    for world in (1, 2, 4, 8):
        shards = sharding.shard_by_target(tgt, world)
        allidx = np.concatenate(shards)
        assert sorted(allidx.tolist()) == list(range(4096))
        owners = {}
        for r, idx in enumerate(shards):
            for t in np.unique(tgt[idx]):
                assert owners.setdefault(int(t), r) == r  # one rank per sub-map
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 0.15 * 4096 / world + 64


def test_the_c_abi_partition_is_the_python_partition():
    """mi355_multi_shard_by_target (one process, one lane per device: csrc/host/lanes.hpp) and sharding.shard_by_target (one
    process per GPU) deal the same pairs to the same lane / rank - ties, empty lanes and skewed sub-maps included."""
    from lidar_graph_slam_amd import multi

    rng = np.random.default_rng(7)
    cases = [rng.integers(0, 64, 4096), rng.integers(0, 3, 50), np.zeros(9, np.int64), np.arange(17) % 5, np.repeat([4, 1, 9], [7, 7, 7]),
             (rng.zipf(1.5, 600) % 40)]
    for tgt in cases:
        for world in (1, 2, 3, 8):
            lane = multi.shard_by_target(tgt, world)
            ref = np.full(len(tgt), -1)
            for r, idx in enumerate(sharding.shard_by_target(tgt, world)):
                ref[idx] = r
            np.testing.assert_array_equal(lane, ref)


def test_multi_create_fails_without_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lidar_graph_slam_amd import _lib as L
    from lidar_graph_slam_amd import multi

    with pytest.raises(L.Mi355Error):
        multi.MultiDevice([0])


def _worker(rank, world, port, n_pairs, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(1)
        tgt = rng.integers(0, 5, n_pairs)
        mine = sharding.shard_by_target(tgt, world)[rank]
        T = np.zeros((len(mine), 4, 4), np.float32)
        for k, p in enumerate(mine):  # stand-in for the GPU registration of pair p
            T[k] = np.eye(4)
            T[k, :3, 3] = (p, tgt[p], rank)
        fit = mine.astype(np.float64) * 0.5
        Tall, fall = sharding.gather_results(mine, T, fit, n_pairs)
        tmax = sharding.max_over_ranks(1.0 + rank)
        q.put((rank, Tall, fall, tmax, len(mine)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather():
    import torch.multiprocessing as mp

    world, n_pairs = 2, 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[4] for r in results) == n_pairs
    for rank, Tall, fall, tmax, _ in results:
        assert tmax == 2.0
        np.testing.assert_array_equal(Tall[:, 0, 3], np.arange(n_pairs))  # every pair present, in global order
        np.testing.assert_allclose(fall, np.arange(n_pairs) * 0.5)
    np.testing.assert_array_equal(results[0][1], results[1][1])  # all ranks hold the same gathered result


def _worker_rows(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tgt = np.zeros(5, dtype=np.int64)  # ONE sub-map: the second rank owns nothing but still takes part in the gather
        mine = sharding.shard_by_target(tgt, world)[rank]
        rows = np.stack([mine * 10.0 + c for c in range(22)], axis=1) if len(mine) else np.zeros((0, 22))
        q.put((rank, len(mine), sharding.gather_rows(mine, rows, 5)))
    finally:
        dist.destroy_process_group()


def test_gather_rows_with_an_idle_rank():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 977) % 2000)
    procs = [ctx.Process(target=_worker_rows, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[1] for r in results) == [0, 5]
    for _, _, g in results:
        assert g.shape == (5, 22)
        np.testing.assert_array_equal(g[:, 0], np.arange(5) * 10.0)
        np.testing.assert_array_equal(g[:, 21], np.arange(5) * 10.0 + 21)
