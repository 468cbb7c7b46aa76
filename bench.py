#!/usr/bin/env python3
"""Benchmark of the hot path: scan-pairs/sec (64k-pt pairs, voxel 0.2 m + 20 GICP iterations) on N MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

One *step* = one pass of the whole hot path over one batch of `--pairs` independent synthetic scan pairs per GPU,
inputs already resident in HBM: distance filter + VoxelGrid 0.2 m on both clouds -> voxel-hash build -> k=20
covariances -> exactly 20 Levenberg-Marquardt GICP iterations -> poses copied back.  Pairs are sharded across ranks with
no data-path collective (weak scaling); the only collective is the MAX of the elapsed time.  BASELINE.json configs[1].
Default: ONE batch of 4 096 pairs at a time (~113 GB of the 288 GB of HBM in use, as `config.hbm_bytes_in_use` of the line measures it:
27.5 MB per pair with the raw clouds, the batch object and its workspaces) —
every kernel duration the roofline block reports is then what rocprofv3 reports for the same command.  `--inflight 2 --pairs 512` keeps two batch objects in flight on two HIP streams in the same memory
(while one batch is in its latency- and bandwidth-bound LM iterations the next one is prefiltered and indexed by VALU-bound
kernels): +8 % pairs/s, `--inflight 3`: +13 %; K steps are then still K whole batches, all finished inside the timed region, but
HIP-event brackets around kernels of interleaved streams also contain the time a kernel waits for the other stream's.

The JSON line also carries
  roofline      the kernel family with the LARGEST share of GPU time in the timed region (HIP events recorded by the library
                on the launching streams around every hot kernel): algorithmic bytes per launch / mean launch
                duration against the 8 TB/s HBM peak; `by_kernel` lists every family the same way and `end_to_end` prices the
                whole step (algorithmic bytes of prefilter + covariances + 20 iterations over ms_per_step).  Durations in the
                timed region are those of kernels that SHARE the GPU (batches in flight, two preparation streams per batch);
                `isolated_*` are the same kernels with nothing overlapping them, from two extra steps after the timed region;
  cpu_baseline  the oracle (CPU restatement of the reference's fast_gicp path, OpenMP) timed on this host's cores over a
                bounded sample of the same workload, at the reference's 4 threads and at 16.
The run FAILS (non-zero exit, no JSON line) when the poses it timed are wrong: error vs ground truth or vs the oracle on
the same input above the stated tolerances.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# ALGORITHMIC bytes per unit of work: what a perfect implementation must move, per point.
ALG_BYTES = {
    "k_gicp_linearize": ("source point", 80),    # src pt 16 + src cov 24 + tgt pt 16 + tgt cov 24
    "k_gicp_correspond": ("source point", 32),   # src pt 16 + tgt pt 16 (the 1-NN pass of update_correspondences / fitness row)
    "k_gicp_correspond_tile": ("source point", 32),
    "k_knn_cov_tile": ("cloud point", 40),       # read 16 + write 24 (neighbour gathers are implementation traffic)
    "k_voxel_centroids": ("raw point", 32),      # read 16 + write <= 16
    "sort_pairs(k_sort_*)": ("sorted element", 64),  # 4 passes x (8 B read + 8 B written) of (key, index) pairs
    "hash_build(k_hash_build|k_gather_insert)": ("cloud point", 36),  # read pt 16 + write sorted pt 16 + index 4
    # --sor only: StatisticalOutlierRemoval, SURVEY 8d: read 16 + mask 1 + compacted 16 per point (priced on the points that SURVIVE
    # it — the count the batch reports — so a lower bound of the bytes)
    "sor(sort,hash,k_knn_cov_tile<SOR>,k_knn_hard<SOR>,threshold,compact)": ("surviving point", 33),
}
TOL_TRUTH_T, TOL_TRUTH_R = 0.015, 1.0e-3   # vs ground truth on the 2 cm-noise scene
TOL_ORACLE_T, TOL_ORACLE_R = 1.0e-3, 1.0e-4  # vs the oracle on the SAME input, same iteration count
N_BASE = 32      # distinct ray-cast pairs per rank: own street scene, own sensor pose, own relative motion (synth.bench_pair_spec);
                 # pair p of the batch is pair p % N_BASE under its own world transform (which re-bins every voxel grid)
N_ORACLE = 16    # pairs of the timed batch that the oracle registers again, from the same world-transformed clouds (one of them the pair
                 # with the largest error against the ground truth)


def shared_bases_path(n_base: int) -> str:
    """The /dev/shm file local rank 0 hands the base pairs to the node's other ranks through.  Keyed by THIS launch - the rendezvous port,
    torchrun's run id and the pid of the launcher every local rank is a child of - so that a file an earlier, crashed run left behind on
    the same port is never picked up."""
    return os.path.join("/dev/shm", f"mi355_bench_bases_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}_{n_base}.npz")


def base_pairs(rank: int, n_base: int):
    """The distinct ray-cast pairs of this rank.  Ray casting is single-threaded NumPy (~1.5 s per pair): a pool of SPAWNED
    workers casts them side by side (fresh interpreters: nothing of this process's GPU state travels).  With several ranks on a node
    the pairs are cast ONCE, by local rank 0 with every host core, and handed to the other ranks through a file in /dev/shm (8 ranks
    casting 32 pairs each would queue 512 casts on the host before any GPU does anything); every rank still places them under its
    own world transforms (main), so no two pairs of the job are the same input."""
    import multiprocessing as mp

    from lidar_graph_slam_amd import synth

    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    shared = shared_bases_path(n_base) if local_world > 1 else None
    if shared and local_rank != 0:
        t0 = time.time()
        while not os.path.exists(shared):
            if time.time() - t0 > 900:
                raise SystemExit(f"bench.py: local rank 0 did not deliver {shared}")
            time.sleep(0.2)
        z = np.load(shared)
        return [(z[f"s{i}"], z[f"t{i}"], z[f"g{i}"]) for i in range(n_base)]
    # (shared: the node's ranks hold the same scenes - seeded by the NODE's rank, so that several nodes still cast different ones - and
    # differ by their world transforms; alone: seeded by the rank)
    specs = [synth.bench_pair_spec(i, int(os.environ.get("GROUP_RANK", "0")) if shared else rank) for i in range(n_base)]
    workers = max(1, min(n_base, (os.cpu_count() or 8) - 1))
    if workers == 1:
        pairs = [synth.make_pair_job(sp) for sp in specs]
    else:
        with mp.get_context("spawn").Pool(workers) as pool:
            pairs = pool.map(synth.make_pair_job, specs)
    if shared:
        tmp = shared + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, **{f"s{i}": p[0] for i, p in enumerate(pairs)}, **{f"t{i}": p[1] for i, p in enumerate(pairs)},
                 **{f"g{i}": np.asarray(p[2]) for i, p in enumerate(pairs)})
        os.replace(tmp, shared)  # (atomic: a reader never sees half a file)
    return [(src, tgt, T_gt) for (src, tgt, T_gt) in pairs]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=4096, help="scan pairs per GPU per step (= per batch): 4 096 pairs = ~113 GB of the 288 GB in use (config.hbm_bytes_in_use); 2 048 (the default of "
                    "rounds 3-4) is 5 %% slower per pair, 1 024: 10 %%, 128: 50 %%")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--leaf", type=float, default=0.2)
    ap.add_argument("--cell", type=float, default=0.0, help="voxel-hash cell (m); 0 = library default")
    ap.add_argument("--inflight", type=int, default=1, help="batch objects in flight, each on its own HIP stream (double buffering): while one batch is in "
                    "its LM iterations the next one is prefiltered and indexed (--inflight 2 --pairs 512: +8 %% pairs/s in the same memory)")
    ap.add_argument("--bases", type=int, default=N_BASE, help="distinct ray-cast pairs per rank (each is its own scene, pose and motion)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the h2d_inclusive block (3 extra steps from host buffers after the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="A/B aid: no HIP events around the kernels (the JSON line then has no roofline block)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--host-inputs", action="store_true", help="hand HOST buffers to the C-ABI every step (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--sor", action="store_true", help="prefilter with the reference's LAUNCH defaults instead of BASELINE's: leaf 0.1 m, StatisticalOutlierRemoval "
                    "mean_k 30 / stddev 1.2 (points_prefiltering.launch.xml:2-5) - a different workload, never the headline value; its line carries the SOR row")
    args = ap.parse_args()
    n_base = max(1, args.bases)
    rank_env = int(os.environ.get("RANK", "0"))
    base = base_pairs(rank_env, n_base)  # before anything touches the GPU

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # BENCH_BOOTSTRAP_ONLY: the rank bootstrap alone - base pairs handed over, process group up, barrier, the shared file gone,
    # one max-over-ranks reduce, rank 0's line - with the GPU section stubbed out: tests/test_bench_host.py runs it with 8 local ranks
    # on a machine without a GPU (the pool has no multi-GPU box: the host side of `--gpus 8` is what can be shown)
    bootstrap_only = bool(os.environ.get("BENCH_BOOTSTRAP_ONLY"))
    if not bootstrap_only and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    # BENCH_DEVICE / BENCH_BACKEND exist so that the multi-rank code path can be exercised on a box with ONE GPU (every rank
    # on device 0, gloo for the two tiny collectives); the driver's runs use neither.
    dev_index = int(os.environ.get("BENCH_DEVICE", local_rank))
    dev = None
    if not bootstrap_only:
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        dist.barrier()  # every rank has read the shared base pairs: local rank 0 takes the file out of /dev/shm again
        if local_rank == 0:
            try:
                os.remove(shared_bases_path(n_base))
            except OSError:
                pass
    if bootstrap_only:
        from lidar_graph_slam_amd import sharding

        slowest = sharding.max_over_ranks(1.0 + 0.001 * rank)  # (the reduce the timed region ends with)
        checksum = float(sum(np.asarray(b[0], np.float64).sum() + np.asarray(b[1], np.float64).sum() for b in base))
        sums = [None] * world
        if world > 1:
            dist.all_gather_object(sums, checksum)
        else:
            sums = [checksum]
        if rank == 0:
            print(json.dumps({"bootstrap_only": True, "n_gpus": world, "max_over_ranks": slowest, "base_pairs_identical_on_every_rank": len(set(sums)) == 1,
                              "shared_file_left_behind": os.path.exists(shared_bases_path(n_base)), "local_world": int(os.environ.get("LOCAL_WORLD_SIZE", "1"))}))
        if world > 1:
            dist.destroy_process_group()
        return

    from lidar_graph_slam_amd import _lib as L
    from lidar_graph_slam_amd import sharding, synth
    from lidar_graph_slam_amd.context import Context
    from lidar_graph_slam_amd.registration import BatchRegistration

    ctx = Context(dev_index)
    # an explicit torch stream: everything torch enqueues below (input synthesis) and everything the library enqueues share
    # it, so device inputs are ordered with the library's work without host synchronisation
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    ctx.set_stream(main_stream.cuda_stream)
    if args.cell > 0:
        ctx.set_grid_cell(args.cell)
    dev_name, cus = ctx.device_info()

    # ---- synthetic inputs, resident in HBM -----------------------------------------------------------------------
    P = args.pairs
    cap = base[0][0].shape[0]
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    src_d = torch.empty((P, cap, 4), dtype=torch.float32, device=dev)
    tgt_d = torch.empty((P, cap, 4), dtype=torch.float32, device=dev)
    base_src = [torch.from_numpy(b[0]).to(dev) for b in base]
    base_tgt = [torch.from_numpy(b[1]).to(dev) for b in base]
    world_T = []
    for p in range(P):
        k = p % n_base
        yaw = float(torch.rand(1, generator=g)) * 2 * np.pi
        tx, ty = (float(torch.rand(1, generator=g)) - 0.5) * 40.0, (float(torch.rand(1, generator=g)) - 0.5) * 40.0
        W = torch.from_numpy(synth.se3(tx, ty, 0.0, 0.0, 0.0, yaw).astype(np.float32)).to(dev)
        world_T.append(W.cpu().numpy())
        for dst, b in ((src_d, base_src[k]), (tgt_d, base_tgt[k])):
            valid = (b[:, :3].abs().sum(dim=1) > 0).unsqueeze(1)  # keep no-return points at the origin
            xyz = b[:, :3] @ W[:3, :3].T + W[:3, 3]
            dst[p, :, :3] = torch.where(valid, xyz, torch.zeros_like(xyz))
            dst[p, :, 3] = b[:, 3]
    counts = np.full(P, cap, dtype=np.int32)
    torch.cuda.synchronize()
    # NOTE: rotating both clouds of a pair by the same world transform W keeps the sensor-frame relative pose:
    # aligned pose T' satisfies T' = W T_gt W^-1.

    params = dict(max_iterations=args.iters, force_iterations=1)
    # --inflight 2: two batch objects on two HIP streams, used alternately (double buffering): while one batch is in its
    # latency-bound LM iterations the next one is prefiltered and indexed.  Every step still processes one full batch.
    batches = [BatchRegistration(P, P, cap, ctx=ctx, **params) for _ in range(max(1, args.inflight))]
    for b_ in batches:
        b_.set_timing(not args.no_kernel_timing)
    streams = []
    if len(batches) > 1:
        streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in batches]  # LM iterations ahead of the (low-priority) indexing chains
        for b_, s_ in zip(batches, streams):
            b_.set_stream(s_.cuda_stream)
    if args.sor:
        args.leaf = 0.1
        pre = L.prefilter_params(min_distance=1.0, leaf_size=0.1, mean_k=30, stddev_mul=1.2)
    else:
        pre = L.prefilter_params(min_distance=1.0, leaf_size=args.leaf, mean_k=0)
    step_no = [0]
    pending = [None] * len(batches)

    if args.host_inputs:
        src_h, tgt_h = src_d.cpu().numpy(), tgt_d.cpu().numpy()

    def step():
        """One batch through the hot path.  With several batches in flight the results returned are those of the batch
        that was enqueued len(batches)-1 steps ago (drain() collects the rest)."""
        k = step_no[0] % len(batches)
        step_no[0] += 1
        batch = batches[k]
        out = None
        if pending[k]:
            out = batch.results(raw=True)  # waits for that batch's stream only; the poses of all pairs, in host memory
            collect(batch)
            pending[k] = None
        if args.host_inputs:
            batch.set_targets(tgt_h, counts, pre)
            batch.set_sources(src_h, counts, pre)
        else:
            if os.environ.get("BENCH_SRC_FIRST"):
                batch.set_sources_dev(src_d.data_ptr(), counts, pre)
                batch.set_targets_dev(tgt_d.data_ptr(), counts, pre)
            else:
                batch.set_targets_dev(tgt_d.data_ptr(), counts, pre)
                batch.set_sources_dev(src_d.data_ptr(), counts, pre)
        batch.align()
        pending[k] = True
        return out

    def drain():
        out = None
        for k, batch in enumerate(batches):
            if pending[k]:
                out = batch.results(raw=True)
                collect(batch)
                pending[k] = None
        return out

    ktimes = {}  # kernel family -> [total ms, launches] over the timed region

    def collect(batch):
        for name, (ms, nl) in batch.kernel_times().items():
            e = ktimes.setdefault(name, [0.0, 0])
            e[0] += ms
            e[1] += nl

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    ktimes.clear()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        r_ = step()
        res = r_ if r_ is not None else res
    r_ = drain()
    res = r_ if r_ is not None else res
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed)  # MAX over ranks (RCCL all-reduce of one double)

    # ---- the same kernels BY THEMSELVES: two more steps of one batch with nothing overlapping (preparation chains one after the
    # other on the batch's stream, no second batch in flight).  Not part of the timed region: it tells what a kernel costs when it
    # has the machine to itself, next to what it costs in the pipelined run above.
    iso = {}
    if not args.no_kernel_timing and rank == 0:
        b0 = batches[0]
        b0.set_side_streams(False)
        for _ in range(2):
            b0.set_targets_dev(tgt_d.data_ptr(), counts, pre)
            b0.set_sources_dev(src_d.data_ptr(), counts, pre)
            b0.align()
            b0.results(raw=True)
            for name, (ms, nl) in b0.kernel_times().items():
                e = iso.setdefault(name, [0.0, 0])
                e[0] += ms
                e[1] += nl
        b0.set_side_streams(None)

    # ---- what the timed batch looked like: iterations-to-convergence per pair (the iteration at which fast_gicp's is_converged
    # first held - the bench forces exactly --iters iterations, so this is what each pair would have run without forcing) and
    # the share of the points the k-NN tile search handed over to the one-wave-per-query kernel
    diag = batches[(step_no[0] - 1) % len(batches)].diagnostics() if rank == 0 else None

    # ---- the PCIe-inclusive rate (SURVEY 8d "H2D included and also reported separately"): three more steps, the same batch handed
    # over as HOST buffers (pageable memory) through mi355_batch_set_targets / _sources.  Never the headline value.
    h2d = None
    if rank == 0 and not args.no_h2d and not args.host_inputs:
        b0 = batches[0]
        src_h, tgt_h = src_d.cpu().numpy(), tgt_d.cpu().numpy()
        n_h2d = 3
        for it in range(n_h2d + 1):
            if it == 1:
                torch.cuda.synchronize()
                th0 = time.perf_counter()
            b0.set_targets(tgt_h, counts, pre)
            b0.set_sources(src_h, counts, pre)
            b0.align()
            res_h = b0.results(raw=True)
        th = (time.perf_counter() - th0) / n_h2d
        b0.kernel_times()  # (drop the brackets of these steps)
        same = bool(np.array_equal(np.asarray(res_h["T"]), np.asarray(res["T"]))) if len(batches) == 1 else None
        h2d = {"value": round(P / th, 2), "unit": "pairs/s", "ms_per_step": round(th * 1e3, 3), "steps": n_h2d,
               "bytes_per_step": int(2 * P * cap * 16), "poses_bit_equal_to_the_timed_region": same,
               "note": "the same batch from HOST buffers (pageable memory) through mi355_batch_set_targets/_sources: upload + the whole step, one batch at a time; "
                       "not the headline value"}
        del src_h, tgt_h

    hbm_free, hbm_total = torch.cuda.mem_get_info(dev)
    hbm_in_use = int(hbm_total - hbm_free)  # everything on the device: the raw clouds of the batch, the batch object, its workspaces

    # ---- correctness of what was timed: pose error vs ground truth for every pair of this rank ------------------------
    errs = []
    for p in range(P):
        W = world_T[p].astype(np.float64)
        T_expected = W @ base[p % n_base][2] @ np.linalg.inv(W)
        errs.append(synth.pose_error(res["T"][p], T_expected))
    errs = np.array(errs)
    n_src = np.asarray(res["n_source"])
    n_tgt = np.asarray(res["n_target"])

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_pairs = P * world * args.steps
    value = total_pairs / elapsed
    # ---- roofline: every timed kernel family of this rank (every rank runs the same shape), the dominant one on top -------
    n_raw = 2.0 * P * cap                      # raw points entering the prefilter per step (both clouds)
    n_pts = float(n_src.sum() + n_tgt.sum())   # points after the prefilter per step
    units_per_step = {
        "k_gicp_linearize": float(n_src.sum()) * args.iters, "k_gicp_correspond": float(n_src.sum()) * args.iters,
        "k_gicp_correspond_tile": float(n_src.sum()) * ktimes.get("k_gicp_correspond_tile", [0, 0])[1] / max(args.steps, 1),
        "k_knn_cov_tile": n_pts, "k_voxel_centroids": n_raw,
        "sort_pairs(k_sort_*)": n_raw + n_pts, "hash_build(k_hash_build|k_gather_insert)": n_pts,
        "sor(sort,hash,k_knn_cov_tile<SOR>,k_knn_hard<SOR>,threshold,compact)": n_pts,
    }
    gpu_ms_total = sum(v[0] for v in ktimes.values())
    by_kernel = {}
    for name, (ms, nl) in sorted(ktimes.items(), key=lambda kv: -kv[1][0]):
        e = {"total_ms_per_step": round(ms / args.steps, 4), "launches_per_step": round(nl / args.steps, 2),
             "avg_launch_ms": round(ms / max(nl, 1), 4), "share_of_timed_gpu_ms": round(ms / max(gpu_ms_total, 1e-9), 4)}
        if name in ALG_BYTES:
            unit, bpu = ALG_BYTES[name]
            bytes_per_launch = bpu * units_per_step[name] * args.steps / max(nl, 1)
            gbs = bytes_per_launch / (ms / max(nl, 1) * 1e-3) / 1e9 if ms > 0 else 0.0
            e.update({"algorithmic_bytes_per_launch": int(bytes_per_launch), "bytes_per_unit": f"{bpu} B per {unit}",
                      "achieved_GBps": round(gbs, 2), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5)})
        if name in iso and iso[name][1]:
            ims = iso[name][0] / iso[name][1]
            e["isolated_avg_launch_ms"] = round(ims, 4)
            if name in ALG_BYTES and ims > 0:
                e["isolated_frac_of_hbm_peak"] = round(e["algorithmic_bytes_per_launch"] / (ims * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        by_kernel[name] = e
    # the dominant family: by what the kernels cost BY THEMSELVES (isolated duration x launches per step) — live durations of
    # overlapping kernels measure how long a kernel shares the machine, not how much of it it needs
    def weight(k):
        e = by_kernel[k]
        return e.get("isolated_avg_launch_ms", e["avg_launch_ms"]) * e["launches_per_step"]
    dom = max((k for k in by_kernel if k in ALG_BYTES), key=weight, default=None)
    if dom is None:
        if args.no_kernel_timing:
            print(json.dumps({"value": round(value, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3), "note": "A/B run without kernel timing"}))
            return
        raise SystemExit("bench.py: the library reported no kernel timings")
    d = by_kernel[dom]
    # The roofline numbers of the line come from the kernel's OWN duration: HIP events on the launching stream around launches
    # that have the GPU to themselves (the two extra steps above, preparation chains serialised).  That is the duration the
    # rocprofv3 kernel trace of this command reports for the kernel (profiles/README.md recomputes it).  The event brackets of
    # the TIMED region sit on side streams that run concurrently: for kernels that overlap another chain's they also contain
    # waiting / sharing (the k-NN kernels do not overlap each other: the library serialises them) - kept under
    # `timed_region_event_brackets`, never used for `frac`.
    iso_ms = d.get("isolated_avg_launch_ms") or d["avg_launch_ms"]
    achieved = d["algorithmic_bytes_per_launch"] / (iso_ms * 1e-3) / 1e9
    roofline = {
        "kernel": dom,
        "bound": "hbm",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": None,
        "avg_launch_ms": iso_ms,
        "launches_measured": int(iso.get(dom, [0, 0])[1]),
        "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
        "timed_region_event_brackets": {"avg_ms": d["avg_launch_ms"], "launches": int(ktimes[dom][1]), "share_of_summed_brackets": d["share_of_timed_gpu_ms"],
                                        "note": "brackets on the concurrent side streams of the timed region; not used for frac"},
        "note": "kernel family with the largest own duration x launches per step (the ranking of the rocprofv3 kernel-stats CSV under profiles/); "
                "achieved = algorithmic bytes per launch (SURVEY 8d x the units of one launch) / avg_launch_ms, measured live in this run "
                "with HIP events on the launching stream while nothing else runs",
        "by_kernel": by_kernel,
    }
    # whole step: algorithmic bytes of prefilter (<= 32 B per raw point) + covariances (40 B per point) + iterations (80 B per
    # source point and iteration) over the wall time of a step
    e2e_bytes = 32.0 * n_raw + 40.0 * n_pts + 80.0 * float(n_src.sum()) * args.iters
    e2e_gbs = e2e_bytes / (elapsed / args.steps) / 1e9
    roofline["end_to_end"] = {"algorithmic_bytes_per_step": int(e2e_bytes), "bytes_per_pair": int(e2e_bytes / P), "achieved": round(e2e_gbs, 2),
                              "unit": "GB/s", "frac": round(e2e_gbs / HBM_PEAK_GBS, 5)}

    # HBM traffic of the dominant kernel from the PMC passes committed under profiles/ (same command, same shape)
    for tname in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
        tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
        if tj.get("pairs") == P and tj.get("kernel") == dom.split("(")[0]:
            roofline["traffic"] = tj["hbm_bytes_per_launch"]
            roofline["traffic_source"] = tj.get("source", "profiles/" + tname)
            break

    # ... and the share of the chip's VALU issue slots the kernels fill (SQ counters of the same command: profiles/<round>_issue.json):
    # the search kernels are bound by their instruction stream, and the HBM fraction alone would not say so
    for iname in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_issue.json") and "ndt" not in f), reverse=True):
        ij = json.load(open(os.path.join(ROOT, "profiles", iname)))
        if ij.get("pairs") != P:
            continue
        for fam_name, e in by_kernel.items():
            hit = ij["kernels"].get({"sort_pairs": "k_sort_tile", "hash_build": "k_hash_build"}.get(fam_name.split("(")[0], fam_name.split("(")[0]))
            if hit:
                e["valu_issue_frac"] = hit["valu_issue_frac"]
        hit = ij["kernels"].get(dom.split("(")[0])
        if hit:
            roofline["valu_issue_frac"] = hit["valu_issue_frac"]
            roofline["valu_issue_frac_source"] = f"profiles/{iname}: {ij.get('unit', '')}"
        break

    # ---- accuracy gate: a fast kernel whose results differ from the reference's is not done -------------------------------
    if errs[:, 0].max() > TOL_TRUTH_T or errs[:, 1].max() > TOL_TRUTH_R:
        raise SystemExit(f"bench.py: pose error vs ground truth {errs[:, 0].max():.4f} m / {errs[:, 1].max():.2e} rad exceeds "
                         f"{TOL_TRUTH_T} m / {TOL_TRUTH_R} rad - refusing to report a throughput for wrong poses")

    fc = diag["first_converged"]
    conv = fc[fc >= 0]
    batch_stats = {
        "iterations_to_convergence": {
            "min": int(conv.min()) if len(conv) else None, "median": float(np.median(conv)) if len(conv) else None,
            "mean": round(float(conv.mean()), 3) if len(conv) else None, "max": int(conv.max()) if len(conv) else None,
            "histogram": {str(int(v)): int(c) for v, c in zip(*np.unique(conv, return_counts=True))},
            "never_within_the_forced_iterations": int((fc < 0).sum()),
            "note": "the iteration count at which fast_gicp's is_converged first held for a pair; the bench runs exactly --iters iterations for every pair",
        },
        "knn_hand_over_fraction": {"targets": round(float(diag["knn_handed_over"][0]) / max(float(diag["knn_points"][0]), 1.0), 6),
                                   "sources": round(float(diag["knn_handed_over"][1]) / max(float(diag["knn_points"][1]), 1.0), 6),
                                   "note": "points the k-NN tile search handed over to k_knn_hard (one wave per query), of all points after the prefilter"},
        "points_after_prefilter": {"min": int(min(n_src.min(), n_tgt.min())), "max": int(max(n_src.max(), n_tgt.max()))},
    }

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle_py as O

        def cpu_prefilter(c):
            c = O.voxel_grid(O.distance_filter(c, 1.0), args.leaf)[0]
            return O.sor(c, 30, 1.2)[0] if args.sor else c

        def cpu_pair(k, gp):
            s, t, _ = base[k % n_base]
            return O.gicp_align(cpu_prefilter(s), cpu_prefilter(t), gp)

        # one process, OpenMP inside a pair: 4 threads = the reference's own configuration (lidar_scan_matcher.param.yaml:10), and
        # 16 (all hardware threads on ONE pair oversubscribe the memory system: 10 pairs/s at 128 vs 24 at 16 in round 1)
        hw = O.max_threads()
        by_threads = {}
        for threads in sorted({min(4, hw), min(16, hw)}):
            gp = O.gicp_params(max_iterations=args.iters, force_iterations=1, num_threads=threads)
            cpu_pair(0, gp)
            cpu_pair(1, gp)  # warm the OpenMP pool and the page cache
            n_done, tc0 = 0, time.perf_counter()
            while (time.perf_counter() - tc0 < args.cpu_seconds / 2 or n_done < 4) and n_done < 2000:
                cpu_pair(n_done, gp)
                n_done += 1
            by_threads[threads] = (n_done, time.perf_counter() - tc0)
        # the WHOLE host, the way independent pairs saturate it: hw / 4 workers, each registering its own pairs with the
        # reference's 4 OpenMP threads (the oracle is a C library: ctypes releases the GIL, every worker thread gets its own
        # OpenMP team).  This is the fair CPU figure for a batch workload (BASELINE.md 2 (ii) "all host cores").
        import threading

        n_workers = max(1, hw // 4)
        gp4 = O.gicp_params(max_iterations=args.iters, force_iterations=1, num_threads=4)
        done = [0] * n_workers
        t_end = [0.0]

        def worker(w):
            k = w
            while time.perf_counter() < t_end[0]:
                cpu_pair(k, gp4)
                k += n_workers
                done[w] += 1

        th = [threading.Thread(target=worker, args=(w,)) for w in range(n_workers)]
        tw0 = time.perf_counter()
        t_end[0] = tw0 + args.cpu_seconds
        for t in th:
            t.start()
        for t in th:
            t.join()
        whole_host = (sum(done), time.perf_counter() - tw0)
        threads = max(by_threads, key=lambda t: by_threads[t][0] / by_threads[t][1])
        n_done, tc = by_threads[threads]
        # parity on the SAME input: N_ORACLE pairs of the timed batch, spread over the batch and over the distinct scenes, exactly
        # as the GPU saw them (world-transformed), through the oracle
        gp = O.gicp_params(max_iterations=args.iters, force_iterations=1, num_threads=min(16, hw))
        sample = sorted({((k * P) // (N_ORACLE - 1) + k) % P for k in range(N_ORACLE - 1)} | {int(np.argmax(errs[:, 0]))})
        dts, drs = [], []
        for p_ in sample:
            s0, t0_ = src_d[p_].cpu().numpy(), tgt_d[p_].cpu().numpy()
            ref0 = O.gicp_align(cpu_prefilter(s0), cpu_prefilter(t0_), gp)
            dt_, dr_ = synth.pose_error(res["T"][p_], ref0["T"])
            dts.append(dt_)
            drs.append(dr_)
        dt, dr = max(dts), max(drs)
        if dt > TOL_ORACLE_T or dr > TOL_ORACLE_R:
            raise SystemExit(f"bench.py: GPU pose differs from the oracle's on the same pair by {dt:.2e} m / {dr:.2e} rad "
                             f"(pairs {sample}; tolerance {TOL_ORACLE_T} m / {TOL_ORACLE_R} rad)")
        model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
        except OSError:
            pass
        cpu = {
            "value": round(whole_host[0] / whole_host[1], 3),
            "unit": "pairs/s",
            "cores": n_workers * 4,
            "kind": "port",
            "sample": f"{whole_host[0]} of the same 64k-pt pairs (distance filter + voxel {args.leaf} m + k=20 covariances + {args.iters} LM iterations) "
                      f"in {whole_host[1]:.1f} s: oracle/liboracle.so, {n_workers} concurrent workers x 4 OpenMP threads = every hardware thread of the "
                      f"host, each worker registering its own pairs (pair-level parallelism, the batch workload's fair CPU form)",
            "one_process": {"pairs_per_s_by_omp_threads": {str(k): round(v[0] / v[1], 2) for k, v in by_threads.items()},
                            "note": "one pair at a time, OpenMP inside the pair: 4 threads is the reference's configuration (lidar_scan_matcher.param.yaml:10)"},
            "reference_config_threads": 4,
            "host_hw_threads": hw,
            "cpu_model": model,
            "gpu_vs_cpu_pose_dt_m": round(dt, 9),
            "gpu_vs_cpu_pose_dr_rad": round(dr, 9),
            "gpu_vs_cpu_pairs_compared": sample,
            "gpu_vs_cpu_pose_dt_m_by_pair": [round(v, 9) for v in dts],
        }

    out = {
        "metric": "scan_pairs_per_sec",
        "value": round(value, 2),
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 per-point maths, f64 accumulation",  # (distances, transforms, Mahalanobis matrices in fp32 on 24-byte covariances; every sum - covariances,
                                                          # normal equations, trial costs - in fp64: the 16 oracle-checked pairs agree with the all-fp64 oracle to ~1e-7 m)
        "data": "synthetic",
        "inputs": "host buffers (PCIe-inclusive; NOT the headline)" if args.host_inputs else "resident in HBM",
        **({"not_the_headline": "--sor: the reference's launch-file prefilter (leaf 0.1 m + StatisticalOutlierRemoval 30 / 1.2), not BASELINE configs[1]"} if args.sor else {}),
        "config": {
            "workload": ("BASELINE configs[1]: 64k-pt scan pairs, distance filter 1.0 m + voxel 0.2 m + GICP k=20, exactly 20 LM iterations, batched" if not args.sor else
                         "64k-pt scan pairs through the reference's launch-file prefilter (distance 1.0 m + voxel 0.1 m + SOR 30 / 1.2) + GICP k=20, exactly 20 LM iterations, batched"),
            "pairs_per_gpu_per_step": P,
            "hbm_bytes_in_use": hbm_in_use,
            "distinct_ray_cast_pairs": n_base,
            "batches_in_flight": len(batches),
            "raw_points_per_cloud": int(cap),
            "points_after_voxel_src_mean": float(n_src.mean()),
            "points_after_voxel_tgt_mean": float(n_tgt.mean()),
            "gicp_iterations": args.iters,
            "voxel_leaf_m": args.leaf,
            "device": dev_name,
            "compute_units": cus,
            "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective",
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
        "h2d_inclusive": h2d,
        "batch": batch_stats,
        "accuracy": {
            "pose_err_vs_truth_dt_m_max": round(float(errs[:, 0].max()), 5),
            "pose_err_vs_truth_dt_m_median": round(float(np.median(errs[:, 0])), 5),
            "pose_err_vs_truth_dr_rad_max": round(float(errs[:, 1].max()), 6),
        },
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
