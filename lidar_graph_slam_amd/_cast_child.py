"""Child interpreter of synth.cast_jobs_in_child: owns the pool of ray-casting workers, never touches the GPU.
usage: python -m lidar_graph_slam_amd._cast_child <directory holding jobs.pkl> [workers]  ->  <directory>/scans.npz"""
import multiprocessing as mp
import os
import pickle
import sys

import numpy as np

from lidar_graph_slam_amd import synth


def main(argv):
    d = argv[0]
    workers = int(argv[1]) if len(argv) > 1 else max(1, min(64, (os.cpu_count() or 8) - 2))
    with open(os.path.join(d, "jobs.pkl"), "rb") as f:
        jobs = pickle.load(f)
    if workers <= 1:
        scans = [synth.cast_job(j) for j in jobs]
    else:
        with mp.get_context("spawn").Pool(workers) as pool:
            scans = pool.map(synth.cast_job, jobs, chunksize=4)
    np.savez(os.path.join(d, "scans.npz"), **{f"s{i}": s for i, s in enumerate(scans)})


if __name__ == "__main__":
    main(sys.argv[1:])
