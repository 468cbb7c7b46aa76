#!/usr/bin/env python3
"""Generates tests/golden/golden_small.npz — small seeded inputs with the outputs on which the C++ oracle
(oracle/liboracle.so) and the independent NumPy/SciPy restatement (oracle/numpy_ref.py) AGREE.

The reference (RyuYamamoto/lidar_graph_slam) ships no golden vectors, tests or fixtures and its numeric engines
(fast_gicp, ndt_omp, PCL) cannot be built here — PARITY UNPINNED — so these fixtures pin the
restated algorithms instead of upstream binaries.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lidar_graph_slam_amd import synth  # noqa: E402
from oracle import numpy_ref as NR  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def main():
    src_raw, tgt_raw, T_gt = synth.make_pair(n_azimuth=48, seed=5)  # 3 072 rays per scan
    out = {"src_raw": src_raw, "tgt_raw": tgt_raw, "T_gt": T_gt}
    # ---- prefilter stages
    d = O.distance_filter(src_raw, 1.0)
    assert np.array_equal(d, NR.distance_filter(src_raw, 1.0))
    v, ovf = O.voxel_grid(d, 0.2)
    assert not ovf and np.allclose(v, NR.voxel_grid(d, 0.2), atol=2e-5)
    s_out, keep, md = O.sor(v, 30, 1.2)
    _, keep2, md2 = NR.sor(v, 30, 1.2)
    assert np.array_equal(keep, keep2) and np.allclose(md, md2, rtol=2e-5)
    out.update(dist=d, voxel=v, sor_keep=keep, sor_mean_dist=md, sor_out=s_out)
    tv = O.voxel_grid(O.distance_filter(tgt_raw, 1.0), 0.2)[0]
    out["tgt_voxel"] = tv
    # ---- k-NN / covariances
    idx, d2 = O.knn(tv, v, 20)
    idx_b, d2_b = O.knn(tv, v, 20, brute_force=True)
    assert np.array_equal(idx, idx_b) and np.array_equal(d2, d2_b)
    out.update(knn_idx=idx, knn_d2=d2)
    cs, ct = O.covariances(v, 20, O.REG_PLANE), O.covariances(tv, 20, O.REG_PLANE)
    agree = np.isclose(cs, NR.covariances(v, 20, "PLANE"), rtol=1e-7, atol=1e-9).all(axis=(1, 2))
    assert agree.mean() > 0.995
    out.update(cov_src=cs, cov_tgt=ct, cov_agree=agree)
    # ---- one linearisation + alignment
    guess = synth.se3(0.3, -0.1, 0.02, 0.01, -0.01, 0.02)
    H, b, err, corr, _ = O.gicp_linearize(v, cs, tv, ct, guess, 2.0)
    H2, b2, err2, corr2 = NR.gicp_linearize(v, cs, tv, ct, guess, 2.0)
    assert np.array_equal(corr, corr2) and np.allclose(H, H2, rtol=1e-9) and np.isclose(err, err2, rtol=1e-9)
    out.update(lin_guess=guess, lin_H=H, lin_b=b, lin_err=err, lin_corr=corr)
    r = O.gicp_align(v, tv, O.gicp_params(max_iterations=8, force_iterations=1))
    T2, _, _ = NR.gicp_align(v, tv, max_iterations=8, force_iterations=True)
    dt, dr = synth.pose_error(r["T"], T2)
    assert dt < 1e-4 and dr < 1e-5, (dt, dr)
    out.update(gicp_T=r["T"], gicp_fitness=O.fitness(v, tv, r["T"]))
    # ---- NDT
    c = O.ndt_cells(tv, O.ndt_params(resolution=2.0))
    ref, min_b, div_b = NR.ndt_cells(tv, 2.0)
    both = [k for k in c["idx"][c["n"] >= 6].tolist() if k in ref]
    out.update(ndt_idx=c["idx"], ndt_n=c["n"], ndt_mean=c["mean"], ndt_icov=c["icov"], ndt_agree_keys=np.array(both))
    p6 = np.array([0.8, 0.05, 0.01, 0.004, -0.003, 0.03])
    sc, g, Hn = O.ndt_derivatives(v, tv, p6, O.ndt_params(resolution=2.0))
    out.update(ndt_p6=p6, ndt_score=sc, ndt_grad=g, ndt_hess=Hn)
    rn = O.ndt_align(v, tv, O.ndt_params(resolution=2.0))  # the ndt_omp default: the line search never iterates
    out.update(ndt_T=rn["T"], ndt_iterations=rn["iterations"])
    rm = O.ndt_align(v, tv, O.ndt_params(resolution=2.0, legacy_mt_skip=0))  # PCL >= 1.10: More-Thuente iterates
    out.update(ndt_T_mt=rm["T"], ndt_iterations_mt=rm["iterations"])
    # ---- "GICP" (pclomp's BFGS GICP): the first outer iteration (one full BFGS minimisation), and the converged pose
    b1 = O.gicp_bfgs_align(v, tv, O.gicp_bfgs_params(max_iterations=1))
    bc = O.gicp_bfgs_align(v, tv)
    out.update(bfgs_T_1=b1["T"], bfgs_cost_1=b1["error"], bfgs_T=bc["T"], bfgs_iterations=bc["iterations"])
    path = os.path.join(ROOT, "tests", "golden", "golden_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; src", v.shape, "tgt", tv.shape, "gicp err vs truth", synth.pose_error(r["T"], T_gt))


if __name__ == "__main__":
    main()
