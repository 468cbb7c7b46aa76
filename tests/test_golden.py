"""Golden-vector tests.  tests/golden/golden_small.npz (made by tests/golden/make_golden.py) holds small seeded inputs
and the outputs on which the C++ oracle and the NumPy/SciPy restatement agreed when the fixture was generated.
CPU part: the oracle still reproduces them (guards against oracle drift).  GPU part: the HIP path reproduces them
through the C-ABI without the oracle being present at all."""
import numpy as np
import pytest

from lidar_graph_slam_amd import synth


def test_oracle_reproduces_golden(golden):
    from oracle import oracle_py as O

    G = golden
    np.testing.assert_array_equal(O.distance_filter(G["src_raw"], 1.0), G["dist"])
    np.testing.assert_array_equal(O.voxel_grid(G["dist"], 0.2)[0], G["voxel"])
    out, keep, md = O.sor(G["voxel"], 30, 1.2)
    np.testing.assert_array_equal(keep, G["sor_keep"])
    np.testing.assert_array_equal(out, G["sor_out"])
    idx, d2 = O.knn(G["tgt_voxel"], G["voxel"], 20)
    np.testing.assert_array_equal(idx, G["knn_idx"])
    np.testing.assert_array_equal(d2, G["knn_d2"])
    np.testing.assert_allclose(O.covariances(G["voxel"], 20, O.REG_PLANE), G["cov_src"], rtol=1e-9, atol=1e-12)
    H, b, err, corr, _ = O.gicp_linearize(G["voxel"], G["cov_src"], G["tgt_voxel"], G["cov_tgt"], G["lin_guess"], 2.0)
    np.testing.assert_array_equal(corr, G["lin_corr"])
    np.testing.assert_allclose(H, G["lin_H"], rtol=1e-10)
    np.testing.assert_allclose(err, G["lin_err"], rtol=1e-10)
    r = O.gicp_align(G["voxel"], G["tgt_voxel"], O.gicp_params(max_iterations=8, force_iterations=1))
    dt, dr = synth.pose_error(r["T"], G["gicp_T"])
    assert dt < 1e-6 and dr < 1e-6
    b1 = O.gicp_bfgs_align(G["voxel"], G["tgt_voxel"], O.gicp_bfgs_params(max_iterations=1))
    np.testing.assert_allclose(b1["T"], G["bfgs_T_1"], atol=1e-6)
    np.testing.assert_allclose(b1["error"], float(G["bfgs_cost_1"]), rtol=1e-6)
    sc, g, Hn = O.ndt_derivatives(G["voxel"], G["tgt_voxel"], G["ndt_p6"], O.ndt_params(resolution=2.0))
    np.testing.assert_allclose(sc, G["ndt_score"], rtol=1e-6)
    np.testing.assert_allclose(g, G["ndt_grad"], rtol=1e-5, atol=1e-5 * np.abs(G["ndt_grad"]).max())


@pytest.mark.gpu
def test_hip_path_reproduces_golden(golden):
    from lidar_graph_slam_amd import filters as F
    from lidar_graph_slam_amd import registration as R

    G = golden
    np.testing.assert_array_equal(F.distance_filter(G["src_raw"], 1.0), G["dist"])
    vg = F.VoxelGrid()
    vg.setLeafSize(0.2, 0.2, 0.2)
    vg.setInputCloud(G["dist"])
    np.testing.assert_array_equal(vg.filter(), G["voxel"])
    sor = F.StatisticalOutlierRemoval()
    sor.setMeanK(30)
    sor.setStddevMulThresh(1.2)
    sor.setInputCloud(G["voxel"])
    out = sor.filter()
    np.testing.assert_array_equal(sor.keep_mask, G["sor_keep"])
    np.testing.assert_array_equal(out, G["sor_out"])
    idx, d2 = F.knn(G["tgt_voxel"], G["voxel"], 20)
    np.testing.assert_array_equal(idx, G["knn_idx"])
    np.testing.assert_array_equal(d2, G["knn_d2"])
    c = F.covariances(G["voxel"], 20)
    ok = G["cov_agree"]
    np.testing.assert_allclose(c[ok], G["cov_src"][ok], rtol=0, atol=3e-6)
    reg = R.FastGICP(max_iterations=8, force_iterations=1)
    reg.setInputTarget(G["tgt_voxel"])
    reg.setInputSource(G["voxel"])
    H, b, err, corr = reg.linearize(G["lin_guess"])
    np.testing.assert_array_equal(corr, G["lin_corr"])
    np.testing.assert_allclose(H, G["lin_H"], rtol=2e-5, atol=2e-5 * np.abs(G["lin_H"]).max())
    np.testing.assert_allclose(err, G["lin_err"], rtol=2e-5)
    reg.align()
    dt, dr = synth.pose_error(reg.getFinalTransformation(), G["gicp_T"])
    assert dt <= 1e-3 and dr <= 1e-4, (dt, dr)
    np.testing.assert_allclose(reg.getFitnessScore(), float(G["gicp_fitness"]), rtol=1e-3)
    ndt = R.NormalDistributionsTransform(resolution=2.0)
    ndt.setInputTarget(G["tgt_voxel"])
    ndt.setInputSource(G["voxel"])
    sc, g, Hn = ndt.derivatives(G["ndt_p6"])
    np.testing.assert_allclose(sc, G["ndt_score"], rtol=1e-4)
    np.testing.assert_allclose(g, G["ndt_grad"], rtol=1e-3, atol=1e-3 * np.abs(G["ndt_grad"]).max())
    ndt.align()  # default: ndt_omp's line search (never iterates)
    dt, dr = synth.pose_error(ndt.getFinalTransformation(), G["ndt_T"])
    assert dt <= 0.01 and dr <= 2e-3, (dt, dr)
    ndt_mt = R.NormalDistributionsTransform(resolution=2.0, legacy_mt_skip=0)  # PCL >= 1.10: More-Thuente iterates
    ndt_mt.setInputTarget(G["tgt_voxel"])
    ndt_mt.setInputSource(G["voxel"])
    ndt_mt.align()
    dt, dr = synth.pose_error(ndt_mt.getFinalTransformation(), G["ndt_T_mt"])
    assert dt <= 0.01 and dr <= 2e-3, (dt, dr)
    bf = R.GeneralizedIterativeClosestPoint(max_iterations=1)  # "GICP": one outer iteration = one full BFGS minimisation
    bf.setInputTarget(G["tgt_voxel"])
    bf.setInputSource(G["voxel"])
    bf.align()
    import parity

    parity.assert_fixed(bf.getFinalTransformation(), G["bfgs_T_1"], "golden/bfgs_outer1", t=2e-3, r=2e-4)  # one outer iteration
    assert abs(bf._result["error"] - float(G["bfgs_cost_1"])) <= 2e-3 * float(G["bfgs_cost_1"])
