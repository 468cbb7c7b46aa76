"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so (see oracle/oracle.h).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
PARITY UNPINNED (no golden vectors exist in the reference).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS = range(5)
OPT_LM, OPT_GN = 0, 1


class GicpParams(C.Structure):
    _fields_ = [
        ("k_correspondences", C.c_int),
        ("max_iterations", C.c_int),
        ("transformation_epsilon", C.c_double),
        ("rotation_epsilon", C.c_double),
        ("max_correspondence_distance", C.c_double),
        ("regularization_method", C.c_int),
        ("optimizer", C.c_int),
        ("lm_max_iterations", C.c_int),
        ("lm_init_lambda_factor", C.c_double),
        ("force_iterations", C.c_int),
        ("num_threads", C.c_int),
    ]


class NdtParams(C.Structure):
    _fields_ = [
        ("transformation_epsilon", C.c_double),
        ("step_size", C.c_double),
        ("resolution", C.c_double),
        ("max_iterations", C.c_int),
        ("outlier_ratio", C.c_double),
        ("min_points_per_voxel", C.c_int),
        ("min_covar_eigvalue_mult", C.c_double),
        ("legacy_mt_skip", C.c_int),
        ("num_threads", C.c_int),
    ]


class Result(C.Structure):
    _fields_ = [
        ("T", C.c_float * 16),
        ("converged", C.c_int),
        ("iterations", C.c_int),
        ("error", C.c_double),
        ("H", C.c_double * 36),
        ("trans_probability", C.c_double),
        ("ms_setup", C.c_double),
        ("ms_align", C.c_double),
    ]


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(path):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"] + (["-B"] if force else []))
    return path


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _f4(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 4, a.shape
    return a


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def gicp_params(**kw) -> GicpParams:
    p = GicpParams()
    lib().orc_gicp_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def ndt_params(**kw) -> NdtParams:
    p = NdtParams()
    lib().orc_ndt_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def max_threads() -> int:
    return int(lib().orc_max_threads())


def distance_filter(pts, min_distance):
    pts = _f4(pts)
    out = np.empty_like(pts)
    n = C.c_size_t(0)
    lib().orc_distance_filter(_p(pts, C.c_float), C.c_size_t(len(pts)), C.c_double(min_distance), _p(out, C.c_float), C.byref(n))
    return out[: n.value].copy()


def voxel_grid(pts, leaf):
    pts = _f4(pts)
    out = np.empty_like(pts)
    n = C.c_size_t(0)
    ovf = C.c_int(0)
    lib().orc_voxel_grid(_p(pts, C.c_float), C.c_size_t(len(pts)), C.c_float(leaf), _p(out, C.c_float), C.byref(n), C.byref(ovf))
    return out[: n.value].copy(), bool(ovf.value)


def sor(pts, mean_k, stddev_mul):
    pts = _f4(pts)
    out = np.empty_like(pts)
    mask = np.zeros(len(pts), dtype=np.uint8)
    md = np.zeros(len(pts), dtype=np.float32)
    n = C.c_size_t(0)
    rc = lib().orc_sor(_p(pts, C.c_float), C.c_size_t(len(pts)), C.c_int(mean_k), C.c_double(stddev_mul), _p(out, C.c_float),
                       C.byref(n), _p(mask, C.c_uint8), _p(md, C.c_float))
    assert rc == 0
    return out[: n.value].copy(), mask.astype(bool), md


def transform(pts, T):
    pts = _f4(pts)
    T = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
    out = np.empty_like(pts)
    lib().orc_transform(_p(pts, C.c_float), C.c_size_t(len(pts)), _p(T, C.c_float), _p(out, C.c_float))
    return out


def knn(pts, queries, k, brute_force=False):
    pts, queries = _f4(pts), _f4(queries)
    idx = np.empty((len(queries), k), dtype=np.int32)
    d2 = np.empty((len(queries), k), dtype=np.float32)
    rc = lib().orc_knn(_p(pts, C.c_float), C.c_size_t(len(pts)), _p(queries, C.c_float), C.c_size_t(len(queries)), C.c_int(k),
                       _p(idx, C.c_int), _p(d2, C.c_float), C.c_int(1 if brute_force else 0))
    assert rc == 0
    return idx, d2


def fitness(src, tgt, T, max_range=np.finfo(np.float64).max):
    src, tgt = _f4(src), _f4(tgt)
    T = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
    s = C.c_double(0)
    rc = lib().orc_fitness(_p(src, C.c_float), C.c_size_t(len(src)), _p(tgt, C.c_float), C.c_size_t(len(tgt)), _p(T, C.c_float),
                           C.c_double(max_range), C.byref(s))
    assert rc == 0
    return s.value


def covariances(pts, k=20, method=REG_PLANE, num_threads=0):
    pts = _f4(pts)
    covs = np.empty((len(pts), 3, 3), dtype=np.float64)
    rc = lib().orc_covariances(_p(pts, C.c_float), C.c_size_t(len(pts)), C.c_int(k), C.c_int(method), _p(covs, C.c_double),
                               C.c_int(num_threads))
    assert rc == 0
    return covs


def se3_exp(d):
    d = np.ascontiguousarray(d, dtype=np.float64).reshape(6)
    T = np.empty(16, dtype=np.float64)
    lib().orc_se3_exp(_p(d, C.c_double), _p(T, C.c_double))
    return T.reshape(4, 4)


def gicp_linearize(src, src_cov, tgt, tgt_cov, T, max_corr):
    src, tgt = _f4(src), _f4(tgt)
    src_cov = np.ascontiguousarray(src_cov, dtype=np.float64)
    tgt_cov = np.ascontiguousarray(tgt_cov, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    H = np.empty(36)
    b = np.empty(6)
    err = C.c_double(0)
    corr = np.empty(len(src), dtype=np.int32)
    mahal = np.empty((len(src), 3, 3))
    lib().orc_gicp_linearize(_p(src, C.c_float), C.c_size_t(len(src)), _p(src_cov, C.c_double), _p(tgt, C.c_float),
                             C.c_size_t(len(tgt)), _p(tgt_cov, C.c_double), _p(T, C.c_double), C.c_double(max_corr),
                             _p(H, C.c_double), _p(b, C.c_double), C.byref(err), _p(corr, C.c_int), _p(mahal, C.c_double))
    return H.reshape(6, 6), b, err.value, corr, mahal


def gicp_error(src, tgt, corr, mahal, T):
    src, tgt = _f4(src), _f4(tgt)
    corr = np.ascontiguousarray(corr, dtype=np.int32)
    mahal = np.ascontiguousarray(mahal, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    err = C.c_double(0)
    lib().orc_gicp_error(_p(src, C.c_float), C.c_size_t(len(src)), _p(tgt, C.c_float), _p(corr, C.c_int), _p(mahal, C.c_double),
                         _p(T, C.c_double), C.byref(err))
    return err.value


def _result(r: Result):
    return {
        "T": np.array(r.T, dtype=np.float32).reshape(4, 4),
        "converged": bool(r.converged),
        "iterations": int(r.iterations),
        "error": float(r.error),
        "H": np.array(r.H, dtype=np.float64).reshape(6, 6),
        "trans_probability": float(r.trans_probability),
        "ms_setup": float(r.ms_setup),
        "ms_align": float(r.ms_align),
    }


def gicp_align(src, tgt, params: GicpParams | None = None, guess=None):
    src, tgt = _f4(src), _f4(tgt)
    params = params or gicp_params()
    g = np.eye(4, dtype=np.float32) if guess is None else np.ascontiguousarray(guess, dtype=np.float32)
    g = g.reshape(16)
    r = Result()
    rc = lib().orc_gicp_align(_p(src, C.c_float), C.c_size_t(len(src)), _p(tgt, C.c_float), C.c_size_t(len(tgt)), C.byref(params),
                              _p(g, C.c_float), C.byref(r))
    assert rc == 0, rc
    return _result(r)


def ndt_cells(tgt, params: NdtParams | None = None):
    tgt = _f4(tgt)
    params = params or ndt_params()
    n = len(tgt)
    idx = np.empty(n, dtype=np.int64)
    cnt = np.empty(n, dtype=np.int32)
    mean = np.empty((n, 3))
    cov = np.empty((n, 3, 3))
    icov = np.empty((n, 3, 3))
    min_b = np.empty(3, dtype=np.int32)
    div_b = np.empty(3, dtype=np.int32)
    nl = C.c_size_t(0)
    lib().orc_ndt_cells(_p(tgt, C.c_float), C.c_size_t(n), C.byref(params), _p(idx, C.c_int64), _p(cnt, C.c_int), _p(mean, C.c_double),
                        _p(cov, C.c_double), _p(icov, C.c_double), _p(min_b, C.c_int), _p(div_b, C.c_int), C.byref(nl))
    m = nl.value
    return {"idx": idx[:m].copy(), "n": cnt[:m].copy(), "mean": mean[:m].copy(), "cov": cov[:m].copy(), "icov": icov[:m].copy(),
            "min_b": min_b, "div_b": div_b}


def ndt_derivatives(src, tgt, p6, params: NdtParams | None = None, hessian_fp64=False):
    src, tgt = _f4(src), _f4(tgt)
    params = params or ndt_params()
    p6 = np.ascontiguousarray(p6, dtype=np.float64).reshape(6)
    score = C.c_double(0)
    g = np.empty(6)
    H = np.empty(36)
    lib().orc_ndt_derivatives(_p(src, C.c_float), C.c_size_t(len(src)), _p(tgt, C.c_float), C.c_size_t(len(tgt)), C.byref(params),
                              _p(p6, C.c_double), C.byref(score), _p(g, C.c_double), _p(H, C.c_double), C.c_int(1 if hessian_fp64 else 0))
    return score.value, g, H.reshape(6, 6)


def ndt_align(src, tgt, params: NdtParams | None = None, guess=None):
    src, tgt = _f4(src), _f4(tgt)
    params = params or ndt_params()
    g = np.eye(4, dtype=np.float32) if guess is None else np.ascontiguousarray(guess, dtype=np.float32)
    g = g.reshape(16)
    r = Result()
    rc = lib().orc_ndt_align(_p(src, C.c_float), C.c_size_t(len(src)), _p(tgt, C.c_float), C.c_size_t(len(tgt)), C.byref(params),
                             _p(g, C.c_float), C.byref(r))
    assert rc == 0, rc
    return _result(r)


class IcpParams(C.Structure):
    _fields_ = [("max_correspondence_distance", C.c_double), ("max_iterations", C.c_int), ("transformation_epsilon", C.c_double),
                ("euclidean_fitness_epsilon", C.c_double), ("rotation_epsilon", C.c_double), ("num_threads", C.c_int)]


def icp_params(**kw) -> IcpParams:
    p = IcpParams()
    lib().orc_icp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def rigid_from_sums(sums17):
    s = np.ascontiguousarray(sums17, dtype=np.float64).reshape(17)
    T = np.empty(16)
    rc = lib().orc_rigid_from_sums(_p(s, C.c_double), _p(T, C.c_double))
    return T.reshape(4, 4), rc


def icp_align(src, tgt, params: IcpParams | None = None, guess=None):
    """pcl::IterativeClosestPoint::align as the back end configures it (graph_based_slam.cpp:142-151)."""
    src, tgt = _f4(src), _f4(tgt)
    params = params or icp_params()
    g = np.eye(4, dtype=np.float32) if guess is None else np.ascontiguousarray(guess, dtype=np.float32)
    g = g.reshape(16)
    r = Result()
    rc = lib().orc_icp_align(_p(src, C.c_float), C.c_size_t(len(src)), _p(tgt, C.c_float), C.c_size_t(len(tgt)), C.byref(params),
                             _p(g, C.c_float), C.byref(r))
    assert rc == 0, rc
    return _result(r)


class GicpBfgsParams(C.Structure):
    _fields_ = [("max_correspondence_distance", C.c_double), ("max_iterations", C.c_int), ("max_optimizer_iterations", C.c_int),
                ("transformation_epsilon", C.c_double), ("rotation_epsilon", C.c_double), ("k_correspondences", C.c_int),
                ("gicp_epsilon", C.c_double), ("num_threads", C.c_int)]


def gicp_bfgs_params(**kw) -> GicpBfgsParams:
    p = GicpBfgsParams()
    lib().orc_gicp_bfgs_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def gicp_bfgs_align(src, tgt, params: GicpBfgsParams | None = None, guess=None):
    """pclomp::GeneralizedIterativeClosestPoint::align ("GICP", lidar_scan_matcher.cpp:73-96)."""
    src, tgt = _f4(src), _f4(tgt)
    params = params or gicp_bfgs_params()
    g = np.eye(4, dtype=np.float32) if guess is None else np.ascontiguousarray(guess, dtype=np.float32)
    g = g.reshape(16)
    r = Result()
    rc = lib().orc_gicp_bfgs_align(_p(src, C.c_float), C.c_size_t(len(src)), _p(tgt, C.c_float), C.c_size_t(len(tgt)), C.byref(params),
                                   _p(g, C.c_float), C.byref(r))
    assert rc == 0, rc
    return _result(r)


def gicp_bfgs_cost(src, tgt, isrc, itgt, mahal_by_src, base, x6):
    """(f, gradient[6]) of the inner BFGS problem for given correspondences."""
    src, tgt = _f4(src), _f4(tgt)
    isrc = np.ascontiguousarray(isrc, dtype=np.int32)
    itgt = np.ascontiguousarray(itgt, dtype=np.int32)
    M = np.ascontiguousarray(mahal_by_src, dtype=np.float64).reshape(len(src), 9)
    b = np.ascontiguousarray(base, dtype=np.float32).reshape(16)
    x = np.ascontiguousarray(x6, dtype=np.float64).reshape(6)
    f = C.c_double(0)
    g = np.empty(6)
    rc = lib().orc_gicp_bfgs_cost(_p(src, C.c_float), _p(tgt, C.c_float), _p(isrc, C.c_int), _p(itgt, C.c_int), C.c_int(len(isrc)),
                                  _p(M, C.c_double), C.c_size_t(len(src)), _p(b, C.c_float), _p(x, C.c_double), C.byref(f), _p(g, C.c_double))
    assert rc == 0, rc
    return f.value, g
