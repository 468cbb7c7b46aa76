"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mi355_lidar.h declares,
mirrors the oracle's parameter structs, and fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

from lidar_graph_slam_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name="mi355_lidar.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    syms = _header_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(L.EXPORTS) == syms  # the Python binding lists exactly the header's entry points
    sm = [s for s in _header_symbols("mi355_scan_matcher.h") if s.startswith("mi355_scan_matcher")]
    assert sorted(L.EXPORTS_SCAN_MATCHER) == sm and all(hasattr(lib, s) for s in sm)
    lc = [s for s in _header_symbols("mi355_loop_closure.h") if s.startswith("mi355_loop_closure")]
    assert sorted(L.EXPORTS_LOOP_CLOSURE) == lc and all(hasattr(lib, s) for s in lc)


def test_default_params_match_reference_yaml():
    g = L.gicp_params()
    # lidar_scan_matcher.param.yaml:9-11,18-19 and fast_gicp defaults the node does not override
    assert (g.k_correspondences, g.max_iterations, g.transformation_epsilon, g.max_correspondence_distance) == (20, 64, 0.01, 2.0)
    assert (g.rotation_epsilon, g.regularization_method, g.optimizer, g.lm_max_iterations, g.lm_init_lambda_factor) == (2e-3, L.REG_PLANE, L.OPT_LM, 10, 1e-9)
    n = L.ndt_params()
    assert (n.transformation_epsilon, n.step_size, n.resolution, n.max_iterations, n.min_points_per_voxel) == (0.01, 0.1, 2.0, 64, 6)
    p = L.prefilter_params()
    # points_prefiltering.launch.xml:2-5,12
    assert (p.leaf_size, p.mean_k, p.stddev_mul, p.min_distance) == (0.1, 30, 1.2, 1.0)


def test_param_structs_mirror_the_oracle():
    from oracle import oracle_py as O

    assert [f[0] for f in L.GicpParams._fields_] == [f[0] for f in O.GicpParams._fields_]
    assert [f[0] for f in L.NdtParams._fields_] == [f[0] for f in O.NdtParams._fields_]
    assert C.sizeof(L.GicpParams) == C.sizeof(O.GicpParams) and C.sizeof(L.NdtParams) == C.sizeof(O.NdtParams)


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lidar_graph_slam_amd.context import Context

    with pytest.raises(L.Mi355Error):
        Context(0)
    assert L.lib().mi355_create(0) is None
    # NULL handles are rejected, never dereferenced
    assert L.lib().mi355_synchronize(None) == L.EINVAL
    assert L.lib().mi355_reg_create(None, 0, None) is None


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lidar_graph_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_py" not in text and "liboracle" not in text and "#include \"../oracle" not in text, f
