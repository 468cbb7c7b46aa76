#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_properties.py tests/test_golden.py tests/test_gpu_multi.py tests/test_gpu_loop_closure.py tests/test_gpu_scan_matcher.py tests/test_gpu_device_clouds.py tests/test_gpu_gicp_bfgs.py -m gpu -q 2>&1 | tail -12
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /tmp/b.json 2>/tmp/b.err; tail -2 /tmp/b.err; python benchmarks/show_bench.py /tmp/b.json
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --pairs 128 > /tmp/b.json 2>/tmp/b.err; python benchmarks/show_bench.py /tmp/b.json | head -3
python benchmarks/single_pair.py
python benchmarks/stream_profile.py 120 FAST_GICP
