#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_filters.py tests/test_golden.py tests/test_gpu_device_clouds.py -m gpu -q 2>&1 | tail -12
timeout 600 python bench.py --sor --steps 4 --warmup 1 --cpu-seconds 6 > /tmp/sor.json 2>/tmp/sor.err; tail -3 /tmp/sor.err; python benchmarks/show_bench.py /tmp/sor.json; cp /tmp/sor.json gpurun_out/r3_bench_sor.json
timeout 300 python benchmarks/other_configs.py --scans 60 2>&1 | tail -30
