#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== two ranks on device 0, nccl"
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 benchmarks/rccl_probe.py 2>&1 | tail -8
echo "== one rank, nccl"
timeout 120 python benchmarks/rccl_probe.py 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /tmp/b.json 2>/tmp/b.err; python benchmarks/show_bench.py /tmp/b.json
python benchmarks/single_pair.py
