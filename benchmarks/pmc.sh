#!/bin/bash
# Runs ON THE GPU BOX: per-kernel PMC counters (one rocprofv3 pass per counter group; --kernel-trace only, as the pool requires).
# Usage: gpurun -- 'bash benchmarks/pmc.sh <tag> <kernel-substring> [bench args]'
TAG=${1:-pmc}; PAT=${2:-k_knn}; shift; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
ARGS="--no-cpu-baseline --steps 2 --warmup 1 --pairs 128 $@"
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
i=0
for G in "$G1" "$G2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$O/g$i" -- python bench.py $ARGS > /dev/null 2> "$O/g$i.err"
done
python - "$O" "$PAT" <<'PY'
import csv,sys,glob,collections
O,pat=sys.argv[1],sys.argv[2]
for g in sorted(glob.glob(O+"/g*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(g)):
        k=r['Kernel_Name']
        if pat not in k: continue
        k=k.split('(')[0][-40:]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        key=(k,r['Dispatch_Id'])
        if key not in seen: seen.add(key); n[k]+=1
    for k,v in acc.items():
        print(k, 'dispatches', n[k])
        for c,x in sorted(v.items()): print(f"   {c:24s} {x/n[k]:16.0f} per dispatch")
PY
