#!/bin/bash
# Runs ON THE GPU BOX: cache-side PMC counters (TCC hits / misses, TCP requests and stalls, TA busy) of the LM and k-NN kernels, one rocprofv3 pass per group.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for G in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/pm -- python3 bench.py --no-cpu-baseline --no-h2d --steps 1 --warmup 1 --pairs 1024 --bases 8 > /dev/null 2> /tmp/pm.err || tail -3 /tmp/pm.err
  python3 - <<'PY'
import csv,glob,collections
fs=glob.glob('/tmp/pm/*/*_counter_collection.csv')
if not fs: print('no counters (bad name?)'); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name']
    if 'linearize' not in k and 'k_gicp_correspond<' not in k and 'knn_cov_tile' not in k: continue
    k=k.split('(')[0][-30:]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k,v in acc.items():
    print(k, len(n[k]), {c: round(x/len(n[k])) for c,x in v.items()})
PY
done
