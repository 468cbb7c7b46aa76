#!/bin/bash
# Runs ON THE GPU BOX: duration of every launch of the LM-loop kernels of ONE bench step, in launch order (rocprofv3 kernel trace).
# With the default two parts the launches of the parts interleave in time: MI355_BATCH_PARTS=1 ITER_TRACE_ARGS="--pairs 2048" gives iterations 0 .. 19 in order.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/itr; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/itr -- python3 bench.py --no-cpu-baseline --no-h2d --bases 8 --no-kernel-timing --steps 1 --warmup 1 $ITER_TRACE_ARGS > /dev/null 2>&1
python - <<'PY'
import csv, glob
f=sorted(glob.glob('/tmp/itr/*/*_kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for pat in ('k_gicp_correspond<','k_gicp_correspond_tile','k_gicp_linearize','k_gicp_research','k_gicp_step'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if pat in r['Kernel_Name']]
    print(pat, len(d), 'last 20 launches (us):', ' '.join(f'{x:.0f}' for x in d[-20:]))
PY
