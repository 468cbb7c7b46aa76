#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
trace() {
  echo "== $*"
  rm -rf /tmp/itr; env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/itr -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 1 --warmup 1 > /tmp/tr.json 2>/dev/null
  python - <<'PY'
import csv, glob, json
print(open('/tmp/tr.json').read().strip()[:120])
f=sorted(glob.glob('/tmp/itr/*/*_kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for pat in ('k_gicp_correspond<',):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if pat in r['Kernel_Name']]
    print(pat, len(d), 'last 20 launches (us):', ' '.join(f'{x:.0f}' for x in d[-20:]), 'sum', round(sum(d[-20:])/1e3,2),'ms')
PY
}
export MI355_WATCH_TAU=0.001 MI355_WATCH_R0=30
trace MI355_BLOCKS_PER_PAIR=16
trace MI355_BLOCKS_PER_PAIR=8
trace MI355_BLOCKS_PER_PAIR=4
trace MI355_BLOCKS_PER_PAIR=32
