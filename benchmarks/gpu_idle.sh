#!/bin/bash
# Runs ON THE GPU BOX: how much of a bench step has NO kernel running (host gaps), from a rocprofv3 kernel trace of a short run:
# union of all kernel intervals over the last two steps against their wall time, and the ten longest gaps with the kernels around them.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/idl; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/idl -- python3 bench.py --no-cpu-baseline --no-h2d --no-kernel-timing --steps 3 --warmup 2 $GPU_IDLE_ARGS > /tmp/idl.json 2>/dev/null
python3 - <<'PY'
import csv, glob, json
f = sorted(glob.glob('/tmp/idl/*/*_kernel_trace.csv'))[-1]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in csv.DictReader(open(f))]
rows.sort()
line = json.loads([l for l in open('/tmp/idl.json') if l.startswith('{')][-1])
step_ns = line['ms_per_step'] * 1e6
lm = [r for r in rows if 'k_gicp_step' in r[2]]
t1 = lm[-1][1]
t0 = t1 - 2 * step_ns
win = [r for r in rows if r[1] > t0 and r[0] < t1]
busy, cur_s, cur_e, gaps, last_n = 0, None, None, [], ''
for s, e, n in win:
    s, e = max(s, t0), min(e, t1)
    if cur_e is None: cur_s, cur_e, last_n = s, e, n
    elif s <= cur_e:
        if e > cur_e: cur_e, last_n = e, n
    else:
        gaps.append((s - cur_e, last_n, n)); busy += cur_e - cur_s; cur_s, cur_e, last_n = s, e, n
busy += cur_e - cur_s
print(f"bench line: {line['value']:.0f} pairs/s, {line['ms_per_step']:.1f} ms/step; window of two steps: busy {busy / (t1 - t0) * 100:.1f} % ({(t1 - t0 - busy) / 2e6:.2f} ms idle per step), {len(gaps)} gaps")
for g in sorted(gaps, reverse=True)[:10]: print(f"  gap {g[0] / 1e3:8.1f} us  after {g[1]}  before {g[2]}")
PY
