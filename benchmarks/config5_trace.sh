#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of benchmarks/config5.py for one engine (profiles/<round>_kernel_stats_<engine>.csv).
# Usage: bash benchmarks/config5_trace.sh <out-name> <METHOD> [config5 args]
NAME=$1; METHOD=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out "$(dirname gpurun_out/$NAME)" "$(dirname /tmp/c5_$NAME)"; rm -rf /tmp/c5_$NAME
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5_$NAME -- python3 benchmarks/config5.py --method $METHOD "$@" 2> gpurun_out/${NAME}.err | tail -1 > gpurun_out/${NAME}.json
f=$(ls /tmp/c5_$NAME/*/*_kernel_stats.csv | tail -1)
cp "$f" gpurun_out/${NAME}_kernel_stats.csv
python3 benchmarks/kstats.py gpurun_out/${NAME}_kernel_stats.csv 30 | grep -v "at::native" | tee gpurun_out/${NAME}_kstats.txt
# the launches of the engine's search kernel in launch order (us): how the work decays as pairs settle and finish
python3 - "$NAME" <<'PY'
import csv, glob, sys
name = sys.argv[1]
f = sorted(glob.glob(f'/tmp/c5_{name}/*/*_kernel_trace.csv'))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
out = []
for pat in ('k_icp_step', 'k_icp_correspond_tile', 'k_bfgs_correspond', 'k_bfgs_cost', 'k_ndt_derivatives', 'k_gicp_correspond<', 'k_gicp_correspond_tile', 'k_gicp_linearize'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if pat in r['Kernel_Name']]
    if d:
        out.append(f"{pat} {len(d)} launches, us in launch order: " + ' '.join(f'{x:.0f}' for x in d[:120]))
open(f'gpurun_out/{name}_launches.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(o[:1500] for o in out))
PY
cat gpurun_out/${NAME}.json | cut -c1-600
