#!/bin/bash
# Runs ON THE GPU BOX: WRITE_SIZE / FETCH_SIZE per launch of the kernels matching a pattern (separate passes, --kernel-trace only).
# Usage: gpurun -- 'bash benchmarks/pmc_write.sh <kernel-substring>'
PAT=${1:-k_gather_insert}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcw
rm -rf "$O"; mkdir -p "$O"
for C in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/$C" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$O/$C.err"
done
python - "$O" "$PAT" <<'PY'
import csv,sys,glob,collections
O,pat=sys.argv[1],sys.argv[2]
for f in sorted(glob.glob(O+"/*/*/*_counter_collection.csv")):
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            a=acc[(r["Kernel_Name"].split("(")[0][-30:], r["Counter_Name"])]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,(n,v) in acc.items(): print(k, "launches", n, "KB per launch %.0f" % (v/n))
PY
