#!/bin/bash
# Runs ON THE GPU BOX: benchmarks/ndt_batch.py with the current library and every lidar_graph_slam_amd/variants/ndt_*.so (built by
# benchmarks/build_variant.sh <name> ndt -D...), optionally under MI355_NDT_WAVES settings: one line per arm.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_ndt
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[1], "ERR", e); sys.exit(0)
k = {b["kernel"].split("(")[0]: b for b in d["by_kernel"]}
print("%-28s %8.1f pairs/s %7.2f ms/step | deriv own %6.2f ms  advance %5.2f  leaves %5.2f  tsort %5.2f | config3 align %s" % (
    sys.argv[1], d["value"], d["ms_per_step"], k["k_ndt_derivatives"]["ms_per_step_own"], k.get("k_ndt_advance", {}).get("ms_per_step_own", 0),
    k.get("k_ndt_leaves", {}).get("ms_per_step_own", 0), k.get("ndt_target_sort", {}).get("ms_per_step_own", 0),
    (d.get("config3") or {}).get("align_ms(median,min)")))
PY
}
for f in main lidar_graph_slam_amd/variants/ndt_*.so; do
  name=$(basename $f .so)
  if [ "$f" = main ]; then unset MI355_LIB; else export MI355_LIB=$PWD/$f; fi
  for w in ${NDT_WAVES_LIST:-65536}; do
    MI355_NDT_WAVES=$w timeout 600 python benchmarks/ndt_batch.py --no-oracle --steps 6 --warmup 2 ${NDT_AB_ARGS:-} > gpurun_out/ab_ndt/$name.$w.json 2> gpurun_out/ab_ndt/$name.$w.err
    show "$name waves=$w" gpurun_out/ab_ndt/$name.$w.json
  done
done
