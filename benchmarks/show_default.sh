#!/bin/bash
# Runs ON THE GPU BOX: the default bench line, with its live and isolated per-kernel numbers side by side.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python bench.py "$@" > gpurun_out/bench_show.json 2> gpurun_out/bench_show.err; echo "bench exit $?"; tail -2 gpurun_out/bench_show.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_show.json").read().strip().splitlines()[-1])
print(d["value"], "pairs/s", d["ms_per_step"], "ms/step; pairs", d["config"]["pairs_per_gpu_per_step"], "in flight", d["config"].get("batches_in_flight"))
r = d["roofline"]
print("roofline:", r["kernel"], "frac", r["frac"], "avg ms", r["avg_launch_ms"], "isolated", r.get("isolated"), "e2e", r["end_to_end"]["frac"])
for k, v in r["by_kernel"].items():
    print("  %-40s live %8.4f ms  isolated %s  frac %s / %s" % (k[:40], v["avg_launch_ms"], v.get("isolated_avg_launch_ms"), v.get("frac_of_hbm_peak"), v.get("isolated_frac_of_hbm_peak")))
print(d.get("cpu_baseline") and d["cpu_baseline"].get("pairs_per_s_by_threads"), d["accuracy"])
PY
