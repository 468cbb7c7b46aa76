#!/usr/bin/env python3
"""Print the per-kernel table of a bench.py JSON line."""
import json
import sys

try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    print("no bench json:", e)
    sys.exit(0)
print("pairs/s", d["value"], "ms/step", d["ms_per_step"], "e2e frac", d["roofline"]["end_to_end"]["frac"], "acc", d["accuracy"])
for k, v in d["roofline"]["by_kernel"].items():
    print(f'  {k[:44]:44s} {v["total_ms_per_step"]:8.3f} ms/step  n={v["launches_per_step"]:5.1f} avg={v["avg_launch_ms"]:8.4f} frac={v.get("frac_of_hbm_peak")}')
