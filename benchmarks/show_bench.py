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
    iso = v.get("isolated_avg_launch_ms")
    own = f'{iso * v["launches_per_step"]:7.3f}' if iso is not None else "      -"
    print(f'  {k[:44]:44s} own {own} ms/step (n={v["launches_per_step"]:5.1f} x {iso if iso is not None else float("nan"):8.4f})  frac_own={v.get("isolated_frac_of_hbm_peak")}  '
          f'| timed region {v["total_ms_per_step"]:8.3f} ms/step avg={v["avg_launch_ms"]:8.4f}')
if d.get("h2d_inclusive"):
    print("h2d_inclusive", d["h2d_inclusive"]["value"], "pairs/s", d["h2d_inclusive"]["ms_per_step"], "ms/step")
