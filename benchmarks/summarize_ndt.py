#!/usr/bin/env python3
"""Distil the raw rocprofv3 output of benchmarks/ndt_evidence.sh (one directory) into small files beside it:
  kernel_stats_ndt.csv   the --kernel-trace --stats table of `python benchmarks/ndt_batch.py --steps 5 --warmup 2 --config3`
  pmc_ndt.csv            per kernel and launch: FETCH_SIZE / WRITE_SIZE (separate passes) and the SQ counters (two passes)
usage: python benchmarks/summarize_ndt.py gpurun_out/r05/ndt"""
import collections
import csv
import glob
import os
import shutil
import sys

SRC = sys.argv[1]


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)
    return fs[-1] if fs else None


def per_kernel(counter_dir):
    """{kernel: {counter: mean per launch}}, {kernel: launches}"""
    f = newest(f"{counter_dir}/*/*_counter_collection.csv")
    if not f:
        return {}, {}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if "mi355::" not in k:
                continue
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(disp[k]) for c, v in cs.items()} for k, cs in per.items()}, {k: len(d) for k, d in disp.items()}


def main():
    ks = newest("trace/*/*_kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(SRC, "kernel_stats_ndt.csv"))
        with open(ks) as f:
            for i, ln in enumerate(f):
                if i < 16:
                    print(ln.rstrip()[:200])
    table = collections.defaultdict(dict)
    launches = {}
    for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
        t, n = per_kernel(d)
        for k, cs in t.items():
            table[k].update(cs)
            launches.setdefault(k, n[k])
    if not table:
        return
    cols = sorted({c for cs in table.values() for c in cs})
    order = sorted(table, key=lambda k: -table[k].get("SQ_BUSY_CYCLES", 0.0) * launches[k])
    with open(os.path.join(SRC, "pmc_ndt.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel (per-launch means; python benchmarks/ndt_batch.py --steps 1|2 --warmup 1 --no-oracle; separate rocprofv3 --pmc passes, --kernel-trace only; "
                    "FETCH_SIZE / WRITE_SIZE in KB, FETCH raw: x2 for 16-B/lane streaming reads)", "launches"] + cols +
                   ["HBM_bytes_FETCHx2_plus_WRITE", "VALU_issue_share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES", "WAIT_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
                    "resident_waves_per_SIMD = SQ_WAVE_CYCLES / (8 SQ_BUSY_CYCLES)", "valu_issue_frac = SQ_ACTIVE_INST_VALU / (8 SQ_BUSY_CYCLES)"])
        issue = {}
        for k in order[:30]:
            t = table[k]
            wc = max(t.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            busy = max(t.get("SQ_BUSY_CYCLES", 0.0), 1.0)  # (sums the busy clocks of the 32 shader engines; benchmarks/summarize_evidence.py ISSUE_NOTE)
            occ, vfrac = t.get("SQ_WAVE_CYCLES", 0.0) / (8.0 * busy), t.get("SQ_ACTIVE_INST_VALU", 0.0) / (8.0 * busy)
            issue.setdefault(k.split("(")[0].split("<")[0].replace("void ", "").replace("mi355::", "").strip(), {"valu_issue_frac": round(vfrac, 4), "resident_waves_per_simd": round(occ, 3),
                                                                                "hbm_bytes_per_launch_FETCHx2_plus_WRITE": int(2048 * t.get("FETCH_SIZE", 0.0) + 1024 * t.get("WRITE_SIZE", 0.0))})
            w.writerow([k[:100], launches[k]] + [round(t.get(c, 0.0), 1) for c in cols] +
                       [int(2048 * t.get("FETCH_SIZE", 0.0) + 1024 * t.get("WRITE_SIZE", 0.0)), round(t.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
                        round(t.get("SQ_WAIT_ANY", 0.0) / wc, 4), round(occ, 3), round(vfrac, 4)])
    import json
    with open(os.path.join(SRC, "ndt_issue.json"), "w") as f:
        json.dump({"source": "pmc_ndt.csv of the same directory", "kernels": issue}, f, indent=1)
        f.write("\n")
    print(open(os.path.join(SRC, "pmc_ndt.csv")).read()[:6000])


if __name__ == "__main__":
    main()
