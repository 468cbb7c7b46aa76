#!/usr/bin/env python3
"""Distil gpurun_out/<round>/ (written by collect_evidence.sh on the GPU box) into the committed files under profiles/.

usage: python benchmarks/summarize_evidence.py r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
PAIRS = 4096
SRC = os.path.join(ROOT, "gpurun_out", R)
DST = os.environ.get("EVIDENCE_DST") or os.path.join(ROOT, "profiles")  # on the GPU box: a small directory under gpurun_out/ (the raw traces stay there)


def last_json(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


def one(pattern):
    # gpurun merges every call's files into the same local directory (rocprofv3 names them by pid): take the newest
    fs = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)
    if not fs:
        raise SystemExit(f"missing {pattern} under {SRC}")
    return fs[-1]


def pmc_per_kernel(counter_dir, counter):
    """mean counter value per launch, per kernel (summed over the per-XCD/SE rows of one dispatch)"""
    per_dispatch = collections.defaultdict(float)
    name = {}
    with open(one(f"{counter_dir}/*/*_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for d, v in per_dispatch.items():
        tot[name[d]] += v
        cnt[name[d]] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


ISSUE_NOTE = ("valu_issue_frac = the share of the chip's VALU issue slots a kernel fills while it runs: SQ_ACTIVE_INST_VALU (summed over the waves, in "
              "quad-cycles) x 4 clocks / (1 024 SIMDs x the kernel's clocks), the clocks taken as SQ_BUSY_CYCLES / 32 (it sums the busy clocks of the 32 shader "
              "engines) = SQ_ACTIVE_INST_VALU / (8 SQ_BUSY_CYCLES).  Checked on k_knn_cov_tile: SQ_WAVE_CYCLES / (8 SQ_BUSY_CYCLES) = 1.9 resident waves per "
              "SIMD where its 19 KB of LDS per wave allow 2.")


def issue_fractions(t):
    busy = max(t.get("SQ_BUSY_CYCLES", 0.0), 1.0)
    return t.get("SQ_WAVE_CYCLES", 0.0) / (8.0 * busy), t.get("SQ_ACTIVE_INST_VALU", 0.0) / (8.0 * busy)


def pmc_sq_table():
    """<round>_pmc_valu.csv: SQ counters per launch and kernel from the two --pmc passes of `python bench.py --steps 1 --warmup 1
    --no-cpu-baseline` (the bench's full batch), plus the derived instructions per point-lane and issue share."""
    names = collections.OrderedDict()
    table = collections.defaultdict(dict)
    launches = {}
    for d in ("pmc_sq1", "pmc_sq2"):
        files = sorted(glob.glob(os.path.join(SRC, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
        if not files:
            return
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        with open(files[-1]) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if "mi355::" not in k:
                    continue
                per[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add(r["Dispatch_Id"])
        for k, v in per.items():
            launches[k] = len(disp[k])
            for c, x in v.items():
                table[k][c] = x / len(disp[k])
                names[c] = 1
    cols = list(names)
    order = sorted(table, key=lambda k: -table[k].get("SQ_BUSY_CYCLES", 0) * launches[k])
    with open(os.path.join(DST, f"{R}_pmc_valu.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow([f"kernel (per-launch averages; python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2d = {PAIRS} pairs per step, {PAIRS // 2} per launch of the "
                    "per-part kernels; two rocprofv3 --pmc passes, --kernel-trace only)", "launches"] + cols + ["VALU_issue_share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES", "WAIT_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
                    "resident_waves_per_SIMD = SQ_WAVE_CYCLES / (8 SQ_BUSY_CYCLES)", "valu_issue_frac = SQ_ACTIVE_INST_VALU / (8 SQ_BUSY_CYCLES)"])
        issue = {}
        for k in order[:24]:
            t = table[k]
            wc = max(t.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            occ, vfrac = issue_fractions(t)
            issue.setdefault(k.split("(")[0].split("<")[0].replace("void ", "").replace("mi355::", "").strip(), {"valu_issue_frac": round(vfrac, 4), "resident_waves_per_simd": round(occ, 3),
                                                                                "valu_issue_share_per_wave": round(t.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
                                                                                "wait_share_per_wave": round(t.get("SQ_WAIT_ANY", 0.0) / wc, 4)})
            w.writerow([k[:90], launches[k]] + [round(t.get(c, 0.0), 1) for c in cols] + [round(t.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4), round(t.get("SQ_WAIT_ANY", 0.0) / wc, 4),
                                                                                         round(occ, 3), round(vfrac, 4)])
    with open(os.path.join(DST, f"{R}_issue.json"), "w") as f:
        json.dump({"pairs": PAIRS, "source": f"profiles/{R}_pmc_valu.csv", "unit": ISSUE_NOTE, "kernels": issue}, f, indent=1)
        f.write("\n")


def main():
    global PAIRS
    os.makedirs(DST, exist_ok=True)
    bench = last_json(os.path.join(SRC, "bench_default.json"))
    pairs = PAIRS = bench["config"]["pairs_per_gpu_per_step"]
    for name in ("bench_default", "bench_hostinputs", "bench_128pairs", "bench_under_trace", "bench_serial_streams", "config5_1gpu", "config5_FAST_GICP", "config5_NDT_OMP",
                 "config5_ICP", "config5_GICP", "bench_sor", "bench_1024pairs", "bench_2048pairs"):
        p = os.path.join(SRC, name + ".json")
        if os.path.exists(p):
            try:
                line = last_json(p)
            except (IndexError, ValueError):
                print("no JSON line in", p, "- that leg failed (see its .err)")
                continue
            with open(os.path.join(DST, f"{R}_{name}.json"), "w") as f:
                f.write(json.dumps(line) + "\n")
    shutil.copy(one("trace/*/*_kernel_stats.csv"), os.path.join(DST, f"{R}_kernel_stats_bench_default.csv"))
    for src_name, dst_name in (("c5trace_ICP_kernel_stats.csv", "kernel_stats_icp.csv"), ("c5trace_GICP_kernel_stats.csv", "kernel_stats_bfgs.csv"),
                               ("c5trace_ICP_launches.txt", "icp_launches.txt"), ("c5trace_GICP_launches.txt", "bfgs_launches.txt"),
                               ("c5trace_ICP.json", "config5_ICP_under_trace.json"), ("c5trace_GICP.json", "config5_GICP_under_trace.json"),
                               ("serial_kernel_stats.csv", "kernel_stats_serial_2048pairs.csv"), ("serial_bench.json", "bench_serial_one_part_under_trace.json")):
        pth = os.path.join(SRC, src_name)
        if os.path.exists(pth):
            shutil.copy(pth, os.path.join(DST, f"{R}_{dst_name}"))
    for txt in ("other_configs", "single_pair", "stream_profile"):
        oc = os.path.join(SRC, txt + ".txt")
        if os.path.exists(oc):
            shutil.copy(oc, os.path.join(DST, f"{R}_{txt}.txt"))
    pmc_sq_table()
    if os.path.exists(os.path.join(SRC, "knn_stats.err")):  # k_knn_cov_tile's own counters beside the PMC instruction counts
        import subprocess
        subprocess.call([sys.executable, os.path.join(ROOT, "benchmarks", "summarize_knn.py"), R])
    for f in ("ndt_batch.json", "ndt_under_trace.json", "kernel_stats_ndt.csv", "pmc_ndt.csv", "ndt_issue.json"):  # benchmarks/ndt_evidence.sh
        pth = os.path.join(SRC, "ndt", f)
        if os.path.exists(pth):
            dst = {"ndt_batch.json": "ndt_batch.json", "ndt_under_trace.json": "ndt_batch_under_trace.json", "kernel_stats_ndt.csv": "kernel_stats_ndt.csv", "pmc_ndt.csv": "pmc_ndt.csv", "ndt_issue.json": "ndt_issue.json"}[f]
            shutil.copy(pth, os.path.join(DST, f"{R}_{dst}"))
    fetch = pmc_per_kernel("pmc_fetch", "FETCH_SIZE")
    write = pmc_per_kernel("pmc_write", "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[0] + write.get(k, (0, 0))[0])):
        f_kb, n = fetch.get(k, (0.0, 0))
        w_kb, _ = write.get(k, (0.0, 0))
        rows.append((k, n, f_kb, w_kb))
    with open(os.path.join(DST, f"{R}_pmc_fetch_write.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch_raw", "WRITE_SIZE_KB_per_launch", "HBM_bytes_per_launch_FETCHx2_plus_WRITE"])
        for k, n, f_kb, w_kb in rows[:40]:
            w.writerow([k, n, round(f_kb, 1), round(w_kb, 1), int(2 * f_kb * 1024 + w_kb * 1024)])
    # the kernel bench.py reports its roofline on (the family with the largest share of GPU time) -> <round>_traffic.json
    dom = bench["roofline"]["kernel"].split("(")[0]
    hit = [r for r in rows if dom in r[0]]
    if hit:
        k, n, f_kb, w_kb = hit[0]
        tj = {
            "kernel": dom,
            "pairs": pairs,
            "hbm_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
            "fetch_size_kb_raw": round(f_kb, 1),
            "write_size_kb": round(w_kb, 1),
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py --steps 2 "
                      "--warmup 1 --no-cpu-baseline`; FETCH_SIZE doubled (gfx950 counts the 128-B requests of 16-B/lane streaming reads as "
                      f"64 B, MI355X_MICROARCH.md §HBM; uncalibrated for the narrow gathers of this kernel); per-launch average over {n} launches; "
                      f"profiles/{R}_pmc_fetch_write.csv",
        }
        with open(os.path.join(DST, f"{R}_traffic.json"), "w") as f:
            json.dump(tj, f, indent=1)
            f.write("\n")
        print("traffic", dom, tj["hbm_bytes_per_launch"], "bytes/launch; algorithmic", bench["roofline"]["algorithmic_bytes_per_launch"])
    print("bench", bench["value"], bench["unit"], "frac", bench["roofline"]["frac"], "cpu", bench.get("cpu_baseline"))


if __name__ == "__main__":
    main()
