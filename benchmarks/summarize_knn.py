#!/usr/bin/env python3
"""gpurun_out/<round>/knn_stats.err (MI355_KNN_STATS=1 python bench.py --pairs 512 ..., collect_evidence.sh) + profiles/<round>_pmc_valu.csv
-> profiles/<round>_knn_tile_counters.txt: candidates per lane and instructions per candidate of k_knn_cov_tile.
usage: python benchmarks/summarize_knn.py r05"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", R, "knn_stats.err")
lines = [l.strip() for l in open(src) if l.startswith("[mi355]")]
blk = lines[-6:]  # the last covariance launch of the run: six lines per launch
valu = None
DST = os.environ.get("EVIDENCE_DST") or os.path.join(ROOT, "profiles")  # (as summarize_evidence.py: on the GPU box a directory under gpurun_out/)
pmc = os.path.join(DST, f"{R}_pmc_valu.csv")
if os.path.exists(pmc):
    rows = list(csv.reader(open(pmc)))
    for r in rows[1:]:
        if r and "k_knn_cov_tile" in r[0]:
            valu = dict(zip(rows[0][1:], r[1:]))
out = ["k_knn_cov_tile<20>: candidates per lane and instructions per candidate (VERDICT r4, item 2)", "",
       "MI355_KNN_STATS=1 python bench.py --pairs 512 --steps 1 --warmup 1 --no-cpu-baseline --no-h2d  (the STATS instantiation of the kernel: the same code",
       "with s_memtime phase clocks and counters; one block of lines per covariance launch, the last launch of the run below)", ""] + blk + ["",
       "Reading: a tile (64 queries, one wave) gathers ~300 candidates into LDS in ~1.8 gathers, files them into four 16-lane sub-lists and every lane",
       "ranks ~185 of them (+ ~12 straight from the tile list in single-sub-tile segments) - 0.6 of the tile's candidates - through the (k + 2)-deep",
       "v_med3 network: 31 VALU instructions per candidate and lane (5 distance, 3 key, 22 + 1 network), ~2.1 rounds per tile.",
       "Shares of a tile's ~133 k shader clocks (the 'gather' of the phase line + the two parts timed apart on the next line): rank ~31 %,",
       "gather ~31 % - the 64 hash probes of its cell lookup alone 12 % (one dependent random 16-byte load per lane, 1.8 times per tile), radii and",
       "boxes 2 %, flat sequence + point loads + compaction 16 % -, pair + covariance ~16 %, fetch ~10 %, sub-list build ~8 %, round end ~3 %.",
       "~0.55 % of the points are handed to k_knn_hard.", ""]
if valu:
    w, wc = float(valu["SQ_WAVES"]), float(valu["SQ_WAVE_CYCLES"])
    out += [f"PMC (profiles/{R}_pmc_valu.csv, 2 048 clouds per launch): SQ_INSTS_VALU {float(valu['SQ_INSTS_VALU']):.4g} per launch over {w:.0f} waves;",
            f"SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {float(valu['SQ_ACTIVE_INST_VALU']) / wc:.3f} per wave (two waves per SIMD: {2 * float(valu['SQ_ACTIVE_INST_VALU']) / wc:.2f} of the SIMD's issue slots),",
            f"SQ_WAIT_ANY / SQ_WAVE_CYCLES = {float(valu['SQ_WAIT_ANY']) / wc:.3f}.  With ~503 k tiles per launch (32.2 M points / 64): {float(valu['SQ_INSTS_VALU']) / (32.2e6 / 64):.0f} VALU instructions per tile,",
            "of which the rank loop is ~197 x 31 = 6.1 k: the kernel is bound by VALU issue, more resident waves could add at most what is left of the slots.", ""]
out += ["What was tried against candidates x instructions (round 5: analysis and the r03 / r04 experiment logs; nothing new kept):",
        "  * a per-lane pre-filter in front of the network (skip when the key cannot enter): the 64 lanes of a wave never agree, the branch is always taken (r04);",
        "  * two-phase ranking with a threshold and a per-lane micro-queue: ~53 of the remaining ~128 candidates still pass a threshold taken from a",
        "    representative sample of 48, the queue drains in lockstep at the pace of its longest lane (~70): (75 x 31 - 128 x 9) / (185 x 31) = 20 % of the",
        "    rank phase = 6 % of the kernel at best, for +6 KB of LDS per wave, i.e. one wave per SIMD less - not built;",
        "  * ring-ordered sub-lists: the rounds ARE rings already (~2.1 per tile: [0, r0] then (r0, need]); finer rings add build passes (8 % each);",
        "  * 8-lane sub-tiles: 10.9 + 1.8 ms against 10.7 + 1.7 ms per 1 024-cloud launch (r03: the trip count of a wave is its longest sub-list);",
        "  * a third wave per SIMD (needs <= 13.3 KB of LDS and <= 168 VGPRs per wave; today 19 KB and 199): r03 measured 8.0 against 8.3 ms with",
        "    576-entry lists but 0.66 - 2.5 % handed over - the same sum.",
        "  * (round 5, built and measured, one box) the cell lookups of the NEXT gather issued ahead - after a gather the first hash probe of every",
        "    cell of the box grown to 1.2 - 2.0 x its radius goes out and comes back under the build + rank; a later round that asks for no more",
        "    finishes the lookups it finds waiting and gathers that box: exact (51 k-NN / covariance parity tests green), 0.37 of the 0.78 later",
        "    gathers per tile found their lookups waiting, the cell-lookup clocks fell 15.5 k -> 13.3 k per tile, but the larger box loads and",
        "    tests more points (the rest of the gather 21.4 k -> 28.3 k): k_knn_cov_tile 50.8 -> 52.3 ms per step at every scale tried - removed.",
        "  * (round 5, built and measured, one box) the cell lookup of the NEXT TILE's first gather issued before the covariance epilogue of the",
        "    current one (same box, no extra points; 0.94 of the tiles found their lookups waiting): the lookup clocks fell 15.9 k -> 11.7 k per",
        "    tile and came back in the phases around them (build + 2.5 k, epilogue + 2.4 k, rank + 2 k: 134.6 k against 134.8 k in all) - with two",
        "    waves filling 0.87 of the SIMD's issue slots a latency that is hidden for one wave is issue time taken from the other; the bench",
        "    line 148.3 / 146.9 ms per step with it against 145.9 / 146.1 without - removed.  The kernel is bound by its instruction count.",
        "  * (round 5, kept) k == K reads the k-th key from its register instead of through a runtime index (K + 2 selects per use, ~150",
        "    instructions per tile): 145.8 -> 145.0 ms per step over four alternations in one box - made AFTER the evidence of this directory was",
        "    collected (0.6 %, inside its box-to-box spread).  Splitting the rank loop into steps all lanes have entries for (no end-of-list",
        "    test: 2 of 31 instructions) and the rest: +0.9 ms; folding k into the covariance epilogue: +29 VGPRs, +-0 - neither kept.",
        "  * (round 6, kept, +-0) the eigenvalue iteration of the covariance epilogue warm-started in fp32 (fewer fp64 divisions): 12.43 -> 12.48 ms per",
        "    2 048-cloud launch - the epilogue's clocks are not in those iterations.  valu_issue_frac of the kernel (the round's SQ counters on one",
        "    scale, profiles/<round>_issue.json): 0.84 of the chip's VALU issue slots at 1.9 resident waves per SIMD.",
        "The product candidates x instructions did not fall in rounds 5 and 6 (DESIGN 8)."]
dst = os.path.join(DST, f"{R}_knn_tile_counters.txt")
open(dst, "w").write("\n".join(out) + "\n")
print("wrote", dst)
