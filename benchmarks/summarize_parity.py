"""gpurun_out/parity_deltas.jsonl (written by tests/parity.py during a `pytest -m gpu` run) -> profiles/rNN_parity_deltas.txt:
the worst GPU-vs-oracle pose difference per test family, in the is_converged measure (max |dt_i|, max |dR - I|) and as
Euclidean norm / rotation angle.  BASELINE.md section 4 quotes this file."""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    rows = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", "parity_deltas.jsonl"))]
    groups = collections.OrderedDict()
    other = []
    for r in rows:
        if "band_t" in r:
            groups.setdefault(r["tag"].rsplit("/", 1)[0], []).append(r)
        else:
            other.append(r)
    lines = ["GPU (HIP path through the C-ABI) vs CPU oracle, same inputs: worst pose difference per test family of one `pytest -m gpu` run.",
             "band_t / band_r: max |dt_i| [m] / max |dR - I| of delta = T_gpu * T_oracle^-1 (how fast_gicp's is_converged measures a step;",
             "the convergence band is 0.01 / 2e-3); dt / dr: Euclidean norm [m] / rotation angle [rad].", "",
             f"{'family':44s} {'n':>4s} {'band_t':>10s} {'band_r':>10s} {'dt':>10s} {'dr':>10s}  mode"]
    for k, v in groups.items():
        mode = "fixed iterations" if v[0].get("fixed") else f"early stop, band x {v[0].get('scale', 1.0):g}"
        lines.append(f"{k:44s} {len(v):4d} {max(x['band_t'] for x in v):10.2e} {max(x['band_r'] for x in v):10.2e} {max(x['dt'] for x in v):10.2e} "
                     f"{max(x['dr'] for x in v):10.2e}  {mode}")
    lines.append("")
    for r in other:
        lines.append(json.dumps(r))
    out = os.path.join(ROOT, "profiles", f"{tag}_parity_deltas.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
