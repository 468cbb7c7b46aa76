"""Print the top rows of a rocprofv3 *_kernel_stats.csv (name, calls, average / max duration, share)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 22
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", round(tot / 1e6, 3), "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:top]:
    print(f"{r['Name'][:50]:50s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs']) / 1e3:10.1f} max_us={float(r['MaxNs']) / 1e3:10.1f} "
          f"share={100 * float(r['TotalDurationNs']) / tot:5.1f}%")
