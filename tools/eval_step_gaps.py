"""Print, in launch order, the kernels of ONE timed evaluation step of bench.py from a rocprofv3
kernel_trace.csv (from the first row_stats_kernel of a filtered top-K call to the first of the next,
taken among the back-to-back timed steps): start offset, duration, idle gap before it, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "row_stats_kernel" in r["Kernel_Name"]
          and (i == 0 or "row_stats_kernel" not in rows[i - 1]["Kernel_Name"])
          and (i < 2 or "row_stats_kernel" not in rows[i - 2]["Kernel_Name"])]
# back-to-back steps: the shortest distance between successive call starts
best = None
for a, b in zip(starts, starts[1:]):
    d = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    if not any("rank_metrics_kernel" in r["Kernel_Name"] for r in rows[a:b]):
        continue                                   # (the call-only timing loop)
    if best is None or d < best[0]:
        best = (d, a, b)
d, lo, hi = best
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
idle = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    idle += max(0, s - prev_end)
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:100]}")
    prev_end = max(prev_end, e)
print(f"step {d / 1e3:.1f} us, idle {idle / 1e3:.1f} us (incl. the gap to the next step: {(int(rows[hi]['Start_Timestamp']) - prev_end) / 1e3:.1f})")
