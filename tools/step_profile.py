"""Per-training-step kernel table from a rocprofv3 kernel_trace.csv of bench.py: the launches
between two consecutive bpr_fwd_kernel launches are one step; prints, per kernel name, launches
per step and microseconds per step (mean over the steps found), and the step's total."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "bpr_fwd_kernel" in r["Kernel_Name"]]
if len(marks) < 12:
    sys.exit("fewer than 12 training steps in the trace")
lo, hi = marks[5], marks[-2]          # skip the eager warm-up steps, keep whole steps
n_steps = marks.index(hi) - 5
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    a = agg[r["Kernel_Name"][:100]]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
total = sum(v[1] for v in agg.values()) / n_steps
wall = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3 / n_steps
print(f"{n_steps} steps; kernel time {total:.1f} us per step, {sum(v[0] for v in agg.values()) / n_steps:.1f} launches per step, "
      f"wall {wall:.1f} us per step")
for name, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{c / n_steps:7.2f} x {us / c:9.1f} us = {us / n_steps:9.1f} us/step  {name}")
