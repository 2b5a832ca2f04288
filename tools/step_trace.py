"""Print, in launch order, the kernels of ONE replayed training step from a rocprofv3
kernel_trace.csv of bench.py (the launches between the last two bpr_fwd_kernel launches that are
followed by a whole step): start offset, duration, grid, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "bpr_fwd_kernel" in r["Kernel_Name"]]
if len(marks) < 12:
    sys.exit("fewer than 12 training steps in the trace")
lo, hi = marks[-3], marks[-2]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  grid {r.get('Grid_Size_X', '?'):>9s}  {r['Kernel_Name'][:90]}")
    prev_end = e
