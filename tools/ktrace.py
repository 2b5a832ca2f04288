"""Print a rocprofv3 kernel_trace.csv in launch order: name, grid, duration (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-int(sys.argv[2]) if len(sys.argv) > 2 else 0:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:12.1f} {d:10.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>10s}  {r['Kernel_Name'][:80]}")
