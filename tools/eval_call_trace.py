"""From a rocprofv3 kernel_trace.csv of tools/eval_probe.py: the launches of the LAST
rk_score_topk_filtered_f32 call (from its first row_stats_kernel on), with start offsets."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "row_stats_kernel" in r["Kernel_Name"]]
first = starts[-2]            # a call has two row_stats launches (users, items)
t0 = int(rows[first]["Start_Timestamp"])
end = first
for i in range(first, len(rows)):
    if "at::native" in rows[i]["Kernel_Name"]:
        break
    end = i
for r in rows[first:end + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{d:8.1f} us  grid {r.get('Grid_Size_X', '?'):>9s}  {r['Kernel_Name'][:90]}")
print(f"call span {(int(rows[end]['End_Timestamp']) - t0) / 1e3:.1f} us")
