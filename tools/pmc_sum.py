"""Sum rocprofv3 counter_collection.csv per (kernel, counter): mean per dispatch."""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, set()])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:60], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1].add(r["Dispatch_Id"])
for (kn, cn), (v, d) in sorted(acc.items()):
    if len(sys.argv) < 3 or sys.argv[2] in kn:
        print(f"{kn:60s} {cn:28s} {v/len(d):16.0f}  ({len(d)} dispatches)")
