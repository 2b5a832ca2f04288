"""Print a rocprofv3 kernel_stats.csv as name / calls / average microseconds."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:12.1f} us")
