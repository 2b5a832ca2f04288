#!/bin/bash
# Memory-side bytes of the SpMM hop on the scale-up rank slices (tools/scaleup_sizing.py), one
# counter per rocprofv3 run.  usage (GPU box, repo root): tools/pmc_slice.sh
set -e
root=${GRAFT_REPO_ROOT:-$PWD}
out=/tmp/pmc_slice
keep=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for rank in 0 7; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/r${rank}_$c -- python3 $root/tools/scaleup_sizing.py --spmm-only --rank=$rank > $out/r${rank}_$c.log 2>&1
    f=$(find $out/r${rank}_$c -name '*counter_collection.csv' | head -1)
    cp $f $out/r${rank}_$c.csv
  done
done
python3 - $out > $keep/r03_pmc_scaleup_slices.txt <<'PY'
import csv, sys, collections
src = sys.argv[1]
print("SpMM hop on one rank's row block of the 10M x 2M x 500M configuration (D = 128), memory-side traffic per launch:")
print("FETCH_SIZE / WRITE_SIZE in KB, mean over the launches of spmm_rows_kernel; bytes = 1000 * (2 * FETCH_SIZE + WRITE_SIZE)")
print("(gfx950 tallies its 128-B read requests at 64 B)\n")
for rank in (0, 7):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v, n = 0.0, 0
        for r in csv.DictReader(open(f"{src}/r{rank}_{c}.csv")):
            if "spmm_rows_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v += float(r["Counter_Value"]); n += 1
        tot[c] = v / max(n, 1)
        tot["n"] = n
    log = [l.strip() for l in open(f"{src}/r{rank}_FETCH_SIZE.log") if "block of rank" in l or "local SpMM" in l]
    print(f"rank {rank}: {tot['n']} launches, FETCH_SIZE {tot['FETCH_SIZE']:.0f} KB, WRITE_SIZE {tot['WRITE_SIZE']:.0f} KB "
          f"-> {1e3 * (2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) / 1e9:.2f} GB per launch")
    for l in log:
        print("   " + l)
PY
cat $keep/r03_pmc_scaleup_slices.txt
