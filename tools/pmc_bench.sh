#!/bin/bash
# Counter passes over bench.py itself (--pmc-pass: few repetitions, same kernels and inputs as the
# default run) and over the projection probe, one counter set per rocprofv3 run.
# usage (on the GPU box, from the repo root): tools/pmc_bench.sh <tag>
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$PWD}
out=/tmp/pmc_bench_$tag
keep=$root/gpurun_out/pmc_bench_$tag
mkdir -p $out $keep
cd /tmp && export TMPDIR=/tmp
run() {   # run <name> <counters...> -- <program...>
  name=$1; shift
  set_=(); while [ "$1" != "--" ]; do set_+=("$1"); shift; done; shift
  timeout -s KILL 240 rocprofv3 --pmc "${set_[@]}" --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1
  f=$(find $out/$name -name '*counter_collection.csv' | head -1)
  cp $f $out/$name.csv
  echo "$name done"
}
B="python3 $root/bench.py --pmc-pass"
run fetch FETCH_SIZE -- $B
run write WRITE_SIZE -- $B
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
run sq_bench $SQ -- $B
run sq_proj $SQ -- python3 $root/tools/proj_probe.py
run grbm GRBM_GUI_ACTIVE -- $B
grep -h '^{' $out/fetch.log | tail -1 > $out/bench_line.json
# only the summaries travel back (the per-dispatch tables are tens of MB)
python3 $root/tools/pmc_traffic.py $out $keep/r
cp $out/bench_line.json $out/*.log $keep/
