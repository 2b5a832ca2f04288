#!/bin/bash
# Counter passes over one filtered top-K call at a named shape (tools/eval_probe.py), one counter
# set per rocprofv3 run; prints the per-kernel means of the filter's kernels.
# usage (on the GPU box, from the repo root): tools/pmc_eval.sh <shape>
shape=${1:-elec}
root=${GRAFT_REPO_ROOT:-$PWD}
out=/tmp/pmc_eval_$shape
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {   # run <name> <counters...>
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $root/tools/eval_probe.py --shape $shape --reps 1 > $out/$name.log 2>&1
  f=$(find $out/$name -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "$name: no counters"; tail -3 $out/$name.log; return; fi
  python3 $root/tools/pmc_sum.py $f | grep "strata_max\|collect_kernel\|score_topk_kernel<28\|refine" 
}
run sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_VALU_MFMA_COEXEC_CYCLES
