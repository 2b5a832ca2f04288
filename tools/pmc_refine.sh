#!/bin/bash
# Counter passes over the refine kernels of one filtered top-K call (tools/eval_probe.py with the
# given arguments), one counter set per rocprofv3 run.
# usage (on the GPU box, from the repo root): tools/pmc_refine.sh --shape elec --items 7875
root=${GRAFT_REPO_ROOT:-$PWD}
out=/tmp/pmc_refine
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_INSTS_LDS" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $root/tools/eval_probe.py --reps 1 "$@" > $out/p$i.log 2>&1
  f=$(find $out/p$i -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "pass $i: no counters"; tail -3 $out/p$i.log; continue; fi
  python3 $root/tools/pmc_sum.py $f refine
done
