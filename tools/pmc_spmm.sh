#!/bin/bash
# PMC passes over the library SpMM hop (tools/_variants/spmm_lab in quick mode), one counter set per run.
# usage: tools/pmc_spmm.sh <tag> [LAB_COLMOD]
set -e
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export LAB_QUICK=1
[ -n "$2" ] && export LAB_COLMOD=$2
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_THREAD_CYCLES_VALU" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_WRITE_sum" ; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- $root/tools/_variants/spmm_lab $root/tools/_variants/elec_csr.bin 5 > $out/p$i.log 2>&1
  f=$(find $out/p$i -name '*counter_collection.csv' | head -1)
  python3 $root/tools/pmc_sum.py $f spmm >> $out/summary.txt
done
cat $out/summary.txt
