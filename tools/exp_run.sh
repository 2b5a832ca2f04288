#!/bin/bash
# run the evaluation probe once per variant library, printing the filter kernel's mean time
for v in "$@"; do
  export RECOMMEND_AMD_LIB=$GRAFT_REPO_ROOT/tools/_variants/lib_$v.so
  rm -rf /tmp/prof_$v
  ( cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $GRAFT_REPO_ROOT/tools/eval_probe.py --reps 3 $EXP_ARGS > $GRAFT_REPO_ROOT/gpurun_out/exp_$v.log 2>&1 ) || exit 1
  echo "== $v"; python $GRAFT_REPO_ROOT/tools/kstats.py $(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1) | head -2
done
