#!/bin/bash
# run the evaluation probe under each "NAME=VALUE" environment setting, printing the top kernels
for kv in "$@"; do
  rm -rf /tmp/prof_e
  ( export $kv; cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -- python $GRAFT_REPO_ROOT/tools/eval_probe.py --reps 3 $EXP_ARGS > $GRAFT_REPO_ROOT/gpurun_out/exp_env.log 2>&1 ) || exit 1
  echo "== $kv"; grep "rk_score_topk_f32" $GRAFT_REPO_ROOT/gpurun_out/exp_env.log; python $GRAFT_REPO_ROOT/tools/kstats.py $(find /tmp/prof_e -name "*kernel_stats.csv" | head -1) > /tmp/k.txt; grep "true" /tmp/k.txt
done
