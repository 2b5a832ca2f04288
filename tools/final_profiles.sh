# The runs behind the numbers in DESIGN.md (GPU box, repo root): the default benchmark line, the same
# command under rocprofv3 --kernel-trace --stats, the kernels of one training step and of one top-K
# call, the other shapes at their presets.  Results under gpurun_out/fin; the judged copies go to profiles/.
set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/fin
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --epoch-steps 400 --scaleup-slice 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
S=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); T=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
cp $S $O/kernel_stats.csv
python3 tools/step_profile.py $T > $O/train_step_kernel_table.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ev -- python3 tools/eval_probe.py > $O/eval_probe.log 2>&1
T2=$(find /tmp/ev -name '*kernel_trace.csv' | head -1)
python3 tools/eval_call_trace.py $T2 > $O/eval_call_trace.txt
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err
for shape in baby sports clothing; do
  timeout -k 10 300 python3 bench.py --shape $shape > $O/bench_$shape.json 2> $O/bench_$shape.err
done
