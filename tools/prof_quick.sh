#!/bin/bash
# quick kernel-trace of both dtypes: gpurun_out/<tag>_{bf16,bx3}_{layer_times,sequence}.txt
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
for dt in bf16 bf16x3; do
  O=$R/gpurun_out/pq_${tag}_$dt
  mkdir -p $O
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py --dtype $dt --steps 2 --warmup 1 --no-mixed --ppo-envs 0 --no-prepare --no-modes --no-dense-leg --no-accuracy --no-cpu-baseline --no-boundary --no-small-batch --no-peaks > $O/trace.log 2>&1
  cd $R
  T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
  cp $S $O/kernel_stats.csv
  python3 tools/layer_times.py $T 4 > $O/layer_times.txt
  python3 tools/launch_sequence.py $T 4 > $O/sequence.txt
  find $O -name "*kernel_trace.csv" -delete
  rm -rf $O/trace
done
