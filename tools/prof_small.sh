#!/bin/bash
# kernel trace of small-batch forwards (the deployment path: B = 1 per env, num_envs: 8): gpurun_out/ps_<tag>_<dtype>_b<B>_layer_times.txt
# usage (on the GPU box, from the repo root): tools/prof_small.sh <tag> [dtype...]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
for dt in "${@:-bf16}"; do
  for B in 1 8; do
    O=$R/gpurun_out/ps_${tag}_${dt}_b$B
    mkdir -p $O
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/tools/run_fwd.py $B 6 $dt > $O/trace.log 2>&1
    cd $R
    T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
    python3 tools/layer_times.py $T 6 > $O/layer_times.txt
    python3 tools/launch_sequence.py $T 6 > $O/sequence.txt 2>/dev/null
    rm -rf $O/trace
  done
done
