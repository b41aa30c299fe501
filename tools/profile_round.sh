#!/bin/bash
# One round of profiles for the benchmark command of one storage type (run ON the GPU box, from the repo root, via gpurun):
#   tools/profile_round.sh <tag> <dtype>      (the command runs 4 forwards: 1 warm-up, bench.py's untimed profiled step, 2 timed)
# writes under gpurun_out/prof_<tag>/ : kernel trace + stats, three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ busy counters;
# separate passes, program directly after `--`, no trace domains combined with --pmc), and the summaries that get
# committed to profiles/: <tag>_kernel_stats.csv, <tag>_layer_times.txt, <tag>_hbm_traffic.csv, <tag>_sq_busy.txt and
# hbm_traffic.json (stamped with the source-tree hash and the dtype).
set -u
tag=$1; dt=$2
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --dtype $dt --steps 2 --warmup 1 --no-mixed --ppo-envs 0 --no-prepare --no-modes --no-dense-leg --no-accuracy --no-cpu-baseline --no-boundary --no-small-batch --no-peaks"
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $ARGS > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $ARGS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $ARGS > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $O/busy -o b --output-format csv -- python3 $ARGS > $O/busy.log 2>&1
cd $R
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
F=$(find $O/fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/write -name "*counter_collection.csv" | head -1)
B=$(find $O/busy -name "*counter_collection.csv" | head -1)
cp $S $O/${tag}_kernel_stats.csv
python3 tools/layer_times.py $T 4 > $O/${tag}_layer_times.txt
python3 tools/pmc_traffic.py $F $W $O/${tag}_hbm_traffic.csv $O/hbm_traffic.json $dt 4 > $O/traffic.log 2>&1
python3 tools/pmc_busy.py $B $T > $O/${tag}_sq_busy.txt 2>&1
# raw traces are large: keep only the summaries in gpurun_out
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
tail -3 $O/traffic.log; head -12 $O/${tag}_sq_busy.txt
