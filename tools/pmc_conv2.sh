#!/bin/bash
# memory-path counters of the implicit-GEMM kernels on one conv shape (run on the GPU box): pmc_conv2.sh "<bench_conv args>"
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_conv2; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/p$i -o p --output-format csv -- python3 $R/tools/bench_conv.py $1 > $O/p$i.log 2>&1 || echo "pass $i failed: $grp"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_conv2/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "igemm" not in k: continue
        agg[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, v in agg.items():
    print(k[:90])
    for c, xs in sorted(v.items()):
        xs.sort()
        print("   %-34s %s" % (c, " ".join("%.4g" % x for _, x in xs)))
PY
