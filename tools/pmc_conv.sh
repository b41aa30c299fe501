#!/bin/bash
# SQ counters of the implicit-GEMM kernels on one conv shape (run on the GPU box): pmc_conv.sh "<bench_conv args>"
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_conv; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/a -o a --output-format csv -- python3 $R/tools/bench_conv.py $1 > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT -d $O/b -o b --output-format csv -- python3 $R/tools/bench_conv.py $1 > $O/b.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_conv/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "igemm" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add((f, r["Dispatch_Id"]))
for k, v in agg.items():
    nd = len(n[k]) / 2
    print(k[:90], "dispatches/pass", nd)
    for c, x in sorted(v.items()): print("   %-28s %.4g per dispatch" % (c, x / nd))
PY
tail -3 $O/a.log $O/b.log | grep -v amdgpu
