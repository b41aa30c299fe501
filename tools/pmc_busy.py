"""MFMA-busy / VALU-busy summary of the top kernels from a rocprofv3 --pmc pass of the SQ counters.
usage: pmc_busy.py <counter_collection.csv> [kernel_trace.csv]
Per kernel (all its dispatches summed):  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * simds_per_se_factor) is not
portable across counter definitions, so the ratios printed here are the ones that need no such factor:
  mfma_busy_per_wave_cycle = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES)   (SQ_WAVE_CYCLES counts quad-cycles, the busy
                             counter cycles; MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units")
  valu_active  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES,   any_active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES,
  waiting      = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier), issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
and the absolute figure that matters for the roofline: matrix-pipe utilisation = MFMA MOPS * 512 flops / (duration * peak)."""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in rows:
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in disp[k]:
        disp[k].add(r["Dispatch_Id"])
        if "End_Timestamp" in r and r["End_Timestamp"]:
            dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
order = sorted(agg, key=lambda k: -dur[k] if dur[k] else -agg[k].get("SQ_WAVE_CYCLES", 0))
print("# rocprofv3 --pmc SQ_* pass; per kernel, all dispatches of the profiled command (3 forwards of batch 256)")
print("%-72s %6s %9s %9s %8s %8s %8s %8s" % ("kernel", "n", "ms(total)", "mfma/wvcy", "valu", "any", "waiting", "stall"))
for k in order[:14]:
    v = agg[k]
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    print("%-72s %6d %9.2f %9.3f %8.3f %8.3f %8.3f %8.3f" % (k.replace("void rgbm::", "")[:72], len(disp[k]), dur[k],
          v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * wc), v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
          v.get("SQ_WAIT_ANY", 0.0) / wc, v.get("SQ_WAIT_INST_ANY", 0.0) / wc))
    mops = v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    busy, sq_busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CYCLES", 0.0)
    if dur[k] > 0:
        print("    MFMA_MOPS_BF16 %.4g  (x512 flop = %.1f TFLOP/s over the kernel's own time)   SQ_VALU_MFMA_BUSY_CYCLES %.4g  SQ_BUSY_CYCLES %.4g" %
              (mops, mops * 512 / (dur[k] * 1e-3) / 1e12, busy, sq_busy))
