"""Every dispatch of the LAST of `iters` iterations of a rocprofv3 --kernel-trace csv, in start order with its duration (the
persistent implicit-GEMM launches all share one grid size, so tools/layer_times.py cannot tell the layers apart).
usage: launch_sequence.py <kernel_trace.csv> [iters]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // iters
last = rows[len(rows) - n:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0.0
for i, r in enumerate(last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot += d
    print("%3d  t=%8.3f  %7.3f ms  grid=%-10s %s" % (i, (int(r["Start_Timestamp"]) - t0) / 1e6, d, r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Kernel_Name"][:110]))
print("sum %.3f ms, span %.3f ms" % (tot, (int(last[-1]["End_Timestamp"]) - t0) / 1e6))
