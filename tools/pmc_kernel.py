"""Aggregate a rocprofv3 --pmc counter_collection csv per kernel.  usage: pmc_kernel.py <csv> [name-substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float); seen = set()
for r in rows:
    k = r["Kernel_Name"][:80]
    if sub not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:8]:
    print(k, "launches", n[k], "total ms %.3f" % (dur[k] / 1e6))
    for c, x in sorted(v.items()): print("    %-36s %.4g   per launch %.4g" % (c, x, x / n[k]))
