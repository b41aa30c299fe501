"""Per-launch-shape kernel times from a rocprofv3 --kernel-trace csv (groups dispatches by kernel name + grid size, so the
layers that share one kernel template are listed separately).  usage: layer_times.py <kernel_trace.csv> [iters]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"][:90], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(a[1] for a in agg.values())
print("total kernel ms per iter: %.2f" % (tot / iters))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.3f ms/iter  %5.1f%%  n=%-4d avg=%.3f  grid=%-10s lds=%-6s %s" % (a[1] / iters, 100 * a[1] / tot, a[0] // iters, a[1] / a[0], k[1], k[2], k[0]))
