"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv> <out.json> <dtype> <forwards>
Bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports KB, and on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes (MI355X_MICROARCH.md, HBM section).  The json is stamped with the hash of the library sources (bench.py::tree_hash)
and the storage type it was measured on; bench.py refuses it for any other tree.  <forwards> = network forwards in the
profiled command (warm-up + timed steps): gives the whole-net bytes per step."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    tot = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(disp[k]), len(disp[k])) for k in tot}


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
dtype, forwards = sys.argv[5], int(sys.argv[6])
rows = []
for k in f:
    fk, n = f[k]
    wk = w.get(k, (0.0, 0))[0]
    rows.append((k, n, fk, wk, (2 * fk + wk) * 1024.0))
rows.sort(key=lambda r: -r[4] * r[1])
with open(sys.argv[3], "w") as fh:
    fh.write("Kernel,Launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,HBM_bytes_per_launch(2*FETCH+WRITE)*1024\n")
    for k, n, fk, wk, b in rows:
        fh.write('"%s",%d,%.1f,%.1f,%.4g\n' % (k[:110], n, fk, wk, b))
from bench import tree_hash  # noqa: E402
total = sum(n * b for k, n, fk, wk, b in rows if "rgbm::" in k)
import socket  # noqa: E402
json.dump({"_meta": {"tree": tree_hash(), "dtype": dtype, "forwards": forwards, "hbm_bytes_per_step": total / forwards,
                     "box": socket.gethostname(), "inputs": "bench.py default (--inputs crop: masks span their crops)",
                     "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, summed over the library's kernels of one forward"},
           "bytes_per_launch": {k: b for k, n, fk, wk, b in rows}}, open(sys.argv[4], "w"), indent=0)
print("wrote", sys.argv[3], sys.argv[4], len(rows), "kernels; whole-net GB per step %.2f" % (total / forwards / 1e9))
