"""Median latency of forward + post-processing at small batches (the deployment shape).  usage: small_lat.py [dtype,...] [B,...] [iters]
With RGBM_HIP_LIB set it times that build of the library (tools/small_ab_libs.sh interleaves builds on one box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet, postprocess
dts = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16", "bf16x3"]
Bs = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 60
for dt in dts:
    net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype=dt)
    for B in Bs:
        inp = synth.adapose_inputs(B, seed=0)
        d = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
        def f():
            o = net(d["img1"], d["choose1"], d["img2"], d["choose2"], d["P1"], d["P2"], d["depths"])
            return postprocess(o["view1_nocs"], o["view1_depth"], o["view1_r"], d["choose1"], d["K1"], d["E1"]), o
        for _ in range(5): f()
        torch.cuda.synchronize()
        lat = []
        for _ in range(iters):
            t = time.perf_counter(); r = f(); torch.cuda.synchronize(); lat.append(time.perf_counter() - t)
        chk = float(sum(v.double().abs().sum() for v in r[1].values()))
        print(f"{dt} B={B}: median {np.median(lat) * 1e3:.3f} ms  min {min(lat) * 1e3:.3f}  checksum {chk:.10e}", flush=True)
