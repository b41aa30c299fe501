"""Same-process A/B of rgbm_debug_flags words at small batches: median latency of forward + post-processing, the values interleaved.
usage: small_flag_ab.py <dtypes> <Bs> <flag,flag,...> [iters]      e.g.  small_flag_ab.py bf16,bf16x3 1,2,8 0,16384"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbmanip_amd import synth, _lib
from rgbmanip_amd.adapose import AdaPoseNet, postprocess
lib = _lib.load()
dts = sys.argv[1].split(",")
Bs = [int(b) for b in sys.argv[2].split(",")]
flags = [int(f) for f in sys.argv[3].split(",")]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 40
for dt in dts:
    net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype=dt)
    for B in Bs:
        inp = synth.adapose_inputs(B, seed=0)
        d = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
        def f():
            o = net(d["img1"], d["choose1"], d["img2"], d["choose2"], d["P1"], d["P2"], d["depths"])
            postprocess(o["view1_nocs"], o["view1_depth"], o["view1_r"], d["choose1"], d["K1"], d["E1"])
            return o
        lat = {fl: [] for fl in flags}; outs = {}
        for rnd in range(3):
            for fl in flags:
                lib.rgbm_debug_flags(fl)
                for _ in range(3): f()
                torch.cuda.synchronize(); l = []
                for _ in range(iters):
                    t = time.perf_counter(); o = f(); torch.cuda.synchronize(); l.append(time.perf_counter() - t)
                lat[fl].append(float(np.median(l)) * 1e3)
                o = {k: v.clone() for k, v in o.items()}
                if fl in outs:
                    assert all(bool((outs[fl][k] == o[k]).all()) for k in o), ("run-to-run", dt, B, fl)
                outs[fl] = o
        lib.rgbm_debug_flags(0)
        ref = outs[flags[0]]
        line = f"{dt} B={B}:"
        for fl in flags:
            diff = max(float((outs[fl][k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ref)
            line += f"  flag {fl}: {min(lat[fl]):.3f} ms (rounds {[round(x, 3) for x in lat[fl]]}) max rel diff to flag {flags[0]} {diff:.1e};"
        print(line, flush=True)
