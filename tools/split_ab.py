"""One-stream forward against AdaPoseNet(split_streams=True) (two half batches on two streams) on the bench's inputs: bit-identity of all ten
outputs and interleaved timing of forward + post-processing.  usage: split_ab.py [dtype] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet, postprocess

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
_, d, _ = bench.make_inputs_crop(B, dev, seed=0)
sd = synth.adapose_state_dict(seed=0)
nets = {"one": AdaPoseNet(sd, dtype=dt), "split": AdaPoseNet(sd, dtype=dt, split_streams=True, split_min_batch=2)}
def step(n):
    out = n(d["img1"], d["choose1"], d["img2"], d["choose2"], d["P1"], d["P2"], d["depths"])
    return out, postprocess(out["view1_nocs"], out["view1_depth"], out["view1_r"], d["choose1"], d["K1"], d["E1"])
outs = {}
for k, n in nets.items():
    o, pp = step(n)
    torch.cuda.synchronize()
    outs[k] = {kk: v.cpu().numpy() for kk, v in o.items()}
bad = [k for k in outs["one"] if not np.array_equal(outs["one"][k].view(np.uint8), outs["split"][k].view(np.uint8))]
print(f"{dt} B={B}: outputs bit-identical: {not bad} {bad}")
for rnd in range(3):
    for k, n in nets.items():
        for _ in range(2): step(n)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(8): step(n)
        torch.cuda.synchronize()
        print(f"{dt} B={B} {k:5s}: {(time.perf_counter() - t) / 8 * 1e3:.2f} ms per step")
