"""Forward + post-processing at batch B as n equal parts on n streams (AdaPoseNet(split_streams=n)) against the one-stream forward.
usage: split_parts_ab.py [dtype] [B] [parts,...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet, postprocess
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
parts = [int(p) for p in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
inp = synth.adapose_inputs(8, seed=0)
d = {k: torch.from_numpy(np.concatenate([v] * (B // 8), 0)).cuda() for k, v in inp.items()}
sd = synth.adapose_state_dict(seed=0)
nets = {p: AdaPoseNet(sd, dtype=dt, split_streams=(p if p > 1 else False), split_min_batch=1) for p in parts}
def step(p):
    o = nets[p](d["img1"], d["choose1"], d["img2"], d["choose2"], d["P1"], d["P2"], d["depths"])
    return o, postprocess(o["view1_nocs"], o["view1_depth"], o["view1_r"], d["choose1"], d["K1"], d["E1"])
outs = {}
for p in parts:
    outs[p] = {k: v.clone() for k, v in step(p)[0].items()}; torch.cuda.synchronize()
for p in parts[1:]:
    nd = {k: int((outs[p][k] != outs[parts[0]][k]).sum()) for k in outs[p]}
    print(f"{dt} parts {p}: elements differing from parts {parts[0]}: {sum(nd.values())}")
for rnd in range(3):
    for p in parts:
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): step(p)
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t) / 3
        print(f"round {rnd} {dt} B={B} parts={p}: {dtm * 1e3:.2f} ms  {B / dtm:.0f} poses/s", flush=True)
