"""Run a few bf16 forwards at B poses (for profiler runs).  usage: run_fwd.py [B] [iters] [dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet, postprocess
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype=dtype)
inp = synth.adapose_inputs(min(B, 8), seed=0)
inp = {k: torch.from_numpy(np.concatenate([v] * ((B + 7) // 8), 0)[:B]).cuda() for k, v in inp.items()}
for _ in range(iters):
    out = net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"])
    postprocess(out["view1_nocs"], out["view1_depth"], out["view1_r"], inp["choose1"], inp["K1"], inp["E1"])
torch.cuda.synchronize()
print("done")
