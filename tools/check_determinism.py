"""Bit-stability of the bf16 forward (all outputs) over repeated runs at a given batch size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.adapose_inputs(B, seed=0).items()}
net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype=dtype, device=0)
ref = None
bad = 0
for r in range(reps):
    o = net(inp["img1"], inp["choose1"].int(), inp["img2"], inp["choose2"].int(), inp["P1"], inp["P2"], inp["depths"])
    cur = {k: v.clone() for k, v in o.items()}
    if ref is None:
        ref = cur
    else:
        for k in ref:
            if not torch.equal(ref[k].view(torch.uint8), cur[k].view(torch.uint8)):
                bad += 1
                print("run", r, k, "differs: max", (ref[k].float() - cur[k].float()).abs().max().item())
print(dtype, "B", B, "runs", reps, "differing outputs", bad)
assert bad == 0
