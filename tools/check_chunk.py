"""Outputs of one batch-256 forward with the whole batch in one cost-volume chunk against 128-view chunks (index-width check)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgbmanip_amd import synth
from rgbmanip_amd.adapose import AdaPoseNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.adapose_inputs(B, seed=0).items()}
sd = synth.adapose_state_dict(seed=0)
outs = []
for chunk in (2 * B, 128):
    net = AdaPoseNet(sd, dtype="bf16", device=0, max_chunk_views=chunk)
    o = net(inp["img1"], inp["choose1"].int(), inp["img2"], inp["choose2"].int(), inp["P1"], inp["P2"], inp["depths"])
    outs.append({k: v.float().cpu().clone() for k, v in o.items()})
    del net
worst = 0.0
for k in outs[0]:
    d = (outs[0][k] - outs[1][k]).abs().max().item() / max(outs[1][k].abs().max().item(), 1e-9)
    worst = max(worst, d)
    print(f"{k:12s} rel diff {d:.3e}  finite {bool(torch.isfinite(outs[0][k]).all())}")
print("worst", worst)
assert worst < 1e-3
